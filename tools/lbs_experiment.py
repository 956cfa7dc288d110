"""Time of dgs_deform_backward (lbs_bwd_kernel<ASM> + reduce) alone: direct back-to-back C-ABI calls (GPU-bound)."""
import ctypes, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
import bench
from dgs_amd import _ops
dev = torch.device("cuda:0")
tr = bench.build_trainer(200000, 800, 800, dev, n_views=4, n_targets=1)
s, d = tr.surfels, tr.deform
lib = _ops.load()
t = d.expand_time(tr.cameras[0].fid)
with torch.no_grad():
    out = d.forward_assembled(s, t)
idx = d._knn_seed
attrs = _ops.fused_node_mlp(d.network, d.nodes, t).detach()
N, M, H = s._xyz.shape[0], d.nodes.shape[0], d.hyper_dim
g = torch.Generator(device="cuda").manual_seed(0)
gm, gs, gr, go = (torch.randn(N, c, device=dev, generator=g) * 1e-3 for c in (3, 2, 4, 1))
g_attrs = torch.empty_like(attrs)
scratch = torch.empty(int(lib.dgs_lbs_scratch_bytes(M, H)), dtype=torch.uint8, device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    rc = lib.dgs_deform_backward(N, M, H, P(s._xyz), P(s.feature), s.feature.shape[1], P(idx), P(d.nodes), P(d._node_radius), P(d._node_weight),
                                 P(attrs), None, P(s._scaling), P(s._rotation), P(s._opacity), P(gm), P(gs), P(gr), P(go), P(s._xyz.grad),
                                 P(s._scaling.grad), P(s._rotation.grad), P(s._opacity.grad), P(s.feature.grad), P(d.nodes.grad),
                                 P(d._node_radius.grad), P(d._node_weight.grad), P(g_attrs), 1, P(scratch), st)
    assert rc == 0
def run(mask, iters=40):
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print("lbs_bwd + reduce: %.1f us per call" % run(0))
