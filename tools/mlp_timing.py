"""Time of the node-MLP kernels alone (dgs_mlp_forward = pack + forward chain, dgs_mlp_backward = backward chain + weight
gradients), M control nodes, back-to-back C-ABI calls.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel figures."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
from dgs_amd import _ops
from dgs_amd.deform import DeformMLP

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DeformMLP().to(dev)
params = _ops.node_mlp_params(net)
x = torch.randn(M, 3, device=dev)
t = torch.full((M, 1), 0.37, device=dev)
outs = [torch.zeros_like(p) for p in params]
g = torch.randn(M, 13, device=dev) * 1e-3


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


attrs, packed, saved = _ops._mlp_forward_raw(x, t, (1.0, 0.0, 0.0, 0.0), params)
print("M=%d  forward (pack + chain): %.1f us   backward (chain + weight gradients): %.1f us" % (
    M, timed(lambda: _ops._mlp_forward_raw(x, t, (1.0, 0.0, 0.0, 0.0), params)),
    timed(lambda: _ops._mlp_backward_raw(g, packed, saved, outs, True))))
