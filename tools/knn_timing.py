"""dgs_knn_refine alone on the bench scene (200k surfels, 1024 nodes, trainer's storage order): time per call, and what the
kernel's filters leave per point / per wave (computed with torch): candidates within the seed bound, 32-node blocks a wave's
box touches."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
import bench
from dgs_amd import _ops
dev = torch.device("cuda:0")
tr = bench.build_trainer(200000, 800, 800, dev, n_views=4, n_targets=1)
s, d = tr.surfels, tr.deform
H = d.hyper_dim
x, f, nodes = s._xyz.detach(), s.feature.detach()[:, :H].contiguous(), d.nodes.detach()
seed = _ops.knn_indices2(x, f, nodes, 3)
def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print("refine (seed = exact answer): %.1f us   plain scan: %.1f us" % (timed(lambda: _ops.knn_indices2(x, f, nodes, 3, seed=seed)),
                                                                     timed(lambda: _ops.knn_indices2(x, f, nodes, 3))))
xq = torch.cat([x, f], 1)
N, M = x.shape[0], nodes.shape[0]
cand = torch.empty(N, dtype=torch.long, device=dev)
T = torch.empty(N, device=dev)
for i in range(0, N, 8192):
    dfull = torch.cdist(xq[i:i + 8192].double(), nodes.double()) ** 2
    t = dfull.gather(1, seed[i:i + 8192]).max(1).values
    d3 = torch.cdist(x[i:i + 8192].double(), nodes[:, :3].double()) ** 2
    cand[i:i + 8192] = (d3 <= t[:, None]).sum(1)
    T[i:i + 8192] = t.float()
print("candidates per point (d3 <= T): mean %.1f  p50 %d  p99 %d  max %d  over the list cap of 12: %.3f %% of the points, %.1f %% of the waves"
      % (cand.float().mean(), cand.median(), cand.float().quantile(0.99), cand.max(), 100 * (cand > 12).float().mean(),
         100 * (cand[:N // 128 * 128].view(-1, 128) > 12).any(1).float().mean()))
r = T.sqrt()
lo = (x - r[:, None])[:N // 128 * 128].view(-1, 128, 3).min(1).values
hi = (x + r[:, None])[:N // 128 * 128].view(-1, 128, 3).max(1).values
nb = nodes[:, :3].view(M // 32, 32, 3)
blo, bhi = nb.min(1).values, nb.max(1).values
touch = ((blo[None] <= hi[:, None]) & (bhi[None] >= lo[:, None])).all(-1).sum(1)
print("32-node blocks touched per wave (of %d): mean %.1f  p50 %d  p99 %d  max %d" % (M // 32, touch.float().mean(), touch.median(),
                                                                                      touch.float().quantile(0.99), touch.max()))
# would a per-node neighbour list certify the seed?  a = the seed nearest in 3-D, r = sqrt(T); every candidate n has
# |n - a| <= r + |x - a|, so if that is below R_a (3-D distance from a to the first node NOT in its list of L nearest nodes) the
# list holds all candidates
n3 = nodes[:, :3]
dn = torch.cdist(n3.double(), n3.double())
dn_sorted = dn.sort(1).values
d3seed = torch.stack([(x - n3[seed[:, k]]).norm(dim=1) for k in range(3)], 1)
a = seed.gather(1, d3seed.argmin(1, keepdim=True)).squeeze(1)
need = r + d3seed.min(1).values
for L in (32, 48, 64, 96):
    Ra = dn_sorted[:, L].float()[a]
    ok = need < Ra
    wave_ok = ok[:N // 128 * 128].view(-1, 128).all(1)
    print("list of %3d nodes: certifies %.2f %% of the points, %.2f %% of the waves (mean R_a %.3f, mean need %.3f)" % (
        L, 100 * ok.float().mean(), 100 * wave_ok.float().mean(), Ra.mean(), need.mean()))
