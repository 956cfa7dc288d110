"""Would a movement certificate let the step skip its neighbour search?  (round 6, sizing only)

For every surfel the K nearest control nodes stay the same, in the same order, while  2 (|x - x0| + max_j |n_j - n0_j|)  is below the
smallest gap between consecutive distances of its K + 1 nearest nodes at the time (x0, n0) of its last search.  This probe runs the
metric workload's trainer and reports, per step: how far surfels and nodes move (11-D), and which fraction of the surfels would fail
that test if every failing surfel were searched again (its certificate renewed) -- the work a certificate-based search would still do.
Run on the GPU box:  python tools/diag/knn_cert_probe.py [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
tr = bench.build_trainer(200_000, 800, 800, dev)
d, s = tr.deform, tr.surfels
H = d.hyper_dim


def points():
    return torch.cat([s._xyz.detach(), s.feature.detach()[:, :H]], 1).clone(), d.nodes.detach().clone()


def certificate(x, n):
    out = torch.empty(x.shape[0], device=dev)
    for a in range(0, x.shape[0], 1 << 15):
        dist = torch.cdist(x[a:a + (1 << 15)], n).topk(d.K + 1, dim=1, largest=False).values
        out[a:a + (1 << 15)] = (dist[:, 1:] - dist[:, :-1]).min(1).values
    return out


for _ in range(5):
    tr.step()
x0, n0 = points()
gap = certificate(x0, n0)
q = torch.tensor([0.001, 0.01, 0.1, 0.5], device=dev)
print("smallest consecutive gap of the K + 1 nearest nodes: quantiles 0.1%% 1%% 10%% 50%% = %s" % ["%.2e" % v for v in torch.quantile(gap, q).tolist()])
node_path = torch.zeros((), device=dev)          # cumulative bound on every node's movement
path_at = torch.zeros(x0.shape[0], device=dev)   # its value when the surfel's certificate was made
n_prev = n0
for it in range(steps):
    tr.step()
    x, n = points()
    step_nodes = (n - n_prev).norm(dim=1).max()
    node_path = node_path + step_nodes
    n_prev = n
    move = (x - x0).norm(dim=1)
    fail = 2.0 * (move + node_path - path_at) >= gap
    nf = int(fail.sum())
    if it < 10 or it % 10 == 9:
        print("step %3d: nodes moved max %.2e, surfels moved (since their search) median %.2e max %.2e; %6d of %d surfels searched again (%.2f %%)"
              % (it, float(step_nodes), float(move.median()), float(move.max()), nf, x.shape[0], 100.0 * nf / x.shape[0]))
    if nf:
        x0[fail] = x[fail]
        gap[fail] = certificate(x[fail], n)
        path_at[fail] = node_path
