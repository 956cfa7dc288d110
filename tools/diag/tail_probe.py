"""Dev tool (GPU box): how tail-bound are the blend kernels on the `trained` bench scene?  Fits the scene like bench.py does, then
times forward / backward blend of one view with the kernels restricted to the N heaviest tiles (dgs_set_option 5 / 4).
usage: python tools/diag/tail_probe.py [pre_iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch

import bench
from diff_surfel_rasterization import _C

dev = torch.device("cuda:0")
tr, _ = bench.trained_trainer(100_000, 800, 800, dev, int(sys.argv[1]) if len(sys.argv) > 1 else 10000)
tr._graph = None
_C.set_capacity(0)
_C.set_option(6, 0)
tr.set_regime(warmup=False, lambda_normal=0.02, lambda_dist=1000.0)
print("live surfels", tr.surfels.num_surfels)
for N in (0, 1, 4, 16, 64, 256):
    _C.set_option(5, N)
    _C.set_option(4, N)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    _C.profile_enable(1)
    _C.profile_reset()
    for _ in range(6):
        tr.iteration = 5          # the same view every time
        tr.step()
    torch.cuda.synchronize()
    p = _C.profile_read()
    _C.profile_enable(0)
    print("heaviest %4s tiles: fwd blend %.3f ms, bwd blend %.3f ms   (S per launch %d)" % (N or "all", p["fwd_ms"] / max(p["fwd_n"], 1), p["bwd_ms"] / max(p["bwd_n"], 1), p["fwd_S"] / max(p["fwd_n"], 1)))
_C.set_option(5, 0)
_C.set_option(4, 0)
