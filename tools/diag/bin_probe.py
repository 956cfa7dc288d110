"""Dev tool (GPU box): binning time (count .. sort, library timing hook) of the eager forward on the uniform 200k scene and on the
`trained` scene.  usage: DGS_SURFEL_LIB=... python tools/diag/bin_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

import bench
from diff_surfel_rasterization import _C

dev = torch.device("cuda:0")


def run(tr, views, tag):
    tr.enable_graph(capacity=(96 if tag == "trained" else 24) * tr.P)
    for _ in range(3):
        tr.step()
    _C.profile_enable(2)
    tr._graph = None
    tr.enable_graph(capacity=tr._capacity)
    torch.cuda.synchronize()
    _C.profile_reset()
    for _ in range(12):
        tr.step()
    torch.cuda.synchronize()
    p = _C.profile_read()
    _C.profile_enable(0)
    print("%-8s binning %.1f us, preprocess %.1f us, blend fwd %.1f us, bwd %.1f us (in-graph, %d launches)" % (
        tag, 1e3 * p["bin_ms"] / max(p["bin_n"], 1), 1e3 * p["pre_ms"] / max(p["pre_n"], 1), 1e3 * p["fwd_ms"] / max(p["fwd_n"], 1),
        1e3 * p["bwd_ms"] / max(p["bwd_n"], 1), p["bin_n"]), flush=True)


tr = bench.build_trainer(200_000, 800, 800, dev)
run(tr, None, "metric")
del tr
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trained_cache import load
tr = load(dev)   # the cached fit (tools/diag/trained_cache.py): the same scene in every run
print("trained (cached): live surfels", tr.surfels.num_surfels)
run(tr, None, "trained")
