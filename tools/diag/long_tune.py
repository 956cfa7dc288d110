"""Dev tool (GPU box): the long-tile path (dgs_set_option 9 / 10 / 11) on the `trained` bench scene and on the uniform metric scene:
blend kernel times per setting.  usage: python tools/diag/long_tune.py [pre_iterations]"""
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
SETTINGS = [("off", 0, 800, 512), ("150/512", 1, 150, 512), ("400/512", 1, 400, 512), ("800/512", 1, 800, 512), ("1600/512", 1, 1600, 512), ("150/256", 1, 150, 256), ("150/1024", 1, 150, 1024), ("150/2048", 1, 150, 2048)]


def run(tr, views, tag):
    tr._graph = None
    _C.set_capacity(0)
    _C.set_option(6, 0)
    snap, it0 = tr._snapshot(), tr.iteration
    for name, on, df, db in SETTINGS:
        _C.set_option(9, on); _C.set_option(10, df); _C.set_option(11, db)
        tr._restore(snap); tr.iteration = it0
        for v in views:
            tr.iteration = v
            tr.step()
        torch.cuda.synchronize()
        tr._restore(snap)
        _C.profile_enable(1)
        _C.profile_reset()
        for _ in range(3):
            for v in views:
                tr.iteration = v
                tr.step()
        torch.cuda.synchronize()
        p = _C.profile_read()
        _C.profile_enable(0)
        print("%-8s %-10s fwd blend %.3f ms, bwd blend %.3f ms" % (tag, name, p["fwd_ms"] / max(p["fwd_n"], 1), p["bwd_ms"] / max(p["bwd_n"], 1)), flush=True)
    _C.set_option(9, 1); _C.set_option(10, 800); _C.set_option(11, 512)
    tr._restore(snap)


tr = bench.build_trainer(200_000, 800, 800, dev)
run(tr, [0, 8, 16, 24], "metric")
del tr
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trained_cache import load
tr = load(dev)   # the cached fit (tools/diag/trained_cache.py): the same scene in every run
print("trained (cached): live surfels", tr.surfels.num_surfels)
run(tr, [5, 17, 33, 41], "trained")
