"""Dev tool (GPU box): the whole-step graph replayed on the CACHED trained scene (tools/diag/trained_cache.py): ms per step, and -- under
rocprofv3 --kernel-trace -- a trace whose last K preprocess launches are the timed steps (tools/graph_step_timeline.py <csv> K 0 0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trained_cache import load
import torch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
tr = load(dev)
tr.enable_graph(capacity=96 * tr.P)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(K):
    tr.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print("cached trained scene: %d live surfels, %.4f ms/step, %.1f views/s, list promise %d" % (tr.surfels.num_surfels, dt * 1e3, 1.0 / dt, tr._list_hint))
