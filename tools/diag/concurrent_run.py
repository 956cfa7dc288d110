"""Diagnostic: N replayed steps of the concurrent-views trainer alone (for a rocprofv3 kernel trace).  argv: [steps=20] [k=2] [concurrent=1]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
conc = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS[os.environ.get("WORKLOAD", "metric")]
tr = bench.build_trainer(P, H, W, dev, views_per_rank=k, concurrent_views=conc)
tr.enable_graph(capacity=24 * P)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("host loop returned after %.4f ms/step (device done after %.4f)" % (th / steps * 1e3, dt / steps * 1e3))
print("k=%d concurrent=%s: %.4f ms/step, %.4f ms/view, overflow=%d recoveries=%d" % (k, conc, dt / steps * 1e3, dt / steps / k * 1e3, int(tr._oflag.item()), tr.overflow_recoveries))
