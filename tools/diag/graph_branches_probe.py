"""Diagnostic: how many branches of a captured HIP graph run at the same time?  One graph: the capture stream forks into n - 1 side
streams, every branch launches one ~60-us kernel on its own 64-MB buffer, all join.  Run under rocprofv3 --kernel-trace and look at
the starts.  argv: [branches=3]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
bufs = [torch.ones(16 << 20, device=dev) for _ in range(n)]
streams = [torch.cuda.Stream() for _ in range(n)]


def work(b):
    for _ in range(3):       # three dependent passes per branch: ~3 x 25 us
        b.mul_(1.0001)


for b in bufs:
    work(b)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s0 = streams[0]
s0.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=s0):
    for s in streams[1:]:
        s.wait_stream(s0)
    for s, b in zip(streams, bufs):
        with torch.cuda.stream(s):
            work(b)
    for s in streams[1:]:
        s0.wait_stream(s)
torch.cuda.current_stream().wait_stream(s0)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("%d branches x 3 kernels: %.1f us per replay" % (n, (time.perf_counter() - t0) / 20 * 1e6))
