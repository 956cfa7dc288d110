import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
import torch
from dgs_amd import _ops
x = torch.randn(200000, 11).cuda(); n = torch.randn(1024, 11).cuda()
for _ in range(3): _ops.knn_indices(x, n, 3)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): _ops.knn_indices(x, n, 3)
torch.cuda.synchronize(); print("knn 200k x 1024 x 11: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
