"""Where does the deformation forward/backward time go? (development aid, GPU only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dgs_amd import deform as D
dev = torch.device("cuda:0")
tr = bench.build_trainer(200000, 800, 800, dev)
s, d = tr.surfels, tr.deform
cam = tr.cameras[0]
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-28s %.3f ms" % (name, (time.perf_counter() - t) / n * 1e3))
x = s.get_xyz.detach(); feat = s.feature; t = d.expand_time(cam.fid)
xq = torch.cat([x, feat[..., :8]], -1).detach(); nodes = d.nodes.detach()
timeit("knn_indices", lambda: D.knn_indices(xq, nodes, 3))
timeit("nn_weights (no grad)", lambda: d.nn_weights(x, feat))
timeit("node_deform MLP fwd", lambda: d.node_deform(t))
def full_fwd():
    return d(x, t, feat, s.motion_mask)
timeit("deform fwd total", full_fwd)
def full_fb():
    o = d(x, t, feat, s.motion_mask)
    (o['d_xyz'].sum() + o['d_rotation'].sum() + o['d_scaling'].sum()).backward()
timeit("deform fwd+bwd total", full_fb)
with torch.no_grad():
    timeit("deform fwd total (no_grad)", full_fwd)
# python overhead check: count kernels
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    full_fb(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
