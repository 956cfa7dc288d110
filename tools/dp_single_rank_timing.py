"""Per-rank cost of the data-parallel step structure on ONE GPU: the trainer is told it is one of two ranks over the real
backend (nccl = RCCL, a single rank), so graph 1a / async SH all-reduce / graph 1b / all-reduce of the rest / graph 2 all run;
only the wire time of a real multi-GPU exchange is missing.  Prints ms/step next to the single-graph step."""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
import torch.distributed as dist
import bench
from diff_surfel_rasterization import _C

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P, H, W = bench.WORKLOADS["metric"]
for pretend, shard in ((1, False), (2, False), (2, True)):   # shard: reduce-scatter / owner's Adam / all-gather of the SH segment (round 6) -- with
    tr = bench.build_trainer(P, H, W, dev)                   # ONE real rank the owner's rows are all rows: the structure's cost, not its saving
    tr.world = pretend
    tr.shard_optimizer = shard
    tr.enable_graph(capacity=24 * P)
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        tr.step()
    torch.cuda.synchronize()
    print("world (pretended) %d, sharded SH update %s: %.4f ms/step" % (pretend, shard, (time.perf_counter() - t0) / n * 1e3), flush=True)
    _C.set_capacity(0)
    del tr
dist.destroy_process_group()
