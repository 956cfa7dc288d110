"""Diagnostic: the data-parallel step's structure on ONE GPU (RCCL with one rank, the trainer told it is one of two) for a kernel trace.
argv: [steps=12] [shard=1]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
import torch.distributed as dist
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shard = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P, H, W = bench.WORKLOADS["metric"]
tr = bench.build_trainer(P, H, W, dev)
tr.world = 2
tr.shard_optimizer = shard
tr.enable_graph(capacity=24 * P)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
torch.cuda.synchronize()
print("pretended world 2, sharded %s: %.4f ms/step" % (shard, (time.perf_counter() - t0) / steps * 1e3))
dist.destroy_process_group()
