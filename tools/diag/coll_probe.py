"""Which collectives the two backends of the tests accept on device tensors (gloo: two ranks sharing cuda:0; nccl = RCCL: one rank):
reduce_scatter_tensor / all_gather_into_tensor, in place (output chunk = slice of the input) and asynchronous."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(r, n, backend, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=r, world_size=n, device_id=dev)
    else:
        dist.init_process_group(backend, rank=r, world_size=n)
    c = 1 << 20
    for name, fn in (
        ("rs in place", lambda full: dist.reduce_scatter_tensor(full[r * c:(r + 1) * c], full)),
        ("rs in place async", lambda full: dist.reduce_scatter_tensor(full[r * c:(r + 1) * c], full, async_op=True).wait()),
        ("ag in place", lambda full: dist.all_gather_into_tensor(full, full[r * c:(r + 1) * c])),
        ("ag in place async", lambda full: dist.all_gather_into_tensor(full, full[r * c:(r + 1) * c], async_op=True).wait()),
        ("rs bf16 in place", lambda full: dist.reduce_scatter_tensor(full.bfloat16()[r * c:(r + 1) * c], full.bfloat16())),
    ):
        full = (torch.arange(n * c, dtype=torch.float32, device=dev) % 97) + 10 * r
        ref = full.clone()
        try:
            fn(full)
            torch.cuda.synchronize()
            if name.startswith("rs in"):
                want = sum(((torch.arange(n * c, dtype=torch.float32, device=dev) % 97) + 10 * q) for q in range(n))[r * c:(r + 1) * c]
                ok = bool(torch.equal(full[r * c:(r + 1) * c], want))
            elif name.startswith("ag"):
                want = torch.cat([((torch.arange(n * c, dtype=torch.float32, device=dev) % 97) + 10 * q)[q * c:(q + 1) * c] for q in range(n)])
                ok = bool(torch.equal(full, want))
            else:
                ok = True
            print(backend, r, name, "ok" if ok else "WRONG VALUES", flush=True)
        except Exception as e:
            print(backend, r, name, "FAIL", type(e).__name__, str(e)[:160].replace("\n", " "), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2, "gloo", 29541), nprocs=2)
    mp.spawn(w, args=(1, "nccl", 29542), nprocs=1)
