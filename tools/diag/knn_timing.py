"""Dev tool (GPU box): seeded KNN refine on a clustered 'trained-like' state and on the uniform state; checks against the plain scan.
usage: [DGS_KNN_REFINE=box] python tools/diag/knn_timing.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
import torch

from dgs_amd import _ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for name, N, fscale, M, coherent in (("uniform, features ~0", 200_000, 0.01, 512, False), ("features drifted", 125_000, 0.6, 512, False),
                                  ("sorted, 1024 nodes", 200_000, 0.01, 1024, True)):
    x = torch.rand(N, 3, device=dev, generator=g) * 2 - 1
    f = torch.randn(N, 8, device=dev, generator=g) * fscale
    nodes = torch.cat([x[torch.randperm(N, device=dev, generator=g)[:M]], torch.randn(M, 8, device=dev, generator=g) * 0.01], 1).contiguous()
    if coherent:      # nodes along a Morton curve, points in the order of their nearest node (the trainer's storage order)
        q = ((nodes[:, :3] + 1) * 511.5).long().clamp(0, 1023)
        code = torch.zeros(M, dtype=torch.long, device=dev)
        for b in range(10):
            for c in range(3):
                code |= ((q[:, c] >> b) & 1) << (3 * b + c)
        nodes = nodes[torch.argsort(code)].contiguous()
        near = _ops.knn_indices2(x, f, nodes, 3)[:, 0]
        order = torch.argsort(near, stable=True)
        x, f = x[order].contiguous(), f[order].contiguous()
    want = _ops.knn_indices2(x, f, nodes, 3)
    stale = _ops.knn_indices2(x + 0.01 * torch.randn(N, 3, device=dev, generator=g), f, nodes, 3)
    seed = stale.clone()
    got = _ops.knn_indices2(x, f, nodes, 3, seed=seed)
    ok = torch.equal(got, want)
    if os.environ.get("KNN_COUNT"):
        c = got[:, 0].float()
        print("hits per point: mean %.1f  p50 %d  p99 %d  max %d" % (c.mean(), c.median(), c.quantile(0.99), c.max()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    seeds = [stale.clone() for _ in range(20)]
    e0.record()
    for sd in seeds:
        _ops.knn_indices2(x, f, nodes, 3, seed=sd)
    e1.record()
    torch.cuda.synchronize()
    print("%-22s N %d: exact %s, %.1f us per refine" % (name, N, ok, e0.elapsed_time(e1) * 1000 / 20))
