"""Quick fwd/bwd timing of the rasterizer alone at a given size (development aid, GPU only)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from diff_surfel_rasterization import GaussianRasterizer, _C
from gpu_utils import settings_from_case
from scene_utils import small_case

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=200000)
ap.add_argument("--H", type=int, default=800)
ap.add_argument("--W", type=int, default=800)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--order", type=int, default=-1)
ap.add_argument("--scale-mul", type=float, default=1.0, help="scale of the splats (6 with --radius 2.2 and 96x96: thousands of entries per tile)")
ap.add_argument("--radius", type=float, default=4.0, help="camera orbit radius")
ap.add_argument("--sh-degree", type=int, default=3)
ap.add_argument("--capacity", type=int, default=0, help="> 0: capacity mode with that many list entries (no host read; every tile picks its sort kernel on the device)")
ap.add_argument("--grid-limit", type=int, default=0, help="diagnostic: blend kernels process only the N heaviest tiles")
ap.add_argument("--sort-regs", type=int, default=-1, help="0: LDS bitonic network, 1: register-resident network (default of the library)")
ap.add_argument("--opt", action="append", default=[], help="KEY=VALUE for dgs_set_option (e.g. 9=0: long-tile path off, 10 / 11: its divisors); repeatable")
ap.add_argument("--cluster", type=int, default=0, help="pack the first N surfels into a small ball (a dense knot on a few tiles, like a densified scene) at a quarter of their opacity")
a = ap.parse_args()
dev = "cuda:0"
if a.order >= 0:
    _C.set_option(1, a.order)
if a.sort_regs >= 0:
    _C.set_option(3, a.sort_regs)
if a.capacity > 0:
    _C.set_capacity(a.capacity)
for kv in a.opt:
    _C.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
cases = [small_case(P=a.P, H=a.H, W=a.W, seed=0, view=v * (64 // a.views), n_views=64, scale_mul=a.scale_mul, radius=a.radius, sh_degree=a.sh_degree)
         for v in range(a.views)]
if a.cluster > 0:
    g = torch.Generator().manual_seed(5)
    for c in cases:
        c["means3D"][:a.cluster] = torch.tensor([0.15, -0.1, 0.2]) + 0.12 * torch.randn(a.cluster, 3, generator=g)
        c["opacities"][:a.cluster] *= 0.25
    for c in cases[1:]:
        c["means3D"], c["opacities"] = cases[0]["means3D"], cases[0]["opacities"]
leaf = {k: cases[0][k].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
rasts = [GaussianRasterizer(settings_from_case(c, dev)) for c in cases]
gc = torch.randn(3, a.H, a.W, device=dev)
go = torch.randn(8, a.H, a.W, device=dev)


def step(i, bwd=True):
    m2 = torch.zeros_like(leaf["means3D"], requires_grad=True)
    color, radii, allmap = rasts[i % len(rasts)](means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"],
                                                 scales=leaf["scales"], rotations=leaf["rotations"])
    if bwd:
        torch.autograd.backward([color, allmap], [gc, go])
        for v in leaf.values():
            v.grad = None
    return color


for i in range(3):
    step(i)
torch.cuda.synchronize()
# bit pattern of one forward: A/B builds that promise a bit-identical forward print the same number
_c = step(0, bwd=False)
print("forward checksum %d" % int(_c.contiguous().view(torch.int32).to(torch.int64).sum().item()))
_C.profile_enable(True)
_C.profile_reset()
if a.grid_limit > 0:
    _C.set_option(5, a.grid_limit)
t = time.time()
for i in range(a.iters):
    step(i, bwd=False)
torch.cuda.synchronize()
tf = (time.time() - t) / a.iters
if a.grid_limit > 0:
    prf = _C.profile_read()
    print("forward blend limited to the %d heaviest tiles: %.3f ms/launch" % (a.grid_limit, prf["fwd_ms"] / max(prf["fwd_n"], 1)))
    _C.set_option(5, 0)
    _C.set_option(4, a.grid_limit)
    _C.profile_reset()
t = time.time()
for i in range(a.iters):
    step(i)
torch.cuda.synchronize()
tfb = (time.time() - t) / a.iters
pr = _C.profile_read()
print("P=%d %dx%d fwd %.3f ms, fwd+bwd %.3f ms%s" % (a.P, a.W, a.H, tf * 1e3, tfb * 1e3, " (capacity mode, overflow=%s)" % _C.read_overflow() if a.capacity > 0 else ""))
print("blend fwd %.3f ms/launch (%d), blend bwd %.3f ms/launch (%d), binning %.3f ms, S/launch %d" % (
    pr["fwd_ms"] / max(pr["fwd_n"], 1), pr["fwd_n"], pr["bwd_ms"] / max(pr["bwd_n"], 1), pr["bwd_n"], pr["bin_ms"] / max(pr["bin_n"], 1),
    pr["fwd_S"] / max(pr["fwd_n"], 1)))
