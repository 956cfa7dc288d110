"""Dev tool (GPU box): the `trained` bench scene once -- distribution of the tile lists and of the traversed lengths (tile_last), then the
blend kernels restricted to the N heaviest tiles (dgs_set_option 5 / 4): how much of a launch is its longest tiles?
usage: python tools/diag/trained_tail.py [pre_iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch

import bench
from diff_surfel_rasterization import _C

dev = torch.device("cuda:0")
tr, _ = bench.trained_trainer(100_000, 800, 800, dev, int(sys.argv[1]) if len(sys.argv) > 1 else 10000)
tr._graph = None
_C.set_capacity(0)
_C.set_option(6, 0)
tr.set_regime(warmup=False, lambda_normal=0.02, lambda_dist=1000.0)
print("live surfels", tr.surfels.num_surfels, "slots", tr.P)
s = tr.surfels
H = W = 800
T = ((H + 15) // 16) ** 2
for v in (5, 17, 33):
    cam = tr.cameras[v]
    with torch.no_grad():
        dv = tr.deform(s.get_xyz.detach(), tr.deform.expand_time(cam.fid), s.feature, s.motion_mask)
        e = torch.empty(0, device=dev)
        R, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
            tr.bg, (s.get_xyz + dv["d_xyz"]).contiguous(), e, s.get_opacity.contiguous(), (s.get_scaling + dv["d_scaling"]).contiguous(),
            s.get_rotation_bias(dv["d_rotation"]).contiguous(), 1.0, e, cam.world_view_transform, cam.full_proj_transform,
            float(np.tan(cam.FoVx * 0.5)), float(np.tan(cam.FoVy * 0.5)), H, W, s.get_features.contiguous(), s.active_sh_degree, cam.camera_center, False, False)
    torch.cuda.synchronize()
    off = _C.debug_layout(1, width=W, height=H)
    ib = img.cpu().numpy()
    rng = ib[off[2]:off[2] + T * 8].view(np.uint32).reshape(T, 2)
    ln = (rng[:, 1] - rng[:, 0]).astype(np.int64)
    tl = ib[off[3]:off[3] + T * 4].view(np.uint32).astype(np.int64)
    goff = _C.debug_layout(0, P=tr.P)
    gb = geom.cpu().numpy()
    rc = gb[goff[4]:goff[4] + tr.P * 8].view(np.uint32).reshape(tr.P, 2)
    rad = radii.cpu().numpy()
    area = ((rc[:, 0] >> 16).astype(np.int64) - (rc[:, 0] & 0xffff)) * ((rc[:, 1] >> 16).astype(np.int64) - (rc[:, 1] & 0xffff))
    area[rad <= 0] = 0
    chunk = (tr.P + 255) // 256
    per_wg = np.add.reduceat(area, np.arange(0, tr.P, chunk))
    print("  tile rectangles: visible %d, tiles per visible surfel mean %.1f p99 %d max %d ; surfels above 32 tiles: %d (%.1f %% of the pairs), above 256: %d (%.1f %%) ; pairs per binning workgroup: mean %.0f max %d"
          % ((rad > 0).sum(), area[rad > 0].mean(), np.quantile(area[rad > 0], .99), area.max(), (area > 32).sum(), 100.0 * area[area > 32].sum() / area.sum(),
             (area > 256).sum(), 100.0 * area[area > 256].sum() / area.sum(), per_wg.mean(), per_wg.max()))
    srt = np.sort(tl)[::-1]
    print("view %d: R %d | list length: max %d p99 %d p90 %d mean(non-empty) %.0f | traversed (tile_last): sum %d max %d top-8 %s p99 %d p90 %d p50 %d | tiles with traversed > 256: %d, > 512: %d, > 1024: %d"
          % (v, R, ln.max(), *np.quantile(ln[ln > 0], [.99, .9]).astype(int), ln[ln > 0].mean(), tl.sum(), tl.max(), srt[:8].tolist(),
             *np.quantile(tl[tl > 0], [.99, .9, .5]).astype(int), int((tl > 256).sum()), int((tl > 512).sum()), int((tl > 1024).sum())))
for N in ():
    _C.set_option(5, N)
    _C.set_option(4, N)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    _C.profile_enable(1)
    _C.profile_reset()
    for _ in range(6):
        tr.iteration = 5          # the same view every time
        tr.step()
    torch.cuda.synchronize()
    p = _C.profile_read()
    _C.profile_enable(0)
    print("heaviest %4s tiles: fwd blend %.3f ms, bwd blend %.3f ms   (S per launch %d)" % (N or "all", p["fwd_ms"] / max(p["fwd_n"], 1), p["bwd_ms"] / max(p["bwd_n"], 1), p["fwd_S"] / max(p["fwd_n"], 1)))
_C.set_option(5, 0)
_C.set_option(4, 0)
