"""Dev tool (GPU box): what does the `trained` bench workload look like?  Fits the synthetic D-NeRF-format scene with the given
settings and prints live surfels, per-view R / visible / tile-list length distribution, opacity and radius quantiles.
usage: python tools/diag/trained_stats.py <iterations> <num_pts> <n_sphere> <n_plate> <detail> [H]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch

its, num_pts, n_sph, n_pl, detail = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
H = int(sys.argv[6]) if len(sys.argv) > 6 else 800
import shutil
import tempfile

from dgs_amd.fit import fit
from dgs_amd.synthetic import DynamicTruth, write_dynamic_dnerf
from diff_surfel_rasterization import _C

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="dgs_stats_")
t0 = time.time()
write_dynamic_dnerf(os.path.join(tmp, "scene"), n_train=48, n_test=2, H=H, W=H, device=dev, truth=DynamicTruth(n_sph, n_pl, detail=detail))
t1 = time.time()
marks = {}
def hook(it, tr):
    if it % 1000 == 0:
        marks[it] = (tr.surfels.num_surfels, tr.P, time.time() - t1)
tr, losses = fit(os.path.join(tmp, "scene"), os.path.join(tmp, "model"), iterations=its, device=dev, num_pts=num_pts, node_num=512, seed=0,
                 node_densify_at=10 ** 9, on_iteration=hook)
t2 = time.time()
shutil.rmtree(tmp, ignore_errors=True)
print("dataset %.1f s, fit %.1f s; surfels by iteration (live, slots, s):" % (t1 - t0, t2 - t1), marks)
l = np.asarray(losses)
print("loss per 1000:", np.round(l[: len(l) // 1000 * 1000].reshape(-1, 1000).mean(1), 4))
s = tr.surfels
alive = s.alive
print("live", int(alive.sum()), "slots", tr.P, "opacity q10/50/90", np.round(np.quantile(s.get_opacity[alive].detach().cpu().numpy(), [.1, .5, .9]), 3))
from dgs_amd.render import render
bg = torch.zeros(3, device=dev)
for v in (0, 17, 33):
    cam = tr.cameras[v] if hasattr(tr, "cameras") else tr.cams[v]
    with torch.no_grad():
        dv = tr.deform(s.get_xyz.detach(), tr.deform.expand_time(cam.fid), s.feature, s.motion_mask)
        e = torch.empty(0, device=dev)
        R, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
            bg, (s.get_xyz + dv["d_xyz"]).contiguous(), e, s.get_opacity.contiguous(), (s.get_scaling + dv["d_scaling"]).contiguous(),
            s.get_rotation_bias(dv["d_rotation"]).contiguous(), 1.0, e, cam.world_view_transform, cam.full_proj_transform,
            float(np.tan(cam.FoVx * 0.5)), float(np.tan(cam.FoVy * 0.5)), H, H, s.get_features.contiguous(), s.active_sh_degree, cam.camera_center, False, True)
    torch.cuda.synchronize()
    T = ((H + 15) // 16) ** 2
    off = _C.debug_layout(1, width=H, height=H)
    rng = img.cpu().numpy()[off[2]:off[2] + T * 8].view(np.uint32).reshape(T, 2)
    ln = (rng[:, 1] - rng[:, 0]).astype(np.int64)
    r = radii.cpu().numpy()
    print("view %d: R %d visible %d  tiles non-empty %d  list len mean(non-empty) %.0f p50 %d p90 %d p99 %d max %d  >2048: %d tiles  radius px q50/90/99 %s  alpha coverage %.2f"
          % (v, R, int((r > 0).sum()), int((ln > 0).sum()), ln[ln > 0].mean(), *np.quantile(ln[ln > 0], [.5, .9, .99]).astype(int), ln.max(), int((ln > 2048).sum()),
             np.quantile(r[r > 0], [.5, .9, .99]).astype(int), float((allmap[1] > 0.5).float().mean())))
