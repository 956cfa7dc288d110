"""Does the workload drift while training on noise targets? (development aid, GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dgs_amd.render import render
dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS["metric"]
tr = bench.build_trainer(P, H, W, dev)
s, d = tr.surfels, tr.deform
for it in range(14):
    cam = tr.cameras[tr.view_for(tr.iteration)]
    with torch.no_grad():
        dv = d(s.get_xyz.detach(), d.expand_time(cam.fid), s.feature, s.motion_mask)
        pkg = render(cam, s, tr.bg, dv['d_xyz'], dv['d_rotation'], dv['d_scaling'])
        r = pkg["radii"].float()
        print("it %2d view %2d vis %6d mean radius %.2f max %.0f  scale mean %.4f  |d_xyz| %.2e |d_scale| %.2e opac mean %.3f alpha mean %.3f" % (
            it, tr.view_for(tr.iteration), int((r > 0).sum()), float(r[r > 0].mean()), float(r.max()), float(s.get_scaling.mean()),
            float(dv['d_xyz'].abs().max()), float(dv['d_scaling'].abs().max()), float(s.get_opacity.mean()), float(pkg["alpha"].mean())))
    tr.step()
