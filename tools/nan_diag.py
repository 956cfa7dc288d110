"""Which intermediate first carries a huge / non-finite gradient? (development aid, GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dgs_amd.losses import training_loss
from dgs_amd.render import render
dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS["metric"]
tr = bench.build_trainer(P, H, W, dev)
s, d = tr.surfels, tr.deform
for it in range(8):
    cam = tr.cameras[it]; gt = tr.targets[it % len(tr.targets)]
    tr.bucket.zero()
    dv = d(s.get_xyz.detach(), d.expand_time(cam.fid), s.feature, s.motion_mask)
    pkg = render(cam, s, tr.bg, dv['d_xyz'], dv['d_rotation'], dv['d_scaling'])
    keep = {k: pkg[k] for k in ("render", "rend_normal", "surf_normal", "rend_dist", "depth", "alpha")}
    for v in keep.values():
        if v.requires_grad: v.retain_grad()
    loss = training_loss(pkg, gt)
    loss.backward()
    msg = ["it %d loss %.5f" % (it, float(loss))]
    for k, v in keep.items():
        if v.grad is not None:
            g = v.grad
            msg.append("%s|g|max %.2e%s" % (k, float(torch.nan_to_num(g, 0, 0, 0).abs().max()), "" if bool(torch.isfinite(g).all()) else " NONFINITE"))
    fl = tr.bucket.flat
    msg.append("flat max %.2e finite %s" % (float(torch.nan_to_num(fl, 0, 0, 0).abs().max()), bool(torch.isfinite(fl).all())))
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "feature"]
    for n, p in zip(names, tr.bucket.params[:7]):
        msg.append("%s %.1e" % (n, float(torch.nan_to_num(p.grad, 0, 0, 0).abs().max())))
    print("  ".join(msg), flush=True)
    with torch.no_grad():
        tr.opt_surfels.step(); (tr.opt_deform.step() if tr.opt_deform is not None else None)
