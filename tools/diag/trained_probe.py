"""Development probe: the pre-fit of bench.py's `trained` workload with its density-control log."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
import bench  # noqa: E402
from dgs_amd.fit import fit  # noqa: E402
from dgs_amd.synthetic import DynamicTruth, write_dynamic_dnerf  # noqa: E402

dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS["trained"]
tmp = tempfile.mkdtemp()
write_dynamic_dnerf(os.path.join(tmp, "scene"), n_train=48, n_test=2, H=H, W=W, device=dev, truth=DynamicTruth(24000, 16000, detail=0.3))
order = os.environ.get("ORDER", "1") == "1"
lines = []
rec = {}
def hook(it, t):
    lo, hi = [int(x) for x in os.environ.get("WATCH", "0,0").split(",")]
    if lo <= it <= hi:
        st = t.opt_surfels.status.tolist()
        print(it, "tr.it", t.iteration, "rec", t.overflow_recoveries, "status", st, "t", float(t.opt_surfels.t), "oflag", int(t._oflag.item()), "cap", getattr(t, "_capacity", 0), "hint", getattr(t, "_list_hint", 0),
              "|g| %.3e" % float(t.bucket.flat[:t.bucket.n_grad].abs().sum()), "xyz %.4f" % float(t.surfels._xyz.detach().double().abs().sum()),
              "warp %.5f" % float(t.deform.network.gaussian_warp.weight.detach().double().abs().sum()), "warmup", t.warmup, flush=True)
    if os.environ.get("DEEP") and lo <= it <= hi:
        with torch.no_grad():
            sf, d = t.surfels, t.deform
            a = sf.alive
            for v in (0, 7, 19, 31):
                cam = t.cameras[v]
                dv = d(sf.get_xyz.detach(), d.expand_time(cam.fid), sf.feature, sf.motion_mask)
                sc = torch.exp(sf._scaling.detach()) + dv["d_scaling"]
                print("   view", v, "nan d_xyz %d d_rot %d d_scale %d" % tuple(int(torch.isnan(dv[k][a]).sum()) for k in ("d_xyz", "d_rotation", "d_scaling")),
                      "max|d_xyz| %.3f max|d_scale| %.4f min scale+d %.5f max %.4f" % (float(dv["d_xyz"][a].abs().max()), float(dv["d_scaling"][a].abs().max()), float(sc[a].min()), float(sc[a].max())),
                      "neg scale", int((sc[a] <= 0).any(dim=1).sum()), flush=True)
            print("   nodes: radius %.4f..%.4f weight %.3f..%.3f hyper %.3f..%.3f | surfel hyper %.3f..%.3f opacity nan %d xyz nan %d" % (
                float(d.node_radius.min()), float(d.node_radius.max()), float(d.node_weight.min()), float(d.node_weight.max()), float(d.nodes[:, 3:].min()), float(d.nodes[:, 3:].max()),
                float(sf.feature[a].min()), float(sf.feature[a].max()), int(torch.isnan(sf._opacity[a]).sum()), int(torch.isnan(sf._xyz[a]).sum())), flush=True)
    if it % 100 == 0:
        s_ = t.surfels
        a = s_.alive
        rec[it] = (t.overflow_recoveries, float(torch.exp(s_._scaling.detach()[a]).max()), float(torch.exp(s_._scaling.detach()[a]).mean()), float(s_.get_opacity.detach()[a].mean()))
tr, losses = fit(os.path.join(tmp, "scene"), os.path.join(tmp, "model"), iterations=int(os.environ.get("ITERS", "10000")), device=dev, num_pts=P, node_num=512,
                 seed=int(os.environ.get("SEED", "0")), warm_up=int(os.environ.get("WARM", "3000")), regularize_from=8000, node_densify_at=10 ** 9, deterministic=os.environ.get("DET", "1") == "1", log=lines.append, reference_update_order=order, on_iteration=hook, graph=None if os.environ.get("GRAPH", "1") == "1" else False,
                 list_capacity=int(os.environ["CAP"]) if os.environ.get("CAP") else None)
keep = [l for l in lines if "cloned" in l]
for l in (lines[-12:] if os.environ.get("TAIL") else keep[:3] + keep[24:36] + keep[-14:]):
    print(l)
if os.environ.get("REC"):
    for it in sorted(rec):
        print(it, "recoveries %d max scale %.3f mean scale %.4f mean opacity %.3f" % rec[it], [l.split("] ")[1] for l in lines if l.startswith("[%d] cloned" % it)])
print("recoveries", tr.overflow_recoveries, "status", tr.opt_surfels.status.tolist(), "t", float(tr.opt_surfels.t), "P", tr.P, "capacity", getattr(tr, "_capacity", None),
      "losses 3690-3710", [round(float(x), 4) for x in losses[3690:3710]], "losses 3790-3800", [round(float(x), 4) for x in losses[3790:3800]])
print("final", tr.surfels.num_surfels, "mean loss last 500", float(np.mean(losses[-500:])))
