"""Development probe: gradients of one step with the mask terms on the device, parameter by parameter, against the CPU oracle path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)
import test_train_step_golden as t  # noqa: E402
from oracle_raster_op import OracleRasterizer  # noqa: E402

g = np.load(t.GOLD.replace(".npz", "_masks.npz"))
out = {}
for tag, dev, rcls, fused in (("cpu", torch.device("cpu"), OracleRasterizer, False), ("gpu", torch.device("cuda:0"), None, True)):
    tr, c, kinds, draws = t._build(g, dev, rcls, fused, masks=True)
    tr.set_regime(warmup=True, lambda_normal=0.0, lambda_dist=0.0)
    bgs = [torch.tensor([0.2, 0.5, 0.7], device=dev), torch.tensor([0.6, 0.1, 0.3], device=dev)]
    tr.bg_draw = lambda _q=bgs: _q.pop(0)
    tr.iteration = 7995
    cam, gt = tr.cameras[2], tr.targets[2]
    loss = tr._fwd_bwd(cam, gt)
    s = tr.surfels
    names = {"xyz": s._xyz, "opacity": s._opacity, "scaling": s._scaling, "rotation": s._rotation, "feature": s.feature,
             "sh": s._features if fused else s._features_dc}
    out[tag] = {k: v.grad.detach().cpu().clone() for k, v in names.items()}
    out[tag]["stats"] = tr.bucket.extra.detach().cpu().clone()
    print(tag, "loss", float(loss), {k: (int(torch.isnan(v).sum()), float(v.nan_to_num().abs().sum())) for k, v in out[tag].items()})
for k in ("xyz", "opacity", "scaling", "rotation", "feature", "stats"):
    a, b = out["cpu"][k], out["gpu"][k]
    n = min(a.shape[0], b.shape[0])
    print(k, "max |cpu - gpu|", float((a[:n] - b[:n]).abs().nan_to_num(nan=9e9).max()), "max |cpu|", float(a.abs().max()))
