"""Development probe: does training go on after an overflow recovery, with and without the reference's update order?"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
from dgs_amd.fit import fit  # noqa: E402
from dgs_amd.synthetic import write_dynamic_dnerf  # noqa: E402

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "scene")
write_dynamic_dnerf(data, n_train=24, n_test=2, H=128, W=128, device=dev)
for cap in [int(c) for c in os.environ.get("CAPS", "60000,120000").split(",")]:
    for order in (False, True):
        rec = []

        def hook(it, t):
            if it % 50 == 0:
                rec.append((it, t.overflow_recoveries, float(t.surfels._xyz.detach().double().abs().sum()), float(t.deform.network.gaussian_warp.weight.detach().double().abs().sum())))
        try:
            tr, losses = fit(data, os.path.join(tmp, "m"), iterations=900, device=dev, num_pts=6000, node_num=128, seed=0, warm_up=300, regularize_from=600,
                             densify_from=200, opacity_reset_interval=500, list_capacity=cap, reference_update_order=order, on_iteration=hook, deterministic=True)
            tr.set_deterministic(False)
        except RuntimeError as e:
            print("cap", cap, "order", order, "->", e)
            continue
        print("cap", cap, "order", order, "recoveries", tr.overflow_recoveries, "surfels", tr.surfels.num_surfels, "mean loss last 100 %.4f" % np.mean(losses[-100:]))
        print("   ", [(it, r, round(x, 2), round(w, 4)) for it, r, x, w in rec[-8:]])
