"""Development probe: run-to-run spread of the mean loss over iterations 801-900 of the small fit of tests/test_learning_gpu.py
(float atomics, captured step), with and without the reference's update order."""
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
kw = dict(iterations=900, device=dev, num_pts=6000, node_num=128, seed=0, warm_up=300, regularize_from=600, densify_from=200, opacity_reset_interval=500)
for order in (True, False):
    vals = []
    for r in range(int(os.environ.get("RUNS", "5"))):
        tr, losses = fit(data, os.path.join(tmp, "m%d%d" % (order, r)), reference_update_order=order, **kw)
        l = np.asarray(losses)
        vals.append((l[800:].mean(), l[700:800].mean(), tr.surfels.num_surfels))
    print("reference_update_order", order, "mean loss 801-900 / 701-800 / surfels:", [(round(a, 4), round(b, 4), n) for a, b, n in vals], flush=True)
    tr, losses = fit(data, os.path.join(tmp, "d%d" % order), reference_update_order=order, deterministic=True, graph=False, **kw)
    tr.set_deterministic(False)
    l = np.asarray(losses)
    print("   deterministic eager:", round(l[800:].mean(), 4), round(l[700:800].mean(), 4), tr.surfels.num_surfels, flush=True)
