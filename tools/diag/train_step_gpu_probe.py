"""Development probe: the joint-stage golden on the device, iteration by iteration."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)
import test_train_step_golden as t  # noqa: E402
from dgs_amd import fit as fit_mod  # noqa: E402

g = np.load(t.GOLD.replace(".npz", "_masks.npz") if os.environ.get("MASKS") == "1" else t.GOLD)
ref = g["per_it"]
orig = fit_mod.run_iteration
k = [0]


def ri(tr, it, sch, **kw):
    try:
        loss = orig(tr, it, sch, **kw)
    except AssertionError as e:
        print(it, "ASSERT", str(e)[:200])
        raise
    s = tr.surfels
    a = s.alive
    p = ref[k[0]]
    print(it, "n", s.num_surfels, int(p[1]), "loss %.6f %.6f" % (float(loss), g["losses"][k[0]]), "xyz %.4f %.4f" % (float(s.get_xyz.detach()[a].abs().sum()), p[4]),
          "op %.4f %.4f" % (float(s.get_opacity.detach()[a].sum()), p[5]), "nd %.5f %.5f" % (float(tr.deform.nodes.detach()[tr.deform.live_nodes].abs().sum()), p[6]), flush=True)
    k[0] += 1
    return loss


fit_mod.run_iteration = ri
try:
    t._run(torch.device("cuda:0"), None, fused=True, strict=False, masks=os.environ.get("MASKS") == "1")
except AssertionError:
    pass
