"""Where does the gradient error at full size come from? (development aid, GPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from gpu_utils import rel_l2, run_hip
from scene_utils import oracle_from_case, small_case
P = int(os.environ.get("P", 200000)); S = int(os.environ.get("S", 800))
case = small_case(P=P, H=S, W=S, seed=0, view=5, n_views=64)
g = np.random.default_rng(3)
gc = g.standard_normal((3, S, S)).astype(np.float32); go = g.standard_normal((8, S, S)).astype(np.float32)
a = run_hip(case, gc, go, debug=False)
a2 = run_hip(case, gc, go, debug=False)
orc = oracle_from_case(case); og = orc.backward(gc, go)
print("color max err", np.abs(a["color"] - orc.color).max(), "allmap max err", np.abs(a["allmap"] - orc.allmap).max())
for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"):
    e = (a[k].astype(np.float64) - og[k]).reshape(P, -1)
    n = np.linalg.norm(e, axis=1)
    tot = np.linalg.norm(e)
    srt = np.sort(n)[::-1]
    print("%-14s relL2 %.3e  run-to-run relL2 %.3e  top10 share of err^2 %.3f  top100 %.3f  |og| %.3e" % (
        k, rel_l2(a[k], og[k]), rel_l2(a[k], a2[k]), (srt[:10] ** 2).sum() / tot ** 2, (srt[:100] ** 2).sum() / tot ** 2, np.linalg.norm(og[k])))
