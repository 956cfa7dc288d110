"""Diagnostic (GPU box): where do the largest HIP-vs-oracle differences of a big render sit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gpu_utils import run_hip, run_hip_raw
from scene_utils import oracle_from_case, small_case

P, H, W, view = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1_000_000, 1600, 1600, 41)
case = small_case(P=P, H=H, W=W, seed=0, view=view, n_views=64)
a = run_hip(case, debug=False)
orc = oracle_from_case(case)
err = np.abs(a["allmap"].astype(np.float64) - orc.allmap)
for c in range(8):
    e = err[c]
    iy, ix = np.unravel_index(np.argmax(e), e.shape)
    print("ch %d: max %.3e at (%d,%d) hip %.6f orc %.6f | >1e-3: %d  >1e-2: %d  >1e-1: %d" % (
        c, e.max(), iy, ix, a["allmap"][c, iy, ix], orc.allmap[c, iy, ix], (e > 1e-3).sum(), (e > 1e-2).sum(), (e > 1e-1).sum()))
ec = np.abs(a["color"].astype(np.float64) - orc.color)
print("color max %.3e, >1e-3: %d" % (ec.max(), (ec > 1e-3).sum()))
c = int(np.argmax(err.reshape(8, -1).max(1)))
iy, ix = np.unravel_index(np.argmax(err[c]), err[c].shape)
print("worst pixel (%d,%d): hip allmap %s" % (iy, ix, a["allmap"][:, iy, ix]))
print("                      orc allmap %s" % (orc.allmap[:, iy, ix],))
nc = orc.field("n_contrib")
print("oracle n_contrib (last, median):", nc[:, iy, ix])
raw = run_hip_raw(case)
print("hip    n_contrib (last, median):", raw["n_contrib"][:, iy, ix], " final_T hip", raw["final_T"][:, iy, ix], "orc", orc.field("final_T")[:, iy, ix])
print("num_rendered hip %d oracle %d" % (raw["R"], orc.num_rendered))
