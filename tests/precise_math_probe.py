"""Run by tests/test_precise_math_gpu.py in a subprocess with DGS_SURFEL_LIB pointing at one build of the library:
prints one JSON line with the distance of the HIP path to the fp32 and fp64 oracles on the parity scenes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

from gpu_utils import median_flips, rel_l2, run_hip, hip_median_contrib
from scene_utils import oracle_from_case, small_case

SCENES = {
    "small": dict(P=2000, H=96, W=112, seed=2, view=3, scale_mul=1.5),
    "mid20k": dict(P=20000, H=256, W=256, seed=21, view=11, n_views=16, scale_mul=1.5),
}
KEYS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh")
out = {}
for name, cfg in SCENES.items():
    case = small_case(**cfg)
    g = np.random.default_rng(1)
    H, W = case["image_height"], case["image_width"]
    gc, go = g.standard_normal((3, H, W)).astype(np.float32), g.standard_normal((8, H, W)).astype(np.float32)
    o32, o64 = oracle_from_case(case), oracle_from_case(case, dtype=np.float64)
    # pixels whose (discontinuous) median pick differs from the oracle's -- proven ties at T = 0.5 -- carry no cotangent on the two
    # median channels: the derivative of another entry's depth is not comparable (gpu_utils.median_flips)
    flips = median_flips(hip_median_contrib(case), o32) | median_flips(o64.field("n_contrib")[1], o32)
    go[5][flips] = 0.0
    go[7][flips] = 0.0
    hip = run_hip(case, gc, go, debug=False)
    g32, g64 = o32.backward(gc, go), o64.backward(gc.astype(np.float64), go.astype(np.float64))
    out[name] = {"median_flips": int(flips.sum()), "color_max": float(np.abs(hip["color"] - o32.color).max()), "color_vs_f64_max": float(np.abs(hip["color"] - o64.color).max()),
                 "grads_vs_f32": {k: rel_l2(hip[k], g32[k]) for k in KEYS}, "grads_vs_f64": {k: rel_l2(hip[k], g64[k]) for k in KEYS},
                 "oracle_f32_vs_f64": {k: rel_l2(g32[k], g64[k]) for k in KEYS}}
print("PROBE " + json.dumps(out))
