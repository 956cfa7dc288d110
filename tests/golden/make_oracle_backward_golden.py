"""Freezes the ORACLE's own forward and backward (oracle/surfel_oracle.c, the all-double build) on four small scenes:
tests/golden/oracle_backward.npz.  The parity tests compare the kernels with the oracle as it is TODAY; without a frozen copy an
edit that moved oracle and kernels together (a shared misreading of backward.cu) would stay green.  tests/test_oracle_golden.py checks
both oracle builds against this file on the CPU.

Scenes: the seeded generator of the tests (tests/scene_utils.small_case: make_scene + orbit camera), four configurations that
between them cover SH degree 0 / 1 / 3, black / white / coloured background, image sizes off the 16-pixel grid, a camera close enough
for surfels behind the near plane, and precomputed colours.  Cotangents: standard normal, numpy default_rng(3) / (5).
Cotangents are stored as float32 (what every consumer feeds), images rounded to float32, gradients as float64; the scenes
themselves are regenerated from their seeds.

    python tests/golden/make_oracle_backward_golden.py        (build container; needs only gcc + numpy)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dynamic-2dgs_amd")]

SCENES = [
    dict(P=300, H=40, W=36, seed=41, view=1, scale_mul=2.0, sh_degree=3),
    dict(P=400, H=33, W=47, seed=42, view=5, scale_mul=1.2, sh_degree=1, bg=(1.0, 1.0, 1.0)),
    dict(P=250, H=48, W=64, seed=43, view=2, scale_mul=1.0, sh_degree=0, bg=(0.2, 0.5, 0.9), radius=2.5),
    dict(P=350, H=64, W=40, seed=44, view=6, scale_mul=3.0, sh_degree=2, precomp=True),
]
GRADS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh", "dL_dcolors")


def run(cfg, dtype):
    from scene_utils import oracle_from_case, small_case
    cfg = dict(cfg)
    precomp = cfg.pop("precomp", False)
    case = small_case(**cfg)
    H, W, P = case["image_height"], case["image_width"], case["means3D"].shape[0]
    g = np.random.default_rng(3)
    gc, go = g.standard_normal((3, H, W)).astype(np.float32), g.standard_normal((8, H, W)).astype(np.float32)
    cp = np.random.default_rng(5).random((P, 3)).astype(np.float32) if precomp else None
    orc = oracle_from_case(case, dtype=dtype, colors_precomp=cp)
    og = orc.backward(gc, go)
    out = dict(color=orc.color, allmap=orc.allmap, radii=orc.radii, n_contrib=orc.field("n_contrib"), num_rendered=np.int64(orc.num_rendered))
    for k in GRADS:
        if k in og and (k != "dL_dsh" or not precomp) and (k != "dL_dcolors" or precomp):
            out[k] = og[k]
    return case, gc, go, cp, out


if __name__ == "__main__":
    blob = {}
    for i, cfg in enumerate(SCENES):
        case, gc, go, cp, out = run(cfg, np.float64)
        for k, v in out.items():
            a = np.asarray(v)
            blob["s%d_%s" % (i, k)] = a if a.dtype.kind != "f" else a.astype(np.float32 if k in ("color", "allmap") else np.float64)   # images rounded to f32, gradients in full
        blob["s%d_gc" % i], blob["s%d_go" % i] = gc, go
        print("scene %d: R=%d visible=%d |dL_dmeans3D|max=%.3e" % (i, out["num_rendered"], int((out["radii"] > 0).sum()), float(np.abs(out["dL_dmeans3D"]).max())))
    np.savez_compressed(os.path.join(HERE, "oracle_backward.npz"), **blob)
    print("wrote", os.path.join(HERE, "oracle_backward.npz"), os.path.getsize(os.path.join(HERE, "oracle_backward.npz")), "bytes")
