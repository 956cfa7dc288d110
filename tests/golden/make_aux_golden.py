"""Generates tests/golden/aux_golden.npz by IMPORTING the reference's utils/sh_utils.py (eval_sh) and
utils/graphics_utils.py (getWorld2View2, getProjectionMatrix) in the build container:
  * sh_colors[d]: eval_sh(d, shs, dirs) for degrees 0..3 on the surfels of tests' small_case(P=300, seed=11) seen from its
    camera -- the view-dependent colour the rasterizer's preprocess must reproduce (forward.cu:20-71 is the same formula,
    plus 0.5 and the clamp at 0);
  * world_view / projection matrices for two parameter sets;
  * lr_position / lr_deform: utils/general_utils.py get_expon_lr_func with the trainer's arguments at a few iterations.
Run from the repo root:  python tests/golden/make_aux_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)

CASE = dict(P=300, H=48, W=56, seed=11, view=2, sh_degree=3)
CAMS = [dict(theta=0.7, t=(0.3, -0.2, 4.0), znear=0.01, zfar=100.0, fovx=0.9, fovy=0.7),
        dict(theta=-1.9, t=(-1.0, 0.4, 2.5), znear=0.2, zfar=50.0, fovx=0.5, fovy=1.1)]


def rot_y(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, "utils", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    from scene_utils import small_case
    sh_utils, gu = load("sh_utils"), load("graphics_utils")
    case = small_case(**CASE)
    dirs = case["means3D"] - case["campos"][None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {}
    for d in range(4):
        # the reference evaluates on [P,3,(deg+1)^2] (gaussian_renderer/__init__.py: shs_view = features.transpose(1, 2))
        out["sh_colors%d" % d] = sh_utils.eval_sh(d, case["shs"].transpose(1, 2), dirs).numpy().astype(np.float32)
    for i, c in enumerate(CAMS):
        R, t = rot_y(c["theta"]), np.array(c["t"])
        out["w2v%d" % i] = gu.getWorld2View2(R, t).astype(np.float32)
        out["proj%d" % i] = gu.getProjectionMatrix(c["znear"], c["zfar"], c["fovx"], c["fovy"]).numpy().astype(np.float32)
    # learning-rate schedules exactly as the trainer builds them (gaussian_model.py:203, deform_model.py:37)
    gen = load("general_utils")
    steps = np.array([0, 1, 2, 100, 2999, 15000, 29999, 30000, 39999, 40000, 60000])
    pos = gen.get_expon_lr_func(lr_init=0.00016 * 5, lr_final=0.0000016 * 5, lr_delay_mult=0.01, max_steps=30000)
    dfm = gen.get_expon_lr_func(lr_init=0.00016 * 5 * 1.0, lr_final=0.0000016 * 1.0, lr_delay_mult=0.01, max_steps=40000)
    out["lr_steps"] = steps
    out["lr_position"] = np.array([pos(int(k)) for k in steps], np.float64)
    out["lr_deform"] = np.array([dfm(int(k)) for k in steps], np.float64)
    np.savez_compressed(os.path.join(HERE, "aux_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
