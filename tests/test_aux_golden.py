"""SH colour evaluation and camera matrices against golden vectors produced by IMPORTING the reference's utils/sh_utils.py
and utils/graphics_utils.py (tests/golden/make_aux_golden.py): pins the colour part of the CPU oracle (and, through the
-m gpu stage tests, of the HIP preprocess kernel) and dgs_amd.cameras to the reference's own code."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_aux_golden import CAMS, CASE, rot_y  # noqa: E402  (input parameters only; the reference is not imported)

from dgs_amd import cameras  # noqa: E402
from scene_utils import oracle_from_case, small_case  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "aux_golden.npz"))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_colours_match_reference_eval_sh(deg):
    case = small_case(**dict(CASE, sh_degree=deg))
    orc = oracle_from_case(case)
    vis = orc.radii > 0
    assert vis.sum() > 50
    want = np.maximum(G["sh_colors%d" % deg] + 0.5, 0.0)   # forward.cu:66-71: + 0.5, clamped at 0
    got = orc.field("rgb")
    assert np.abs(got[vis] - want[vis]).max() <= 2e-6
    raw = (G["sh_colors%d" % deg] + 0.5)[vis]
    differs = orc.field("clamped")[vis].astype(bool) != (raw < 0.0)   # the clamp flags the backward uses
    assert not differs.any() or np.abs(raw[differs]).max() < 1e-6


@pytest.mark.parametrize("i", [0, 1])
def test_camera_matrices_match_reference_graphics_utils(i):
    c = CAMS[i]
    R, t = rot_y(c["theta"]), np.array(c["t"])
    assert np.abs(cameras.world_to_view(R, t) - G["w2v%d" % i]).max() <= 1e-7
    P = cameras.projection_matrix(c["znear"], c["zfar"], c["fovx"], c["fovy"]).numpy()
    assert np.abs(P - G["proj%d" % i]).max() <= 1e-6 * np.abs(G["proj%d" % i]).max()


@pytest.mark.gpu
def test_hip_preprocess_colours_match_reference_eval_sh():
    from gpu_utils import run_hip_raw
    case = small_case(**CASE)
    raw = run_hip_raw(case)
    vis = raw["radii"] > 0
    want = np.maximum(G["sh_colors3"] + 0.5, 0.0)
    got = raw["rec"][:, 15:18]
    assert np.abs(got[vis] - want[vis]).max() <= 2e-6


def test_lr_schedules_match_reference():
    """expon_lr + the Trainer's schedule constants against the imported get_expon_lr_func with the trainer's arguments."""
    from dgs_amd.train import Trainer, expon_lr
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aux_golden.npz"))
    p0, p1, pn = Trainer.SCHED_POSITION
    d0, d1, dn = Trainer.SCHED_DEFORM
    for k, want_p, want_d in zip(g["lr_steps"], g["lr_position"], g["lr_deform"]):
        assert abs(expon_lr(int(k), p0 * 5, p1 * 5, pn) - want_p) <= 1e-12 * want_p
        assert abs(expon_lr(int(k), d0, d1, dn) - want_d) <= 1e-12 * want_d
