"""The oracle against a frozen copy of itself (tests/golden/oracle_backward.npz, written by tests/golden/make_oracle_backward_golden.py
from the all-double build): forward AND the hand-derived backward of oracle/surfel_oracle.c on four small scenes.  The GPU parity
tests compare the kernels with the oracle as it is today; this file pins the oracle, so an edit that moves both together is seen.
CPU only."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_backward_golden as mk  # noqa: E402  (scene list + the one function that runs a scene)

GOLD = np.load(os.path.join(HERE, "golden", "oracle_backward.npz"))


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("i", range(len(mk.SCENES)))
@pytest.mark.parametrize("dtype,tol_img,tol_grad", [(np.float64, 1e-6, 1e-9), (np.float32, 2e-5, 2e-3)])
def test_oracle_equals_its_frozen_outputs(i, dtype, tol_img, tol_grad):
    """f64 build: images to the f32 rounding of the file, gradients to 1e-9 (OpenMP order of the double sums only).
    f32 build (the reference's float/double mix): images 2e-5, gradients 2e-3 relative L2 -- the distance between the two builds on
    these scenes (the GPU tests hold the kernels to 2e-4 of the f32 build)."""
    case, gc, go, cp, out = mk.run(mk.SCENES[i], dtype)
    g = lambda k: GOLD["s%d_%s" % (i, k)]
    assert np.array_equal(gc, g("gc")) and np.array_equal(go, g("go")), "the cotangents of the file are not the ones generated here"
    assert np.array_equal(out["radii"], g("radii"))
    assert int(out["num_rendered"]) == int(g("num_rendered"))
    diff_nc = out["n_contrib"] != g("n_contrib")
    assert diff_nc.mean() <= (0.0 if dtype == np.float64 else 2e-3), "last / median contributor differs on %d pixels" % diff_nc.sum()
    ok = ~diff_nc.any(axis=0)   # (f32 build: a pixel whose contributor flipped on a threshold is not compared)
    assert np.abs(np.asarray(out["color"], np.float64) - g("color"))[:, ok].max() <= tol_img * max(1.0, np.abs(g("color")).max())
    assert np.abs(np.asarray(out["allmap"], np.float64) - g("allmap"))[:, ok].max() <= 10 * tol_img * max(1.0, np.abs(g("allmap")).max())
    n = 0
    for k in mk.GRADS:
        if "s%d_%s" % (i, k) in GOLD.files:
            assert _rel(out[k], g(k)) <= tol_grad, "%s: rel-L2 %.3e" % (k, _rel(out[k], g(k)))
            n += 1
    assert n == 6
