"""Generates tests/golden/densify_golden.npz by IMPORTING the reference's scene/gaussian_model.py in the build container
and running its densification on CPU:  training_setup -> three Adam steps with random gradients (so the optimiser holds
non-trivial moments) -> add_densification_stats x3 -> densify_and_prune(0.0002, 0.01, extent, 20) -> reset_opacity.

The module needs stand-ins for two absent packages that the exercised code never calls (`plyfile`, `simple_knn._C`), and
its hard-wired device="cuda" factory calls are redirected to the CPU while it runs.  The fixture stores inputs, the
standard-normal draws `torch.normal` consumed in densify_and_split (so the test does not depend on the RNG stream), and
the resulting parameters / Adam moments / statistics.
Run from the repo root:  python tests/golden/make_densify_golden.py
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

P, SEED, EXTENT = 240, 5, 4.0
NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "feature"]


@contextlib.contextmanager
def cuda_to_cpu():
    """torch.zeros(..., device="cuda") etc. -> CPU for the duration of the reference calls."""
    saved = {}
    for fn in ("zeros", "ones", "empty", "tensor", "zeros_like", "ones_like"):
        orig = getattr(torch, fn)
        saved[fn] = orig

        def wrap(*a, __orig=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, fn, wrap)
    cuda_saved = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    try:
        yield
    finally:
        for fn, orig in saved.items():
            setattr(torch, fn, orig)
        torch.cuda.empty_cache = cuda_saved


def import_reference_model():
    ply = types.ModuleType("plyfile"); ply.PlyData = ply.PlyElement = object
    knn = types.ModuleType("simple_knn"); knn_c = types.ModuleType("simple_knn._C"); knn_c.distCUDA2 = None
    sys.modules.update({"plyfile": ply, "simple_knn": knn, "simple_knn._C": knn_c})
    sys.path.insert(0, REF)
    import importlib.util
    # by file: importing the `scene` package would pull in the dataset readers (imageio, cv2 ...)
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene", "gaussian_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.GaussianModel


def main():
    GaussianModel = import_reference_model()
    g = torch.Generator().manual_seed(SEED)
    rn = lambda *s: torch.randn(*s, generator=g)
    init = {
        "xyz": rn(P, 3) * 1.3,
        "f_dc": rn(P, 1, 3) * 0.5,
        "f_rest": rn(P, 15, 3) * 0.05,
        "opacity": rn(P, 1) * 2.0,
        # around percent_dense * extent = 0.04 so that both clone and split fire; a few beyond 0.1 * extent (pruned)
        "scaling": torch.log(0.04 * torch.exp(rn(P, 2) * 0.8)),
        "rotation": rn(P, 4),
        "feature": rn(P, 8) * 0.01,
    }
    gm = GaussianModel(3, fea_dim=8, with_motion_mask=False)
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation", "feature": "feature"}
    for n in NAMES:
        setattr(gm, attr[n], torch.nn.Parameter(init[n].clone()))
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    out = {}
    with cuda_to_cpu():
        gm.training_setup(args)
        gm.max_radii2D = torch.zeros(P)
        for it in range(3):
            for n in NAMES:
                p = getattr(gm, attr[n])
                p.grad = rn(*p.shape) * 1e-3
            gm.optimizer.step()
            gm.optimizer.zero_grad(set_to_none=True)
        # state before densification
        for n in NAMES:
            p = getattr(gm, attr[n])
            out["pre_" + n] = p.detach().numpy().copy()
            out["pre_m_" + n] = gm.optimizer.state[p]["exp_avg"].numpy().copy()
            out["pre_v_" + n] = gm.optimizer.state[p]["exp_avg_sq"].numpy().copy()
        # three views of statistics (gaussian_model.py:484-486)
        for it in range(3):
            vis = torch.rand(P, generator=g) < 0.7
            vsp = types.SimpleNamespace(grad=rn(P, 3) * 3e-4)
            gm.add_densification_stats(vsp, vis)
            gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], torch.randint(0, 40, (int(vis.sum()),), generator=g).float())
            out["vis%d" % it] = vis.numpy()
            out["vsp_grad%d" % it] = vsp.grad.numpy()
        out["accum"] = gm.xyz_gradient_accum.numpy().copy()
        out["denom"] = gm.denom.numpy().copy()
        out["max_radii2D"] = gm.max_radii2D.numpy().copy()
        # record the standard-normal draws of densify_and_split: torch.normal(mean, std) == mean + std * z
        drawn = []
        orig_normal = torch.normal

        def normal(mean, std):
            z = torch.randn(mean.shape, generator=g)
            drawn.append(z.clone())
            return mean + std * z
        torch.normal = normal
        try:
            gm.densify_and_prune(0.0002, 0.01, EXTENT, 20)
        finally:
            torch.normal = orig_normal
        assert len(drawn) == 1
        out["noise"] = drawn[0].numpy()
        for n in NAMES:
            p = getattr(gm, attr[n])
            out["post_" + n] = p.detach().numpy().copy()
            out["post_m_" + n] = gm.optimizer.state[p]["exp_avg"].numpy().copy()
            out["post_v_" + n] = gm.optimizer.state[p]["exp_avg_sq"].numpy().copy()
        out["post_accum"] = gm.xyz_gradient_accum.numpy().copy()
        out["post_denom"] = gm.denom.numpy().copy()
        out["post_max_radii2D"] = gm.max_radii2D.numpy().copy()
        gm.reset_opacity()
        out["reset_opacity"] = gm._opacity.detach().numpy().copy()
        out["reset_m_opacity"] = gm.optimizer.state[gm._opacity]["exp_avg"].numpy().copy()
        out["reset_v_opacity"] = gm.optimizer.state[gm._opacity]["exp_avg_sq"].numpy().copy()
    out["extent"] = np.float32(EXTENT)
    np.savez_compressed(os.path.join(HERE, "densify_golden.npz"), **out)
    print("P %d -> %d ; noise rows %d" % (P, out["post_xyz"].shape[0], out["noise"].shape[0]))


if __name__ == "__main__":
    main()
