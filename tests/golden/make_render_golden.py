"""Generates tests/golden/render_golden.npz by IMPORTING the reference's gaussian_renderer/__init__.py (render()) in the
build container and running it on the CPU with the rasterizer replaced by the repo's CPU oracle (tests/oracle_raster_op.py
registered as `diff_surfel_rasterization`): the golden dict is the REFERENCE's wiring and allmap post-processing applied to
the oracle's rasterizer outputs, which is what dgs_amd.render.render restates.
Stubs: cv2, matplotlib, scene.gaussian_model (type annotation only); `.cuda()` and device="cuda" are patched away.
Run from the repo root:  python tests/golden/make_render_golden.py
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)

CASE = dict(P=700, H=48, W=56, seed=5, view=3)


def inputs():
    """Model, camera and deformation inputs shared with the test (deterministic formulas)."""
    from dgs_amd.cameras import orbit_cameras
    from dgs_amd.model import SurfelModel
    from dgs_amd.synthetic import make_scene
    torch.manual_seed(0)
    pc = SurfelModel(make_scene(CASE["P"], seed=CASE["seed"]))
    with torch.no_grad():
        pc._scaling += 0.9          # splats of a few pixels at this image size
    cam = orbit_cameras(8, CASE["W"], CASE["H"])[CASE["view"]]
    i = torch.arange(CASE["P"], dtype=torch.float64)
    d_xyz = torch.stack([0.02 * torch.sin(0.7 * i + k) for k in range(3)], 1).float()
    d_rot = torch.stack([0.05 * torch.cos(0.3 * i + k) for k in range(4)], 1).float()
    d_scale = torch.stack([0.002 * torch.sin(0.11 * i + k) for k in range(2)], 1).float()
    return pc, cam, torch.zeros(3), d_xyz, d_rot, d_scale


def extra_inputs():
    """d_opacity [P,1] and d_color [P,3] (the reference's pred_opacity / pred_color outputs; None in its default configuration)."""
    i = torch.arange(CASE["P"], dtype=torch.float64)
    d_opacity = (0.1 * torch.sin(0.37 * i))[:, None].float()
    d_color = torch.stack([0.3 * torch.cos(0.21 * i + k) for k in range(3)], 1).float()
    return d_opacity, d_color


def import_reference_render():
    import diff_surfel_rasterization as product   # settings tuple of the product package (no native code needed for it)
    import oracle_raster_op
    fake = types.ModuleType("diff_surfel_rasterization")
    fake.GaussianRasterizationSettings = product.GaussianRasterizationSettings
    fake.GaussianRasterizer = oracle_raster_op.OracleRasterizer
    saved = {k: sys.modules.get(k) for k in ("diff_surfel_rasterization", "utils", "scene", "scene.gaussian_model")}
    sys.modules["diff_surfel_rasterization"] = fake
    for name in ("cv2", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(name, types.ModuleType(name))
    scene = types.ModuleType("scene"); gm = types.ModuleType("scene.gaussian_model")
    gm.GaussianModel = object
    scene.gaussian_model = gm
    sys.modules["scene"], sys.modules["scene.gaussian_model"] = scene, gm
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fn_name in ("zeros_like", "tensor"):
        orig = getattr(torch, fn_name)

        def patched(*a, _orig=orig, **k):
            if k.get("device") == "cuda":
                k.pop("device")
            return _orig(*a, **k)
        setattr(torch, fn_name, patched)
    sys.path.insert(0, REF)
    import gaussian_renderer as ref_renderer
    return ref_renderer


def main():
    pc, cam, bg, d_xyz, d_rot, d_scale = inputs()
    ref = import_reference_render()
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=1)
    with torch.no_grad():
        out = ref.render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale)
    keys = ("render", "alpha", "rend_normal", "rend_dist", "depth", "surf_normal", "surf_point", "radii", "visibility_filter")
    arrays = {k: out[k].detach().numpy() for k in keys}
    assert set(out.keys()) == set(keys) | {"viewspace_points", "bg_color"}, sorted(out.keys())
    assert (arrays["radii"] > 0).sum() > 100
    np.savez_compressed(os.path.join(HERE, "render_golden.npz"), **arrays)
    print({k: v.shape for k, v in arrays.items()}, "visible", int((arrays["radii"] > 0).sum()))
    # second vector: the optional per-Gaussian opacity / colour offsets (gaussian_renderer/__init__.py:85,114)
    d_opacity, d_color = extra_inputs()
    with torch.no_grad():
        out2 = ref.render(cam, pc, pipe, bg, d_xyz, d_rot, d_scale, d_opacity=d_opacity, d_color=d_color)
    arrays2 = {k: out2[k].detach().numpy() for k in keys}
    assert np.abs(arrays2["render"] - arrays["render"]).max() > 1e-2
    np.savez_compressed(os.path.join(HERE, "render_golden_offsets.npz"), **arrays2)


if __name__ == "__main__":
    main()
