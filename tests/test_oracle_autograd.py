"""Pins the CPU oracle: its fp64 build must agree with torch.autograd (fp64) of an independent
dense re-derivation of the forward (tests/dense_torch_ref.py).  This is the substitute for the
reference's missing golden vectors."""
import numpy as np
import pytest
import torch

from dense_torch_ref import dense_render
from scene_utils import oracle_from_case, small_case


def _run(case, precomp=False, seed=5):
    dt = torch.float64
    case = {k: (v.to(dt) if torch.is_tensor(v) else v) for k, v in case.items()}
    # exact unit quaternions in fp64: the reference's quaternion vjp is w.r.t. the normalised
    # quaternion (auxiliary.h:213-257), which equals the autograd gradient only for |q| == 1
    case["rotations"] = torch.nn.functional.normalize(case["rotations"], dim=-1)
    leaves = {k: case[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    H, W = case["image_height"], case["image_width"]
    kw = {}
    if precomp:
        g = torch.Generator().manual_seed(11)
        cp = torch.rand(leaves["means3D"].shape[0], 3, generator=g, dtype=dt).requires_grad_(True)
        kw["colors_precomp"] = cp
    else:
        kw["shs"] = leaves["shs"]
        kw["sh_degree"] = case["sh_degree"]
    color, allmap, radii, transMat = dense_render(
        leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"].reshape(-1),
        case["viewmatrix"].to(dt), case["campos"].to(dt), case["bg"].to(dt), case["tanfovx"], case["tanfovy"], H, W, **kw)
    g = torch.Generator().manual_seed(seed)
    gc = torch.randn(3, H, W, generator=g, dtype=dt)
    go = torch.randn(8, H, W, generator=g, dtype=dt)
    (color * gc).sum().add((allmap * go).sum()).backward()

    orc = oracle_from_case(case, dtype=np.float64,
                           colors_precomp=kw["colors_precomp"].detach().numpy() if precomp else None)
    og = orc.backward(gc.numpy(), go.numpy())
    return dict(color=color, allmap=allmap, radii=radii, transMat=transMat, leaves=leaves, cp=kw.get("colors_precomp")), orc, og


def _close(a, b, tol=1e-9, name=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol * scale, "%s: max abs err %.3e (scale %.3e)" % (name, err, scale)


@pytest.mark.parametrize("cfg", [
    dict(P=160, H=40, W=36, seed=0, view=3, sh_degree=3, bg=(0.0, 0.0, 0.0)),
    dict(P=120, H=32, W=48, seed=1, view=6, sh_degree=1, bg=(1.0, 1.0, 1.0), scale_mul=2.0),
    dict(P=90, H=33, W=17, seed=2, view=0, sh_degree=0, bg=(0.3, 0.6, 0.1), scale_mul=3.0, radius=2.0),
])
def test_oracle_f64_matches_autograd(cfg):
    case = small_case(**cfg)
    ref, orc, og = _run(case)
    assert orc.num_rendered > 0
    _close(orc.color, ref["color"].detach().numpy(), name="color")
    _close(orc.allmap, ref["allmap"].detach().numpy(), name="allmap")
    assert np.array_equal(orc.radii, ref["radii"].numpy())
    L = ref["leaves"]
    _close(og["dL_dmeans3D"], L["means3D"].grad.numpy(), name="dL_dmeans3D")
    _close(og["dL_dscales"], L["scales"].grad.numpy(), name="dL_dscales")
    _close(og["dL_drotations"], L["rotations"].grad.numpy(), name="dL_drotations")  # unit quaternions
    _close(og["dL_dopacity"], L["opacities"].grad.numpy(), name="dL_dopacity")
    _close(og["dL_dsh"], L["shs"].grad.numpy(), name="dL_dsh")
    # dL_dtransMat after the AABB chain == total autograd gradient on transMat
    vis = orc.radii > 0
    tg = ref["transMat"].grad.numpy()
    _close(og["dL_dtransMat"][vis], tg[vis], name="dL_dtransMat")
    # densification signal, backward.cu:645-648
    T = ref["transMat"].detach().numpy()
    W, H = case["image_width"], case["image_height"]
    hack = np.stack([tg[:, 2] * T[:, 8] * (W / 2.0), tg[:, 5] * T[:, 8] * (H / 2.0)], -1)
    _close(og["dL_dmeans2D"][vis, :2], hack[vis], name="dL_dmeans2D")
    assert np.all(og["dL_dmeans2D"][:, 2] == 0)


def test_oracle_f64_matches_autograd_precomputed_colors():
    case = small_case(P=100, H=32, W=32, seed=3, view=2, scale_mul=2.0)
    ref, orc, og = _run(case, precomp=True)
    _close(orc.color, ref["color"].detach().numpy(), name="color")
    _close(og["dL_dcolors"], ref["cp"].grad.numpy(), name="dL_dcolors")
    _close(og["dL_dmeans3D"], ref["leaves"]["means3D"].grad.numpy(), name="dL_dmeans3D")


def test_oracle_f32_close_to_f64():
    case = small_case(P=300, H=48, W=64, seed=4, view=5, scale_mul=1.5)
    o32 = oracle_from_case(case, dtype=np.float32)
    o64 = oracle_from_case({k: (v.double() if torch.is_tensor(v) else v) for k, v in case.items()}, dtype=np.float64)
    assert o32.num_rendered == o64.num_rendered
    assert np.array_equal(o32.radii, o64.radii)
    assert np.abs(o32.color - o64.color).max() < 2e-5
    assert np.abs(o32.allmap - o64.allmap).max() < 2e-4
