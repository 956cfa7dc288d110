"""Synthetic scene S(P, H, W, seed) of SURVEY.md section 8(d).

Stands in for the D-NeRF data (absent from the container).  Generated on the CPU with a seeded
torch.Generator so that the MI355X path and the CPU oracle see bit-identical inputs.  Mirrors the
reference initialisation where one exists: xyz ~ U[-1.3, 1.3]^3 (scene/dataset_readers.py:385),
feature = -1e-2 (scene/gaussian_model.py:177), SH layout [P,16,3] (gaussian_model.py:103-107).
"""
import math
from typing import NamedTuple

import torch


class SurfelScene(NamedTuple):
    xyz: torch.Tensor            # [P,3]
    log_scale: torch.Tensor      # [P,2]  (pre-activation, exp() applied by the renderer)
    rotation: torch.Tensor       # [P,4]  (r,x,y,z), pre-normalisation
    opacity_logit: torch.Tensor  # [P,1]
    f_dc: torch.Tensor           # [P,1,3]
    f_rest: torch.Tensor         # [P,15,3]
    feature: torch.Tensor        # [P,8] hyper coordinates


def make_scene(P: int, seed: int = 0) -> SurfelScene:
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2.0 - 1.0) * 1.3
    base = math.log(0.5 * (17.576 / max(P, 1)) ** (1.0 / 3.0))
    log_scale = base + 0.3 * torch.randn(P, 2, generator=g)
    rot = torch.randn(P, 4, generator=g)
    opacity_logit = 2.0 * torch.randn(P, 1, generator=g)
    f_dc = 0.5 * torch.randn(P, 1, 3, generator=g)
    f_rest = 0.05 * torch.randn(P, 15, 3, generator=g)
    feature = torch.full((P, 8), -1e-2)
    return SurfelScene(xyz, log_scale, rot, opacity_logit, f_dc, f_rest, feature)


def activated(scene: SurfelScene):
    """Inputs as the rasterizer sees them (gaussian_renderer/__init__.py:83-122 with zero deformation)."""
    scales = torch.exp(scene.log_scale)
    rotations = torch.nn.functional.normalize(scene.rotation, dim=-1)
    opac = torch.sigmoid(scene.opacity_logit)
    shs = torch.cat([scene.f_dc, scene.f_rest], dim=1).contiguous()
    return scene.xyz.contiguous(), scales.contiguous(), rotations.contiguous(), opac.contiguous(), shs


def target_image(H: int, W: int, seed: int = 1) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.rand(3, H, W, generator=g)
