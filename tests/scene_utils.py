"""Shared helpers for the tests: small seeded scenes in the rasterizer's input convention."""
import math

import numpy as np
import torch

from dgs_amd.cameras import orbit_cameras
from dgs_amd.synthetic import activated, make_scene


def small_case(P=160, H=40, W=36, seed=0, view=3, n_views=8, sh_degree=3, bg=(0.0, 0.0, 0.0), scale_mul=1.0,
               fov=0.6911, radius=4.0, dtype=torch.float32):
    """Returns a dict of CPU tensors: the positional inputs of GaussianRasterizer + settings."""
    scene = make_scene(P, seed)
    xyz, scales, rots, opac, shs = activated(scene)
    cam = orbit_cameras(n_views, W, H, fov=fov, radius=radius)[view]
    d = dict(
        means3D=xyz.to(dtype), scales=(scales * scale_mul).to(dtype), rotations=rots.to(dtype), opacities=opac.to(dtype),
        shs=shs.to(dtype), sh_degree=sh_degree,
        viewmatrix=cam.world_view_transform.to(dtype), projmatrix=cam.full_proj_transform.to(dtype),
        campos=cam.camera_center.to(dtype), bg=torch.tensor(bg, dtype=dtype),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), image_height=H, image_width=W,
    )
    return d


def oracle_from_case(case, dtype=np.float32, colors_precomp=None):
    from oracle.surfel_oracle import OracleRaster
    kw = dict(
        means3D=case["means3D"].numpy(), opacities=case["opacities"].numpy(), scales=case["scales"].numpy(),
        rotations=case["rotations"].numpy(), viewmatrix=case["viewmatrix"].numpy(), projmatrix=case["projmatrix"].numpy(),
        campos=case["campos"].numpy(), bg=case["bg"].numpy(), tanfovx=case["tanfovx"], tanfovy=case["tanfovy"],
        image_height=case["image_height"], image_width=case["image_width"], dtype=dtype)
    if colors_precomp is not None:
        kw["colors_precomp"] = colors_precomp
    else:
        kw["shs"] = case["shs"].numpy()
        kw["sh_degree"] = case["sh_degree"]
    return OracleRaster(**kw)
