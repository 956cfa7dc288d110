"""Loss / depth-to-normal code against golden vectors produced by IMPORTING the reference's utils/loss_utils.py and
utils/point_utils.py (tests/golden/make_loss_golden.py).  CPU: the PyTorch formulations of dgs_amd (what the CPU baseline
runs).  GPU: the fused kernels of libdgs_train_ops.so."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_loss_golden import View, depth_map, images  # noqa: E402  (input formulas only; the reference is not imported)

from dgs_amd import losses  # noqa: E402
from dgs_amd import render as render_mod  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "loss_golden.npz"))


def _camera(dev="cpu"):
    v = View()
    return SimpleNamespace(image_height=v.image_height, image_width=v.image_width, FoVx=v.FoVx, FoVy=v.FoVy,
                           world_view_transform=v.world_view_transform.to(dev))


def test_l1_and_ssim_match_reference_golden():
    a, b = images()
    assert abs(float(losses.l1_loss(a, b)) - float(G["l1"])) <= 1e-7
    assert abs(float(losses.ssim_torch(a, b)) - float(G["ssim"])) <= 2e-6
    assert abs(float(losses.ssim_torch(a, a)) - float(G["ssim_self"])) <= 2e-6


def test_depth_to_normal_matches_reference_golden():
    cam = _camera()
    normal, points = render_mod.depth_to_normal(cam, depth_map())
    assert np.abs(points.numpy() - G["points"]).max() <= 2e-6 * np.abs(G["points"]).max()
    assert np.abs(normal.numpy() - G["normal"]).max() <= 2e-5


@pytest.mark.gpu
def test_fused_ssim_kernel_matches_reference_golden():
    from dgs_amd import _ops
    a, b = (t.cuda() for t in images())
    assert abs(float(_ops.fused_ssim(a, b)) - float(G["ssim"])) <= 2e-6
    assert abs(float(_ops.fused_ssim(a, a)) - float(G["ssim_self"])) <= 2e-6


@pytest.mark.gpu
def test_fused_regulariser_uses_the_reference_normals():
    """lambda_normal * mean(1 - <rend_normal_world, surf_normal>) from the kernel == the same expression built from the
    surf normals the REFERENCE's depth_to_normal returned for this depth map and camera (alpha = 1, no distortion)."""
    from dgs_amd import _ops
    cam = _camera("cuda")
    depth = depth_map().cuda()
    H, W = depth.shape[1:]
    gen = torch.Generator(device="cuda").manual_seed(3)
    n_view = torch.nn.functional.normalize(torch.randn(3, H, W, device="cuda", generator=gen), dim=0)
    allmap = torch.zeros(8, H, W, device="cuda")
    allmap[1] = 1.0
    allmap[2:5] = n_view
    allmap[5] = depth[0]
    rays_d, rays_o = render_mod.camera_rays(cam, "cuda")
    got = float(_ops.fused_reg_loss(allmap, rays_d, rays_o, cam.world_view_transform, 1.0, 0.0))
    n_world = (n_view.permute(1, 2, 0) @ cam.world_view_transform[:3, :3].T).permute(2, 0, 1)   # gaussian_renderer/__init__.py:177
    surf = torch.from_numpy(G["normal"]).cuda().permute(2, 0, 1)
    want = float((1 - (n_world * surf).sum(0)).mean())
    assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (got, want)
