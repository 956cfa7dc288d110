"""Camera construction with the reference's matrix conventions.

Restates (does not import) scene/cameras.py:49-59, utils/graphics_utils.py:34-77 and
utils/pose_utils.py:5-21,63-68 of the reference: matrices are stored transposed (row-vector
convention), so ``world_view_transform.flatten()`` is the column-major W2C the kernels index
with translation at [12..14] (cuda_rasterizer/forward.cu:84).
"""
import math
from typing import NamedTuple

import numpy as np
import torch


def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """Camera-to-world matrix on a sphere (utils/pose_utils.py:63-68), float64 numpy."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float64)
    rphi = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]], np.float64)
    rth = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]], np.float64)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float64)
    return flip @ rth @ rphi @ trans


def c2w_to_RT(c2w: np.ndarray):
    """NeRF-style c2w -> (R, T) exactly as scene/dataset_readers.py:293-296 / render.py:145-148."""
    m = np.linalg.inv(c2w)
    R = -np.transpose(m[:3, :3])
    R[:, 0] = -R[:, 0]
    T = -m[:3, 3]
    return R, T


def world_to_view(R: np.ndarray, T: np.ndarray) -> np.ndarray:
    """utils/graphics_utils.py:43-54 with translate=0, scale=1."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    return np.float32(Rt)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """utils/graphics_utils.py:57-77."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera(NamedTuple):
    """The attributes gaussian_renderer.render() reads from a viewpoint camera
    (gaussian_renderer/__init__.py:56-72,176 and utils/point_utils.py:10-25)."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] = W2C^T
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    fid: torch.Tensor                    # [1] time in [0,1]

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.fid.to(device))


def make_camera(c2w: np.ndarray, fovx: float, fovy: float, width: int, height: int, fid: float,
                znear: float = 0.01, zfar: float = 100.0) -> Camera:
    R, T = c2w_to_RT(c2w)
    wvt = torch.tensor(world_to_view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Camera(int(height), int(width), float(fovx), float(fovy), wvt, full, center,
                  torch.tensor([float(fid)], dtype=torch.float32))


def orbit_cameras(n_views: int, width: int, height: int, fov: float = 0.6911, radius: float = 4.0,
                  phi_deg: float = -30.0):
    """The V-view orbit of SURVEY.md section 8(d): theta_k = -180 + 360 k / V, fid_k = k / (V-1)."""
    cams = []
    for k in range(n_views):
        theta = -180.0 + 360.0 * k / n_views
        fid = k / max(n_views - 1, 1)
        cams.append(make_camera(pose_spherical(theta, phi_deg, radius), fov, fov, width, height, fid))
    return cams
