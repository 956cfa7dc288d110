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


# ---- a scene WITH AN ANSWER: hidden dynamic surfels rendered into a D-NeRF-format dataset ---------------------------------------
def _quat_from_normal(n: torch.Tensor) -> torch.Tensor:
    """(r, x, y, z) of a rotation whose third column is the unit normal n [P,3] (the surfel's plane is spanned by the other two)."""
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(n).clone()
    up[(n[:, 2].abs() > 0.9)] = torch.tensor([1.0, 0.0, 0.0])
    t1 = torch.nn.functional.normalize(torch.cross(up, n, dim=-1), dim=-1)
    t2 = torch.cross(n, t1, dim=-1)
    R = torch.stack([t1, t2, n], dim=-1)   # columns
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q = torch.zeros(n.shape[0], 4)
    # branch on the largest diagonal term (standard, numerically safe)
    for i in range(n.shape[0]):
        m = R[i]
        if tr[i] > 0:
            s = math.sqrt(float(tr[i]) + 1.0) * 2
            q[i] = torch.tensor([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
        else:
            k = int(torch.argmax(torch.diagonal(m)))
            a, b = (k + 1) % 3, (k + 2) % 3
            s = math.sqrt(max(1.0 + float(m[k, k] - m[a, a] - m[b, b]), 1e-12)) * 2
            v = [0.0, 0.0, 0.0]
            v[k] = 0.25 * s
            v[a] = float(m[a, k] + m[k, a]) / s
            v[b] = float(m[b, k] + m[k, b]) / s
            q[i] = torch.tensor([float(m[b, a] - m[a, b]) / s] + v)
    return q


class DynamicTruth:
    """Ground truth of the learnable test scene: a bobbing sphere and a swinging plate, a few thousand opaque surfels with smooth
    colours.  state(t) gives the rasterizer's inputs at time t in [0, 1].  detail > 0 adds a fine texture of that amplitude (period
    ~ 4 surfel spacings), which a fit can only reproduce with about as many surfels as the truth has: the size knob of
    bench.py's `trained` workload."""

    def __init__(self, n_sphere=2400, n_plate=1600, detail=0.0):
        i = torch.arange(n_sphere, dtype=torch.float32) + 0.5
        phi = torch.acos(1 - 2 * i / n_sphere)
        th = math.pi * (1 + 5 ** 0.5) * i
        self.sph_n = torch.stack([torch.cos(th) * torch.sin(phi), torch.sin(th) * torch.sin(phi), torch.cos(phi)], -1)
        self.sph_r = 0.55
        g = int(round(n_plate ** 0.5))
        u = (torch.arange(g, dtype=torch.float32) + 0.5) / g - 0.5
        self.plate_uv = torch.stack(torch.meshgrid(u, u, indexing="ij"), -1).reshape(-1, 2)
        self.sph_scale = 0.62 * math.sqrt(4 * math.pi * self.sph_r ** 2 / n_sphere)
        self.plate_scale = 0.62 / g
        self.sph_rgb = 0.5 + 0.45 * torch.sin(4.0 * self.sph_n + torch.tensor([0.0, 2.0, 4.0]))
        self.plate_rgb = torch.stack([0.5 + 0.45 * torch.sin(9.0 * self.plate_uv[:, 0]), 0.5 + 0.45 * torch.cos(7.0 * self.plate_uv[:, 1]),
                                      0.55 + 0.4 * torch.sin(5.0 * (self.plate_uv[:, 0] + self.plate_uv[:, 1]))], -1)
        if detail > 0:
            ks, kp = 0.5 * math.pi / (self.sph_scale / 0.62 / self.sph_r), 0.5 * math.pi * g
            ph = torch.tensor([0.0, 1.3, 2.9])
            self.sph_rgb = (self.sph_rgb + detail * torch.sin(ks * self.sph_n + ph) * torch.sin(ks * self.sph_n.roll(1, -1) + ph.flip(0))).clamp(0.02, 0.98)
            self.plate_rgb = (self.plate_rgb + detail * torch.sin(kp * self.plate_uv[:, :1] + ph) * torch.sin(kp * self.plate_uv[:, 1:] + ph.flip(0))).clamp(0.02, 0.98)
        self.sph_q = _quat_from_normal(self.sph_n)
        self.P = n_sphere + self.plate_uv.shape[0]

    def state(self, t: float):
        """-> (means3D [P,3], scales [P,2], rotations [P,4], opacities [P,1], shs [P,16,3]) at time t."""
        c_sph = torch.tensor([-0.55, 0.0, 0.25 * math.sin(2 * math.pi * t)])
        xyz_s = c_sph + self.sph_r * self.sph_n
        ang = 0.9 * (t - 0.5)
        ca, sa = math.cos(ang), math.sin(ang)
        Ry = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])
        local = torch.cat([self.plate_uv, torch.zeros(self.plate_uv.shape[0], 1)], -1) * 1.1
        xyz_p = torch.tensor([0.75, 0.0, 0.1]) + local @ Ry.T
        n_p = (Ry @ torch.tensor([0.0, 0.0, 1.0])).expand(self.plate_uv.shape[0], 3)
        xyz = torch.cat([xyz_s, xyz_p])
        scales = torch.cat([torch.full((xyz_s.shape[0], 2), self.sph_scale), torch.full((xyz_p.shape[0], 2), self.plate_scale * 1.1)])
        rot = torch.cat([self.sph_q, _quat_from_normal(n_p[:1]).expand(xyz_p.shape[0], 4)])
        opac = torch.full((self.P, 1), 0.97)
        shs = torch.zeros(self.P, 16, 3)
        shs[:, 0] = (torch.cat([self.sph_rgb, self.plate_rgb]) - 0.5) / 0.28209479177387814
        return xyz.contiguous(), scales.contiguous(), rot.contiguous(), opac, shs


def write_dynamic_dnerf(path, n_train=60, n_test=12, H=200, W=200, device="cuda:0", fov=0.6911, radius=4.0, truth=None):
    """Render DynamicTruth with THIS package's rasterizer into a D-NeRF / Blender-format dataset (transforms_{train,test}.json with
    camera_angle_x, per-frame time and transform_matrix; RGBA PNGs, straight alpha) that dgs_amd.io.load_dnerf -- and the
    reference's readNerfSyntheticInfo -- read.  Every frame has its own camera and its own time, like D-NeRF.  GPU only."""
    import json
    import os

    import numpy as np
    from PIL import Image

    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from .cameras import make_camera, pose_spherical
    truth = truth or DynamicTruth()
    bg = torch.zeros(3, device=device)
    for split, n, off in (("train", n_train, 0.0), ("test", n_test, 0.37)):
        os.makedirs(os.path.join(path, split), exist_ok=True)
        frames = []
        for k in range(n):
            theta = -180.0 + 360.0 * (((k + off) * 0.6180339887) % 1.0)
            phi = -15.0 - 35.0 * (((k + off) * 0.7548776662) % 1.0)
            t = k / max(n - 1, 1) if split == "train" else (k + 0.5) / n
            c2w = pose_spherical(theta, phi, radius)
            cam = make_camera(c2w, fov, fov, W, H, t).to(device)
            xyz, scales, rot, opac, shs = (x.to(device) for x in truth.state(t))
            rast = GaussianRasterizer(GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=math.tan(fov * 0.5), tanfovy=math.tan(fov * 0.5), bg=bg, scale_modifier=1.0,
                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center,
                prefiltered=False, debug=False))
            with torch.no_grad():
                color, _, allmap = rast(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=opac, shs=shs, scales=scales, rotations=rot)
            alpha = allmap[1:2].clamp(0, 1)
            straight = torch.where(alpha > 1e-4, color / alpha.clamp_min(1e-4), torch.zeros_like(color)).clamp(0, 1)
            rgba = torch.cat([straight, alpha]).permute(1, 2, 0).cpu().numpy()
            Image.fromarray(np.round(rgba * 255.0).astype(np.uint8), "RGBA").save(os.path.join(path, split, "r_%03d.png" % k))
            frames.append({"file_path": "./%s/r_%03d" % (split, k), "time": float(t), "transform_matrix": c2w.tolist()})
        with open(os.path.join(path, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": fov, "frames": frames}, f)
    return truth
