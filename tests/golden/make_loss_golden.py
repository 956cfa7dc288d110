"""Generates tests/golden/loss_golden.npz by IMPORTING the reference's loss / point utilities
(/root/reference/utils/loss_utils.py: l1_loss, ssim;  utils/point_utils.py: depths_to_points, depth_to_normal) in the
build container and running them on the CPU.  cv2 / matplotlib (imported, unused on this path) are stubbed and `.cuda()` is
patched to the identity.  Only inputs and outputs are stored; the inputs come from the deterministic formulas below, which
the tests share.  Run from the repo root:  python tests/golden/make_loss_golden.py
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def images(C=3, H=40, W=52):
    """Two smooth-plus-texture images in [0, 1]."""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    chans1, chans2 = [], []
    for c in range(C):
        a = 0.5 + 0.3 * torch.sin(0.21 * x + 0.13 * y + c) + 0.15 * torch.sin(1.7 * x * (c + 1) + 0.9 * y)
        b = 0.5 + 0.3 * torch.sin(0.19 * x + 0.16 * y + 0.4 + c) + 0.12 * torch.cos(1.3 * x + 2.1 * y * (c + 1))
        chans1.append(a.clamp(0, 1))
        chans2.append(b.clamp(0, 1))
    return torch.stack(chans1).float(), torch.stack(chans2).float()


def depth_map(H=40, W=52):
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    return (3.0 + 0.4 * torch.sin(0.23 * x) * torch.cos(0.17 * y) + 0.01 * x).float()[None]


class View:
    """The attributes depths_to_points reads from a camera."""

    def __init__(self, H=40, W=52):
        self.image_height, self.image_width = H, W
        self.FoVx, self.FoVy = 0.9, 0.72
        ang = 0.3
        R = torch.tensor([[math.cos(ang), 0.0, math.sin(ang)], [0.0, 1.0, 0.0], [-math.sin(ang), 0.0, math.cos(ang)]])
        w2c = torch.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = torch.tensor([0.2, -0.1, 4.0])
        self.world_view_transform = w2c.T.contiguous()   # stored transposed, like the reference's cameras


def import_reference():
    for name in ("cv2", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    mods = {}
    for name in ("loss_utils", "point_utils"):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, "utils", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["loss_utils"], mods["point_utils"]


def main():
    lu, pu = import_reference()
    a, b = images()
    view = View()
    depth = depth_map()
    normal, points = pu.depth_to_normal(view, depth)
    out = dict(
        ssim=np.float64(lu.ssim(a, b).item()), l1=np.float64(lu.l1_loss(a, b).item()),
        ssim_self=np.float64(lu.ssim(a, a).item()),
        normal=normal.numpy().astype(np.float32), points=points.numpy().astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
