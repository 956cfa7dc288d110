"""Generates tests/golden/init_golden.npz by IMPORTING the reference's scene/gaussian_model.py and running
GaussianModel.create_from_pcd (gaussian_model.py:143-179) on 300 seeded points on the CPU.
`simple_knn._C.distCUDA2` (an absent CUDA extension) is stood in by its published semantics -- mean squared distance to the
three nearest other points (submodules/simple-knn/simple_knn.cu:148-183) -- computed here by brute force in float64; the rest
(SH conversion, log-scales, rotations, opacity, hyper coordinates, parameter layouts) is the reference's own code.
Run from the repo root:  python tests/golden/make_init_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_densify_golden import cuda_to_cpu, import_reference_model  # noqa: E402


def dist2_bruteforce(points):
    p = points.double()
    d = torch.cdist(p, p).pow(2)
    d.fill_diagonal_(float("inf"))
    return d.topk(3, dim=1, largest=False).values.mean(dim=1).float()


def main():
    import types
    GaussianModel = import_reference_model()
    # the module was loaded by file (not registered in sys.modules): reach its globals through a method of the class
    glob = GaussianModel.create_from_pcd.__globals__
    glob["distCUDA2"] = dist2_bruteforce
    mod = types.SimpleNamespace(BasicPointCloud=glob["BasicPointCloud"])
    g = torch.Generator().manual_seed(11)
    pts = ((torch.rand(300, 3, generator=g) * 2.6) - 1.3).numpy()
    cols = torch.rand(300, 3, generator=g).numpy()
    pcd = mod.BasicPointCloud(points=pts, colors=cols, normals=np.zeros_like(pts))
    gm = GaussianModel(3, fea_dim=8, with_motion_mask=False)
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    to_saved = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else to_saved(self, *a, **k)
    try:
        with cuda_to_cpu():
            gm.create_from_pcd(pcd, print_info=False)
    finally:
        torch.Tensor.cuda = saved
        torch.Tensor.to = to_saved
    out = dict(points=pts, colors=cols, xyz=gm._xyz.detach().numpy(), f_dc=gm._features_dc.detach().numpy(),
               f_rest=gm._features_rest.detach().numpy(), scaling=gm._scaling.detach().numpy(), rotation=gm._rotation.detach().numpy(),
               opacity=gm._opacity.detach().numpy(), feature=gm.feature.detach().numpy())
    np.savez_compressed(os.path.join(HERE, "init_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
