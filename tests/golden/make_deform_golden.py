"""Generates tests/golden/deform_golden.npz by IMPORTING the reference's deformation module
(/root/reference/utils/time_utils.py) in the build container and running it on the CPU.

The reference needs pytorch3d (absent here, and unpinned in the reference: readme.md:61); its
knn_points is stubbed with the published semantics (squared L2, K nearest, ascending) -- the only
third-party arithmetic on this path.  `.cuda()` is patched to the identity.  Only inputs and outputs
are stored; parameters are filled by the deterministic formula `fill_params` shared with the test.
Run from the repo root:  python tests/golden/make_deform_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def fill_params(module):
    """Deterministic, init-order independent parameter values (same names in both implementations)."""
    for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
        n = p.numel()
        k = sum(ord(c) for c in name) % 97
        base = torch.sin(0.37 * torch.arange(n, dtype=torch.float64) + k).to(torch.float32).reshape(p.shape)
        fan_in = p.shape[-1] if p.dim() > 1 else 1
        scale = 0.8 / np.sqrt(fan_in) if p.dim() > 1 else 0.05
        p.data = base * scale


def knn_stub(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **kw):
    d = ((p1[0][:, None, :] - p2[0][None, :, :]) ** 2).sum(-1)
    dist, idx = torch.topk(d, K, dim=-1, largest=False, sorted=True)
    nn_pts = p2[0][idx][None] if return_nn else None
    return dist[None], idx[None], nn_pts


def import_reference():
    for name in ("pytorch3d", "pytorch3d.ops", "pytorch3d.loss", "pytorch3d.loss.mesh_laplacian_smoothing", "pytorch3d.io"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pytorch3d.ops"].knn_points = knn_stub
    sys.modules["pytorch3d.ops"].ball_query = lambda *a, **k: None  # imported by utils/deform_utils.py, unused here
    sys.modules["pytorch3d"].ops = sys.modules["pytorch3d.ops"]
    sys.modules["pytorch3d.loss.mesh_laplacian_smoothing"].cot_laplacian = lambda *a, **k: None
    sys.modules["pytorch3d.io"].load_ply = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    import utils.time_utils as tu
    return tu


def main(node_num=48, N=300, out_name="deform_golden.npz"):
    """(48 nodes, 300 Gaussians): the small fixture; (512, 5000): BASELINE.json config 1 at its stated size (`deform_golden_c1.npz`;
    512 nodes is also the configuration the fused gfx950 kernels cover: node count a multiple of 64)."""
    tu = import_reference()
    torch.manual_seed(0)
    ref = tu.ControlNodeWarp(is_blender=True, node_num=node_num, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=True,
                             with_arap_loss=False, with_node_weight=True)
    fill_params(ref)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(N, 3, generator=g) * 2 - 1) * 1.3
    feature = 0.05 * torch.randn(N, 8, generator=g)
    ref.nodes.data = torch.cat([x[:node_num].clone() + 0.01, 0.01 + 0.02 * torch.rand(node_num, 8, generator=g)], -1)
    ref._node_radius.data = torch.log(torch.tensor(0.26)) + 0.1 * torch.randn(node_num, generator=g)
    ref._node_weight.data = 0.3 * torch.randn(node_num, 1, generator=g)
    t = torch.full((node_num, 1), 0.37)
    motion_mask = torch.sigmoid(torch.randn(N, 1, generator=g))
    ref.train()
    with torch.no_grad():
        out = ref(x, t, feature=feature, motion_mask=motion_mask, iteration=30000)
        w, d, idx = ref.cal_nn_weight(x=x, feature=feature)
        net = ref.network(ref.nodes[..., :3], t)
    np.savez_compressed(
        os.path.join(HERE, out_name),
        x=x.numpy(), feature=feature.numpy(), t=t.numpy(), motion_mask=motion_mask.numpy(),
        nodes=ref.nodes.data.numpy(), node_radius=ref._node_radius.data.numpy(), node_weight=ref._node_weight.data.numpy(),
        d_xyz=out["d_xyz"].numpy(), d_rotation=out["d_rotation"].numpy(), d_scaling=out["d_scaling"].numpy(),
        nn_weight=w.numpy(), nn_dist=d.numpy(), nn_idx=idx.numpy(),
        net_d_xyz=net["d_xyz"].numpy(), net_d_rotation=net["d_rotation"].numpy(), net_d_scaling=net["d_scaling"].numpy(),
        net_local_rotation=net["local_rotation"].numpy(),
        n_params=np.array(sum(p.numel() for p in ref.network.parameters())))
    print("wrote %s; network params:" % out_name, sum(p.numel() for p in ref.network.parameters()))


if __name__ == "__main__":
    main()
    main(512, 5000, "deform_golden_c1.npz")
