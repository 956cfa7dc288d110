"""dgs_amd.deform (PyTorch restatement) against golden vectors produced by the imported reference
(tests/golden/make_deform_golden.py; BASELINE.json config 1 'node deform forward on PyTorch-CPU')."""
import os
import sys

import numpy as np
import pytest
import torch

from dgs_amd.deform import ControlNodes, count_parameters, knn_points

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_deform_golden import fill_params  # noqa: E402  (shared deterministic parameter formula)


def _load(name="deform_golden.npz"):
    return np.load(os.path.join(HERE, "golden", name))


def _model(g):
    m = ControlNodes(node_num=g["nodes"].shape[0], K=3, hyper_dim=8, local_frame=True)
    fill_params(m)
    m.nodes.data = torch.tensor(g["nodes"])
    m._node_radius.data = torch.tensor(g["node_radius"])
    m._node_weight.data = torch.tensor(g["node_weight"])
    return m


def test_parameter_count_matches_reference():
    g = _load()
    m = _model(g)
    assert count_parameters(m.network) == int(g["n_params"]) == 523051


@pytest.mark.parametrize("name", ["deform_golden.npz", "deform_golden_c1.npz"])   # 48 nodes / 300 Gaussians; 512 / 5000 (BASELINE config 1)
def test_deform_forward_matches_reference_golden(name):
    g = _load(name)
    m = _model(g)
    x, feature, t, mm = (torch.tensor(g[k]) for k in ("x", "feature", "t", "motion_mask"))
    with torch.no_grad():
        net = m.network(m.nodes[..., :3], t)
        w, d, idx = m.nn_weights(x, feature)
        out = m(x, t, feature, mm)
    for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation"):
        assert np.allclose(net[k].numpy(), g["net_" + k], rtol=1e-5, atol=1e-6), k
    assert np.array_equal(idx.numpy(), g["nn_idx"])
    assert np.allclose(d.numpy(), g["nn_dist"], rtol=1e-5, atol=1e-7)
    assert np.allclose(w.numpy(), g["nn_weight"], rtol=1e-5, atol=1e-7)
    for k in ("d_xyz", "d_rotation", "d_scaling"):
        assert np.allclose(out[k].numpy(), g[k], rtol=1e-4, atol=1e-6), k


def test_knn_matches_bruteforce_and_is_differentiable():
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(500, 11, generator=gen, requires_grad=True)
    n = torch.randn(64, 11, generator=gen, requires_grad=True)
    d, i = knn_points(x, n, 3, chunk=128)
    full = ((x[:, None] - n[None]) ** 2).sum(-1)
    dr, ir = torch.topk(full, 3, dim=-1, largest=False)
    assert torch.equal(i, ir) and torch.allclose(d, dr, atol=1e-6)
    d.sum().backward()
    gx = x.grad.clone()
    x.grad = None
    dr.sum().backward()
    assert torch.allclose(gx, x.grad, atol=1e-5)


@pytest.mark.gpu
def test_fused_hip_deformation_matches_reference_golden():
    """The gfx950 kernels of the deformation (seeded / scanning KNN, node MLP on v_mfma_f32_4x4x1, fused skinning) against the vectors
    of the IMPORTED reference at BASELINE config 1's size (5000 Gaussians, 512 nodes) -- directly, not through the PyTorch
    restatement they replace."""
    g = _load("deform_golden_c1.npz")
    dev = torch.device("cuda:0")
    m = _model(g).to(dev)
    assert m.node_num % 64 == 0
    x, feature, t, mm = (torch.tensor(g[k], device=dev) for k in ("x", "feature", "t", "motion_mask"))
    with torch.no_grad():
        out = m(x, t, feature, mm)            # HIP tensors: ControlNodes.forward takes the fused kernels (_forward_fused)
        from dgs_amd import _ops
        H = m.hyper_dim
        nodes = torch.cat([m.nodes[..., :3], m.nodes[..., 3:]], dim=-1)
        idx = _ops.knn_indices(torch.cat([x, feature[..., :H]], dim=-1), nodes, m.K)
        attrs = _ops.fused_node_mlp(m.network, m.nodes, t)
    assert np.array_equal(idx.cpu().numpy(), g["nn_idx"])
    net = np.concatenate([g["net_local_rotation"] + np.array([1.0, 0, 0, 0], np.float32), g["net_d_xyz"], g["net_d_rotation"], g["net_d_scaling"]], -1)
    assert np.allclose(attrs.cpu().numpy(), net, rtol=2e-5, atol=2e-6)      # [M,13] attribute table: local rotation (+ bias), d_xyz, d_rotation, d_scaling
    for k in ("d_xyz", "d_rotation", "d_scaling"):
        assert np.allclose(out[k].cpu().numpy(), g[k], rtol=1e-4, atol=2e-6), k
