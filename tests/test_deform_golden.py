"""dgs_amd.deform (PyTorch restatement) against golden vectors produced by the imported reference
(tests/golden/make_deform_golden.py; BASELINE.json config 1 'node deform forward on PyTorch-CPU')."""
import os
import sys

import numpy as np
import torch

from dgs_amd.deform import ControlNodes, count_parameters, knn_points

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_deform_golden import fill_params  # noqa: E402  (shared deterministic parameter formula)


def _load():
    return np.load(os.path.join(HERE, "golden", "deform_golden.npz"))


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


def test_deform_forward_matches_reference_golden():
    g = _load()
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
