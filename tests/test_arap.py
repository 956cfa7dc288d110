"""dgs_amd.arap against the imported reference's ControlNodeWarp.arap_loss (tests/golden/make_arap_golden.py): loss value and
its gradient, with and without the 512-node subsample; the weight schedule."""
import os
import sys

import numpy as np
import torch

from dgs_amd import arap
from dgs_amd.deform import ControlNodes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_deform_golden import fill_params  # noqa: E402


def _model(g, tag):
    M = g[tag + "_nodes"].shape[0]
    d = ControlNodes(node_num=M, K=3, hyper_dim=8, local_frame=True)
    fill_params(d)
    with torch.no_grad():
        d.network.gaussian_warp.weight.mul_(50.0)
        d.nodes.copy_(torch.from_numpy(g[tag + "_nodes"].copy()))
    return d


def test_arap_loss_matches_reference():
    g = np.load(os.path.join(HERE, "golden", "arap_golden.npz"))
    for tag in ("small", "large"):
        d = _model(g, tag)
        idx = torch.from_numpy(g[tag + "_sample_idx"].copy()).long() if tag + "_sample_idx" in g else None
        loss = arap.arap_loss(d, t_samp=torch.from_numpy(g[tag + "_t_samp"].copy()), sample_idx=idx)
        assert abs(float(loss) - float(g[tag + "_loss"])) <= 2e-4 * float(g[tag + "_loss"]), (tag, float(loss), float(g[tag + "_loss"]))
        loss.backward()
        got, want = d.network.gaussian_warp.weight.grad.numpy(), g[tag + "_grad_warp"]
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), tag
    assert "large_sample_idx" in g and "small_sample_idx" not in g


def test_arap_loss_draws_its_own_samples():
    g = np.load(os.path.join(HERE, "golden", "arap_golden.npz"))
    d = _model(g, "large")
    gen = torch.Generator().manual_seed(5)
    a = arap.arap_loss(d, t=torch.tensor([0.4]), generator=gen)
    b = arap.arap_loss(d, t=torch.tensor([0.4]), generator=torch.Generator().manual_seed(5))
    c = arap.arap_loss(d, generator=gen)
    assert float(a) == float(b) and float(a) > 0 and float(c) > 0 and float(c) != float(a)


def test_lambda_arap_schedule_matches_reference():
    g = np.load(os.path.join(HERE, "golden", "arap_golden.npz"))
    for s, want in zip(g["lambda_steps"], g["lambda_arap"]):
        assert abs(arap.lambda_arap(int(s)) - want) <= 1e-12 * max(want, 1e-30), (s, want)


def test_trainer_adds_the_weighted_arap_term(monkeypatch):
    import dgs_amd.render as render_mod
    from dgs_amd.train import Trainer
    from oracle_raster_op import OracleRasterizer
    from test_train_step_cpu import _build
    monkeypatch.setattr(render_mod, "GaussianRasterizer", OracleRasterizer)
    res = {}
    for on in (False, True):
        surfels, deform, cams, targets, bg = _build(P=120, S=32, nodes=24, views=2)
        tr = Trainer(surfels, deform, cams, targets, bg, arap=on)
        tr.iteration = 100                      # lambda = 1e-4
        tr.opt_surfels.step = lambda: None
        tr.opt_deform.step = lambda: None
        loss = float(tr.step())
        res[on] = (loss, deform.network.gaussian_warp.weight.grad.clone(), deform)
    d = res[True][2]
    term = 1e-4 * float(arap.arap_loss(d, generator=torch.Generator().manual_seed(1234)))
    assert term > 0 and abs((res[True][0] - res[False][0]) - term) <= 1e-6 * abs(res[True][0]) + 1e-3 * term
    assert not torch.equal(res[True][1], res[False][1])
    # weight zero after iteration 20000: the term vanishes
    tr.iteration = 25000
    assert arap.lambda_arap(tr.iteration) == 0


def test_split_dp_step_is_refused_while_arap_is_active(monkeypatch):
    """The split data-parallel backward differentiates w.r.t. the assembled rasterizer inputs only; the ARAP term reaches
    the network weights directly, so while its weight is non-zero the trainer must fall back to the unsplit step."""
    import dgs_amd.render as render_mod
    from dgs_amd.train import Trainer, split_step_allowed
    from oracle_raster_op import OracleRasterizer
    from test_train_step_cpu import _build
    assert split_step_allowed(2, True, True, False) is True
    assert split_step_allowed(2, True, True, True) is False      # ARAP active
    assert split_step_allowed(1, True, True, False) is False     # single rank: nothing to overlap
    assert split_step_allowed(2, True, False, False) is False    # not the fused HIP path
    monkeypatch.setattr(render_mod, "GaussianRasterizer", OracleRasterizer)
    surfels, deform, cams, targets, bg = _build(P=60, S=32, nodes=24, views=2)
    tr = Trainer(surfels, deform, cams, targets, bg, arap=True)
    tr.iteration = 100
    assert tr._arap_active()
    tr.iteration = 25000                                        # weight schedule has reached zero
    assert not tr._arap_active()
    tr.arap = False
    tr.iteration = 100
    assert not tr._arap_active()


def test_padding_nodes_are_invisible_to_arap_and_checkpoints(tmp_path):
    """densify_nodes(pad_to=64) parks padding nodes far outside the scene; the ARAP connectivity and the checkpoint must
    behave as if they did not exist."""
    from dgs_amd import io as dio
    torch.manual_seed(0)
    d = ControlNodes(node_num=40, K=3, hyper_dim=8, local_frame=True)
    d.init_from_points(torch.rand(500, 3) * 2 - 1)
    ts = torch.tensor([0.3, 0.33])
    ref = arap.arap_loss(d, t_samp=ts, sample_idx=None, generator=torch.Generator().manual_seed(1))
    n_pad = 24
    with torch.no_grad():
        pad = torch.zeros(n_pad, d.nodes.shape[1])
        pad[:, :3] = d.FAR
        d.nodes = torch.nn.Parameter(torch.cat((d.nodes.detach(), pad)))
        d._node_radius = torch.nn.Parameter(torch.cat((d._node_radius.detach(), d._node_radius.detach().mean().expand(n_pad))))
        d._node_weight = torch.nn.Parameter(torch.cat((d._node_weight.detach(), torch.zeros(n_pad, 1))))
    assert int(d.live_nodes.sum()) == 40 and d.nodes.shape[0] == 64
    got = arap.arap_loss(d, t_samp=ts, sample_idx=None, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-9)
    path = dio.save_deform(d, str(tmp_path), 7)
    saved = torch.load(path, weights_only=True)
    assert saved["nodes"].shape[0] == 40 and saved["_node_radius"].shape[0] == 40 and saved["_node_weight"].shape[0] == 40
    assert torch.equal(saved["nodes"], d.nodes.detach()[:40])


def test_elastic_and_acc_loss_match_reference():
    """The two node regularisers of the reference's node pre-training stage (train_gui.py:502-504; ControlNodeWarp.elastic_loss /
    acc_loss, utils/time_utils.py:1091-1120) against what the imported reference computes (tests/golden/make_node_reg_golden.py):
    value and gradient with respect to the translation head, same time samples."""
    g = np.load(os.path.join(HERE, "golden", "node_reg_golden.npz"))
    M = g["nodes"].shape[0]
    d = ControlNodes(node_num=M, K=3, hyper_dim=8, local_frame=True)
    fill_params(d)
    with torch.no_grad():
        d.network.gaussian_warp.weight.mul_(50.0)
        d.nodes.copy_(torch.from_numpy(g["nodes"].copy()))
        d._node_radius.copy_(torch.from_numpy(g["node_radius_raw"].copy()))
        d._node_weight.copy_(torch.from_numpy(g["node_weight_raw"].copy()))
    for name, loss in (("elastic", arap.elastic_loss(d, delta_t=0.02, t_samp=torch.from_numpy(g["elastic_t_samp"].copy()))),
                       ("acc", arap.acc_loss(d, delta_t=0.06, t0=torch.from_numpy(g["acc_t0"].copy())))):
        d.zero_grad()
        loss.backward()
        want = float(g[name + "_loss"])
        assert abs(float(loss) - want) <= 2e-4 * abs(want), (name, float(loss), want)
        gw, ww = d.network.gaussian_warp.weight.grad.numpy(), g[name + "_grad_warp"]
        assert np.abs(gw - ww).max() <= 2e-3 * np.abs(ww).max(), (name, np.abs(gw - ww).max(), np.abs(ww).max())
    # drawing their own samples: reproducible per generator, different for different ones
    a = arap.elastic_loss(d, t=torch.tensor([0.4]), delta_t=0.02, generator=torch.Generator().manual_seed(5))
    b = arap.elastic_loss(d, t=torch.tensor([0.4]), delta_t=0.02, generator=torch.Generator().manual_seed(5))
    c = arap.acc_loss(d, t=torch.tensor([0.4]), generator=torch.Generator().manual_seed(6))
    assert float(a) == float(b) and torch.isfinite(c)
