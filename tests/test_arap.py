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


def test_graph_distances_and_rotation_term_match_reference():
    """The pieces off the default training path (tests/golden/make_floyd_golden.py, imported reference): Floyd-Warshall graph distances
    (utils/time_utils.py:1122-1131), graph skinning weights (:969-984), connectivity from trajectories / in 'floyd' mode
    (utils/deform_utils.py:58-110), and arap_loss_with_rot (time_utils.py:1035-1042 over deform_utils.py:246-289) with its gradients."""
    g = np.load(os.path.join(HERE, "golden", "floyd_golden.npz"))
    cur, x = torch.from_numpy(g["cur_node"].copy()), torch.from_numpy(g["x"].copy())
    for K in (2, 4):
        d = arap.geodesic_distance_floyd(cur, K=K).numpy()
        assert np.array_equal(d, g["geo_K%d" % K]), K              # same operations in the same order: bit for bit
        assert np.array_equal(d, d.T) and (np.diag(d) == 0).all()
    assert np.isinf(g["geo_K2"]).any() and not np.isinf(g["geo_K4"]).any()   # the fixture has a disconnected and a connected graph
    cache = {}
    w, dist, idx = arap.nn_weight_floyd(x, cur, K=8, GraphK=3, temperature=1e-3, t0=torch.tensor([0.3]), cache=cache)
    assert np.array_equal(idx.numpy(), g["floyd_idx"]) and np.array_equal(dist.numpy(), g["floyd_d"])
    np.testing.assert_allclose(w.numpy(), g["floyd_w"], rtol=1e-6, atol=1e-30)
    held = cache["nn_dist"]
    arap.nn_weight_floyd(x, cur + 1.0, K=8, GraphK=3, temperature=1e-3, t0=torch.tensor([0.305]), cache=cache)
    assert cache["nn_dist"] is held                               # within 1e-2 of the cached time: the graph is not rebuilt
    arap.nn_weight_floyd(x, cur * 2.0, K=8, GraphK=3, temperature=1e-3, t0=torch.tensor([0.32]), cache=cache)
    assert cache["nn_dist"] is not held and torch.allclose(cache["nn_dist"], 2.0 * held)
    w, dist, idx = arap.nn_weight_floyd(cur, cur, K=9, GraphK=4, temperature=1e-1, XisNode=True, t0=torch.tensor([0.3]))
    assert np.array_equal(idx.numpy(), g["floyd_self_idx"]) and np.array_equal(dist.numpy(), g["floyd_self_d"])
    np.testing.assert_allclose(w.numpy(), g["floyd_self_w"], rtol=1e-6, atol=1e-30)
    assert not (idx == torch.arange(cur.shape[0])[:, None]).any()  # the node itself is skipped
    traj = torch.from_numpy(g["traj"].copy())
    rad = torch.from_numpy(g["node_radius_conn"].copy())
    for tag, kw in (("nn", dict(radius=0.15, K=6, trajectory=traj, GraphK=3)), ("floyd", dict(radius=0.15, K=6, trajectory=traj, mode="floyd", GraphK=3)),
                    ("pts_floyd", dict(radius=0.15, K=6, mode="floyd", GraphK=3)), ("rad", dict(radius=0.2, K=6, node_radius=rad, adaptive_weighting=False))):
        ii, jj, nn, weight = arap.connectivity_from_points(cur, **kw)
        for name, v in (("ii", ii), ("jj", jj), ("nn", nn)):
            assert np.array_equal(v.numpy(), g["conn_%s_%s" % (tag, name)]), (tag, name)
        np.testing.assert_allclose(weight.numpy(), g["conn_%s_w" % tag], rtol=1e-6, equal_nan=True)
    # the reference's adaptive weights as soon as ANY neighbour is dropped (the mean distance is inf): NaN in the rows with a dropped
    # neighbour, uniform 1 / K in the others -- kept as it is
    wf = g["conn_floyd_w"]
    assert np.isnan(wf).any() and np.isfinite(g["conn_rad_w"]).all() and np.allclose(wf[np.isfinite(wf)], 1.0 / 6)
    # ---- arap_loss_with_rot: absolute node rotations and the residual default
    for tag, as_res in (("rot", False), ("norot", True)):
        M = g["ball"].shape[0]
        d = ControlNodes(node_num=M, K=3, hyper_dim=8, local_frame=True)
        fill_params(d)
        with torch.no_grad():
            d.network.gaussian_warp.weight.mul_(50.0)
            d.nodes.copy_(torch.cat([torch.from_numpy(g["ball"].copy()), 0.01 * torch.ones(M, 8)], -1))
            d._node_radius.copy_(torch.from_numpy(g[tag + "_node_radius_raw"].copy()))
        loss = arap.arap_loss_with_rot(d, d_rot_as_res=as_res, t_samp=torch.from_numpy(g[tag + "_t_samp"].copy()), fid=int(g[tag + "_fid"]))
        d.zero_grad()
        loss.backward()
        want = float(g[tag + "_loss"])
        assert abs(float(loss) - want) <= 2e-4 * abs(want), (tag, float(loss), want)
        gw, ww = d.network.gaussian_warp.weight.grad.numpy(), g[tag + "_grad_warp"]
        assert np.abs(gw - ww).max() <= 2e-3 * np.abs(ww).max(), (tag, np.abs(gw - ww).max(), np.abs(ww).max())
        if not as_res:
            gr, wr = d.network.gaussian_rotation.weight.grad.numpy(), g[tag + "_grad_rot"]
            assert np.abs(wr).max() > 0 and np.abs(gr - wr).max() <= 2e-3 * np.abs(wr).max(), (np.abs(gr - wr).max(), np.abs(wr).max())
    # drawing its own samples: reproducible per generator
    a = arap.arap_loss_with_rot(d, t_samp_num=16, generator=torch.Generator().manual_seed(5))
    b = arap.arap_loss_with_rot(d, t_samp_num=16, generator=torch.Generator().manual_seed(5))
    assert float(a) == float(b)
