"""Adaptive density control on slots (dgs_amd/densify.py) against the reference's GaussianModel.densify_and_prune /
reset_opacity run in the build container (tests/golden/make_densify_golden.py), plus Trainer.grow and the world_size-2
replica-consistency check.  The result must be the reference's SET of surfels (values, Adam moments, statistics); the
order is not part of the contract (slots)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dgs_amd.render as render_mod
from dgs_amd import densify
from dgs_amd.model import SurfelModel
from dgs_amd.synthetic import SurfelScene
from dgs_amd.train import Trainer
from oracle_raster_op import OracleRasterizer
from test_train_step_cpu import _build, _free_port

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_golden.npz")
NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "feature"]
ARGS = dict(lr=0.0, eps=1e-15)


def _model_from_golden(g, capacity):
    t = lambda k: torch.from_numpy(g[k].copy())
    scene = SurfelScene(t("pre_xyz"), t("pre_scaling"), t("pre_rotation"), t("pre_opacity"), t("pre_f_dc"), t("pre_f_rest"), t("pre_feature"))
    P = scene.xyz.shape[0]
    model = SurfelModel(scene, capacity=capacity)
    opt = torch.optim.Adam(model.optimizer_groups(), **ARGS)
    rows = densify.surfel_rows(model)
    for n in NAMES:
        p = rows[n]
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        m[:P], v[:P] = t("pre_m_" + n), t("pre_v_" + n)
        opt.state[p] = {"step": torch.tensor(3.0), "exp_avg": m, "exp_avg_sq": v}
    model.xyz_gradient_accum[:P] = t("accum")
    model.denom[:P] = t("denom")
    model.max_radii2D[:P] = t("max_radii2D").to(torch.int32)
    return model, opt


def _alive_table(model, moments):
    """One row per live surfel: all parameters and both moments, flattened and concatenated; rows sorted."""
    alive = model.alive
    cols = []
    for n, p in densify.surfel_rows(model).items():
        m, v = moments(p)
        cols += [p.detach()[alive].flatten(1), m[alive].flatten(1), v[alive].flatten(1)]
    return _sorted_rows(torch.cat(cols, dim=1).numpy())


def _golden_table(g, prefix):
    cols = []
    for n in NAMES:
        k = g[prefix + n]
        cols += [k.reshape(k.shape[0], -1), g[prefix + "m_" + n].reshape(k.shape[0], -1), g[prefix + "v_" + n].reshape(k.shape[0], -1)]
    return _sorted_rows(np.concatenate(cols, axis=1))


def _sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("capacity", [1024, 700])
def test_densify_and_prune_matches_reference(capacity):
    g = np.load(GOLD)
    model, opt = _model_from_golden(g, capacity)
    moments = densify.TorchAdamMoments(opt)
    P = g["pre_xyz"].shape[0]
    assert model.num_surfels == P and model.get_xyz.shape[0] == capacity
    out = densify.densify_and_prune(model, moments, 0.0002, 0.01, float(g["extent"]), 20, noise=torch.from_numpy(g["noise"].copy()))
    assert not isinstance(out, int)
    n_clone, n_split, n_pruned = out
    assert 2 * n_split == g["noise"].shape[0] and n_clone > 0 and n_split > 0 and n_pruned > 0
    assert model.num_surfels == g["post_xyz"].shape[0] == P + n_clone + n_split - n_pruned
    got, want = _alive_table(model, moments), _golden_table(g, "post_")
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    assert float(model.xyz_gradient_accum.abs().sum()) == 0 and float(model.denom.abs().sum()) == 0 and int(model.max_radii2D.max()) == 0
    # dead slots: opacity exactly zero, moments zero
    dead = ~model.alive
    assert float(torch.sigmoid(model._opacity.detach()[dead]).max()) == 0.0
    for p in densify.surfel_rows(model).values():
        m, v = moments(p)
        assert float(m[dead].abs().sum()) == 0 and float(v[dead].abs().sum()) == 0

    densify.reset_opacity(model, moments)
    got = np.sort(model._opacity.detach()[model.alive].numpy().reshape(-1))
    np.testing.assert_allclose(got, np.sort(g["reset_opacity"].reshape(-1)), rtol=1e-6)
    assert float(moments(model._opacity)[0].abs().sum()) == 0 and float(moments(model._opacity)[1].abs().sum()) == 0
    assert float(torch.sigmoid(model._opacity.detach()[~model.alive]).max()) == 0.0


def test_out_of_slots_is_reported_before_anything_changes():
    g = np.load(GOLD)
    model, opt = _model_from_golden(g, g["pre_xyz"].shape[0] + 8)
    before = [p.detach().clone() for p in densify.surfel_rows(model).values()]
    out = densify.densify_and_prune(model, densify.TorchAdamMoments(opt), 0.0002, 0.01, float(g["extent"]), 20,
                                    noise=torch.from_numpy(g["noise"].copy()))
    assert isinstance(out, int) and out > 0
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, densify.surfel_rows(model).values()))


@pytest.fixture
def _oracle_backend(monkeypatch):
    monkeypatch.setattr(render_mod, "GaussianRasterizer", OracleRasterizer)


def test_dead_slots_are_inert_and_trainer_grows(_oracle_backend):
    """A trainer with spare (dead) slots takes the same steps as one without; densification that runs out of slots
    grows the state and training continues with the moments carried over."""
    surfels, deform, cams, targets, bg = _build(P=200)
    tr0 = Trainer(surfels, deform, cams, targets, bg)
    surfels1, deform1, _, _, _ = _build(P=200)
    scene1 = SurfelScene(surfels1._xyz.detach(), surfels1._scaling.detach(), surfels1._rotation.detach(), surfels1._opacity.detach(),
                         surfels1._features_dc.detach(), surfels1._features_rest.detach(), surfels1.feature.detach())
    padded = SurfelModel(scene1, capacity=216)
    tr1 = Trainer(padded, deform1, cams, targets, bg)
    for _ in range(2):
        l0, l1 = float(tr0.step()), float(tr1.step())
        assert abs(l0 - l1) <= 1e-6 * abs(l0)
    dead = ~padded.alive
    assert int(dead.sum()) == 16
    for n, p in densify.surfel_rows(padded).items():
        q = densify.surfel_rows(surfels)[n]
        assert torch.allclose(p.detach()[:200], q.detach(), rtol=1e-5, atol=1e-7), n
        assert float(p.grad[dead].abs().sum()) == 0.0, n
    assert float(torch.sigmoid(padded._opacity.detach()[dead]).max()) == 0.0

    # force a densification that needs more than the 16 spare slots
    padded.xyz_gradient_accum.zero_()
    padded.xyz_gradient_accum[:120] = 1.0
    padded.denom[:200] = 1.0
    m_before = tr1._moments()(padded._xyz)[0][:200].clone()
    xyz_before = padded._xyz.detach()[:200].clone()
    n_clone, n_split, n_pruned = tr1.densify_and_prune(extent=4.0, max_screen_size=20)
    assert n_clone + n_split == 120 and tr1.P > 216 and padded.get_xyz.shape[0] == tr1.P
    assert padded.num_surfels == 200 + n_clone + n_split - n_pruned
    # unsplit, unpruned originals keep their values and moments across the re-allocation
    keep = padded.alive[:200] & (padded._xyz.detach()[:200] == xyz_before).all(dim=1)
    assert int(keep.sum()) > 0
    assert torch.equal(tr1._moments()(padded._xyz)[0][:200][keep], m_before[keep])
    loss = float(tr1.step())
    assert loss == loss
    tr1.reset_opacity()
    assert float(torch.sigmoid(padded._opacity.detach()[padded.alive]).max()) <= 0.0100001


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    render_mod.GaussianRasterizer = OracleRasterizer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        surfels, deform, cams, targets, bg = _build(P=200)
        scene = SurfelScene(surfels._xyz.detach(), surfels._scaling.detach(), surfels._rotation.detach(), surfels._opacity.detach(),
                            surfels._features_dc.detach(), surfels._features_rest.detach(), surfels.feature.detach())
        model = SurfelModel(scene, capacity=512)
        tr = Trainer(model, deform, cams, targets, bg)
        tr.step()
        tr.step()
        # thresholds low enough that this tiny scene clones and splits
        counts = tr.densify_and_prune(max_grad=1e-7, extent=0.5, max_screen_size=20, seed=3)
        tr.step()
        state = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params] + [model.alive.float()])
        gathered = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        if rank == 0:
            q.put((same, tuple(int(c) for c in counts), model.num_surfels))
    finally:
        dist.destroy_process_group()


def test_replicas_stay_identical_through_densification_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, counts, alive = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert same
    assert counts[0] + counts[1] > 0 and alive == 200 + counts[0] + counts[1] - counts[2]


def test_node_densification_matches_reference():
    """ControlNodes.densify_nodes against the imported reference's ControlNodeWarp.densify (make_node_densify_golden.py):
    importance, new nodes at the weighted means, pruned orphans, Adam moments of the three node parameters."""
    from dgs_amd.deform import ControlNodes
    g = np.load(os.path.join(os.path.dirname(GOLD), "node_densify_golden.npz"))
    t = lambda k: torch.from_numpy(g[k].copy())
    d = ControlNodes(node_num=48, K=3, hyper_dim=8, local_frame=True)
    with torch.no_grad():
        d.nodes.copy_(t("pre_nodes")); d._node_radius.copy_(t("pre__node_radius")); d._node_weight.copy_(t("pre__node_weight"))
    mom = {id(getattr(d, n)): (t("pre_m_" + n), t("pre_v_" + n)) for n in ("nodes", "_node_radius", "_node_weight")}
    imp, avg, edges = d.node_importance(t("x"), t("x_grad").norm(dim=-1), t("feature"))
    np.testing.assert_allclose(imp.numpy(), g["importance"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(edges.numpy(), g["edge_count"], rtol=2e-5, atol=1e-9)
    ok = ~np.isnan(g["avg_x"]).any(axis=1)
    assert np.array_equal(ok, ~avg.isnan().any(dim=1).numpy())
    np.testing.assert_allclose(avg.numpy()[ok], g["avg_x"][ok], rtol=2e-4, atol=1e-6)
    n_add, n_prune, new_m = d.densify_nodes(0.0002, t("x"), t("x_grad"), t("feature"), moments=lambda p: mom[id(p)])
    assert (n_add, n_prune) == (6, 4) and d.node_num == g["post_nodes"].shape[0] == 50
    for n in ("nodes", "_node_radius", "_node_weight"):
        np.testing.assert_allclose(getattr(d, n).detach().numpy(), g["post_" + n], rtol=2e-4, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(new_m[n][0].numpy(), g["post_m_" + n], rtol=1e-6, atol=0, err_msg=n)
        np.testing.assert_allclose(new_m[n][1].numpy(), g["post_v_" + n], rtol=1e-6, atol=0, err_msg=n)
    # padding to a multiple of 64: the extra nodes are unreachable and come out again at the next call
    d2 = ControlNodes(node_num=48, K=3, hyper_dim=8, local_frame=True)
    with torch.no_grad():
        d2.nodes.copy_(t("pre_nodes")); d2._node_radius.copy_(t("pre__node_radius")); d2._node_weight.copy_(t("pre__node_weight"))
    d2.densify_nodes(0.0002, t("x"), t("x_grad"), t("feature"), pad_to=64)
    assert d2.node_num == 64 and torch.equal(d2.nodes.detach()[:50], d.nodes.detach())
    _, _, idx = d2.nn_weights(t("x"), t("feature"))
    assert int(idx.max()) < 50
    res = d2.densify_nodes(1e9, t("x"), torch.zeros_like(t("x_grad")), t("feature"))
    assert res is not None and res[0] == 0 and res[1] >= 14 and d2.node_num <= 50


def test_trainer_densifies_nodes_and_keeps_training(_oracle_backend):
    surfels, deform, cams, targets, bg = _build(P=200)
    tr = Trainer(surfels, deform, cams, targets, bg)
    for _ in range(2):
        tr.step()
    with torch.no_grad():
        deform.nodes[3:6, :3] += 50.0      # orphan three nodes
    t_before = tr.opt_deform.state[deform.network.linear[0].weight]["exp_avg"].clone()
    out = tr.densify_nodes(max_grad=1e-7)
    assert out is not None and out[1] >= 3 and deform.node_num == 32 + out[0] - out[1]
    assert torch.equal(tr.opt_deform.state[deform.network.linear[0].weight]["exp_avg"], t_before)   # network moments carried over
    assert tr.opt_deform.state[deform.nodes]["exp_avg"].shape == deform.nodes.shape
    l = float(tr.step())
    assert l == l and deform.nodes.grad is not None and deform.nodes.grad.shape == deform.nodes.shape
