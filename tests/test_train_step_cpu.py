"""Host logic on the CPU: render() dict, loss, Trainer step and the world_size-2 gloo data-parallel path.
The rasterizer call is served by the test-only oracle operator (tests/oracle_raster_op.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dgs_amd.render as render_mod
from dgs_amd.cameras import orbit_cameras
from dgs_amd.deform import ControlNodes
from dgs_amd.losses import training_loss
from dgs_amd.model import SurfelModel
from dgs_amd.synthetic import make_scene, target_image
from dgs_amd.train import Trainer
from oracle_raster_op import OracleRasterizer


def _build(P=300, S=48, nodes=32, views=4, seed=0):
    torch.manual_seed(seed)
    scene = make_scene(P, seed)
    scene = scene._replace(log_scale=scene.log_scale + 0.9)  # bigger surfels: everything visible at 48x48
    surfels = SurfelModel(scene)
    deform = ControlNodes(node_num=nodes)
    deform.init_from_points(scene.xyz)
    with torch.no_grad():  # non-trivial deformation (default init is ~1e-5)
        deform.network.gaussian_warp.weight.mul_(2e3)
        deform.network.gaussian_rotation.weight.mul_(2e3)
    cams = orbit_cameras(views, S, S)
    targets = [target_image(S, S, seed=1 + v) for v in range(views)]
    return surfels, deform, cams, targets, torch.zeros(3)


@pytest.fixture(autouse=True)
def _oracle_backend(monkeypatch):
    monkeypatch.setattr(render_mod, "GaussianRasterizer", OracleRasterizer)


def test_render_dict_and_loss_backward():
    surfels, deform, cams, targets, bg = _build()
    cam = cams[1]
    dv = deform(surfels.get_xyz.detach(), deform.expand_time(cam.fid), surfels.feature, surfels.motion_mask)
    pkg = render_mod.render(cam, surfels, bg, dv['d_xyz'], dv['d_rotation'], dv['d_scaling'])
    for k in ("render", "viewspace_points", "visibility_filter", "radii", "alpha", "rend_normal", "rend_dist", "depth",
              "surf_normal", "surf_point", "bg_color"):  # gaussian_renderer/__init__.py:154-217
        assert k in pkg, k
    assert pkg["render"].shape == (3, 48, 48) and pkg["alpha"].shape == (1, 48, 48) and pkg["surf_normal"].shape == (3, 48, 48)
    loss = training_loss(pkg, targets[1])
    loss.backward()
    assert torch.isfinite(loss)
    assert surfels._xyz.grad.abs().sum() > 0 and surfels._features_rest.grad.abs().sum() > 0
    assert surfels.feature.grad.abs().sum() > 0          # through the KNN distances (hyper coordinates)
    assert deform.network.linear[0].weight.grad.abs().sum() > 0 and deform.nodes.grad[:, 3:].abs().sum() > 0
    assert pkg["viewspace_points"].grad[:, :2].abs().sum() > 0 and float(pkg["viewspace_points"].grad[:, 2].abs().sum()) == 0


def test_trainer_single_process_updates_everything():
    surfels, deform, cams, targets, bg = _build()
    tr = Trainer(surfels, deform, cams, targets, bg)
    before = [p.detach().clone() for p in tr.bucket.params]
    l0 = float(tr.step())
    l1 = float(tr.step())
    assert l0 == l0 and l1 == l1
    changed = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, tr.bucket.params))
    assert changed == len(before)
    assert float(surfels.denom.sum()) > 0 and int(surfels.max_radii2D.max()) > 0


def test_node_storage_order_is_free():
    """Trainer.sort_nodes / reorder_nodes permute the control nodes in place (Morton order: what lets the neighbour kernel skip
    blocks of far nodes): same deformation of every surfel, Adam moments and the neighbour seed follow their nodes."""
    surfels, deform, cams, targets, bg = _build(nodes=64)
    tr = Trainer(surfels, deform, cams, targets, bg)
    tr.step(); tr.step()                                   # non-zero moments
    t = deform.expand_time(cams[1].fid)
    with torch.no_grad():
        want = deform(surfels.get_xyz.detach(), t, surfels.feature, surfels.motion_mask)
    x0 = deform.nodes.detach().clone()
    m0 = tr._any_moments(deform.nodes)[0].clone()
    r0 = deform._node_radius.detach().clone()
    deform._knn_seed = torch.stack([torch.arange(surfels.get_xyz.shape[0]) % 64, torch.full((surfels.get_xyz.shape[0],), -1)], 1)
    seed0 = deform._knn_seed.clone()
    tr.sort_nodes()
    x1 = deform.nodes.detach()
    # a permutation: every old row is somewhere, with its radius and its moment
    same = (x1[:, None, :] == x0[None, :, :]).all(-1)
    assert bool((same.sum(1) == 1).all())
    perm = same.float().argmax(1)
    assert torch.equal(torch.sort(perm).values, torch.arange(64))
    assert not torch.equal(perm, torch.arange(64))
    assert torch.equal(deform._node_radius.detach(), r0[perm]) and torch.equal(tr._any_moments(deform.nodes)[0], m0[perm])
    assert torch.equal(perm[deform._knn_seed[:, 0]], seed0[:, 0]) and torch.equal(deform._knn_seed[:, 1], seed0[:, 1])
    del deform._knn_seed
    with torch.no_grad():
        got = deform(surfels.get_xyz.detach(), t, surfels.feature, surfels.motion_mask)
    for k in ("d_xyz", "d_rotation", "d_scaling"):
        assert torch.allclose(got[k], want[k], rtol=1e-5, atol=1e-7), k
    # Morton order: consecutive nodes are close (mean step well under the mean distance of random pairs)
    step = (x1[1:, :3] - x1[:-1, :3]).norm(dim=1).mean()
    assert float(step) < 0.6 * float(torch.pdist(x1[:, :3]).mean())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    render_mod.GaussianRasterizer = OracleRasterizer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        surfels, deform, cams, targets, bg = _build()
        tr = Trainer(surfels, deform, cams, targets, bg)
        assert tr.view_for(0) == rank
        # bytes on the wire: count what the step hands to the collectives and compare with the trainer's own accounting
        sent = []
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            sent.append(t.numel() * t.element_size())
            return real_all_reduce(t, *a, **k)
        dist.all_reduce = counting_all_reduce
        tr.step()
        dist.all_reduce = real_all_reduce
        wire = tr.wire_bytes_per_step()
        P_ = surfels.get_xyz.shape[0]
        n_deform = sum(p.numel() for p in deform.parameters() if p.requires_grad)
        # 66 floats per surfel (xyz 3, SH 48, opacity 1, scaling 2, rotation 4, hyper feature 8) + deformation + 2 statistics
        assert wire["total"] == 4 * (66 * P_ + n_deform + 2 * P_) + 4 * (P_ + 4) and sum(sent) == wire["total"], (sent, wire)
        flat_after = tr.bucket.flat.clone()
        tr.step()
        params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params])
        gathered = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        stats = torch.cat([surfels.xyz_gradient_accum.reshape(-1), surfels.denom.reshape(-1), surfels.max_radii2D.float()])
        gs = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(gs, stats)
        same_stats = all(torch.equal(gs[0], g) for g in gs)
        if rank == 0:
            q.put((same, same_stats, flat_after.numpy().copy()))  # by value: a tensor would be shared through an fd of this process, which may exit first
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, same_stats, flat_dp = q.get()
    flat_dp = torch.from_numpy(flat_dp)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert same and same_stats
    # the all-reduced bucket == mean of the two views' single-process gradients (+ summed statistics)
    render_mod.GaussianRasterizer = OracleRasterizer
    flats = []
    for view in range(world):
        surfels, deform, cams, targets, bg = _build()
        tr = Trainer(surfels, deform, cams, targets, bg)
        tr.view_for = lambda it, j=0, v=view: v
        tr.opt_surfels.step = lambda: None
        tr.opt_deform.step = lambda: None
        tr.step()
        flats.append(tr.bucket.flat.clone())
    n = tr.bucket.n_grad
    expect = torch.cat([(flats[0][:n] + flats[1][:n]) / 2, flats[0][n:] + flats[1][n:]])
    assert torch.allclose(flat_dp, expect, rtol=1e-4, atol=1e-7)


def _single_view_flats(views):
    """Gradient bucket (+ statistics tail) of one single-view step per view, optimisers stubbed."""
    flats = []
    for view in views:
        surfels, deform, cams, targets, bg = _build()
        tr = Trainer(surfels, deform, cams, targets, bg)
        tr.view_for = lambda it, j=0, v=view: v
        tr.opt_surfels.step = lambda: None
        tr.opt_deform.step = lambda: None
        tr.step()
        flats.append(tr.bucket.flat.clone())
    return flats, tr.bucket.n_grad


def test_views_per_rank_accumulates_k_views_before_one_update():
    """Trainer(views_per_rank=2), one process: a step renders views 0 and 1, ADDS their gradients and updates once -- the bucket the
    optimiser sees is the mean of the two single-view gradients, the statistics tail their sum, the radii their maximum; the schedule
    is view_for(i, j) = ((i k + j) world + rank) mod V; the neighbour search of the second view is the first one's."""
    surfels, deform, cams, targets, bg = _build()
    tr = Trainer(surfels, deform, cams, targets, bg, views_per_rank=2)
    assert [tr.view_for(0, 0), tr.view_for(0, 1), tr.view_for(1, 0), tr.view_for(1, 1), tr.view_for(2, 0)] == [0, 1, 2, 3, 0]
    seen = []
    tr.opt_surfels.step = lambda: seen.append(tr.bucket.flat.clone())
    tr.opt_deform.step = lambda: None
    loss = tr.step()
    assert tr.iteration == 1 and len(seen) == 1
    flats, n = _single_view_flats([0, 1])
    expect = torch.cat([(flats[0][:n] + flats[1][:n]) / 2, flats[0][n:] + flats[1][n:]])
    assert torch.allclose(seen[0], expect, rtol=1e-4, atol=1e-7)
    assert float(loss) == float(loss) and float(surfels.denom.sum()) > 0
    # ... and a real run updates everything once per step
    surfels, deform, cams, targets, bg = _build()
    tr = Trainer(surfels, deform, cams, targets, bg, views_per_rank=2)
    before = [p.detach().clone() for p in tr.bucket.params]
    tr.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, tr.bucket.params))
    assert int(surfels.denom.max()) == 2          # two views counted by the statistics of ONE step


def _dp_k_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    render_mod.GaussianRasterizer = OracleRasterizer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        surfels, deform, cams, targets, bg = _build()
        tr = Trainer(surfels, deform, cams, targets, bg, views_per_rank=2)
        assert [tr.view_for(0, 0), tr.view_for(0, 1)] == [rank, 2 + rank]
        sent = []
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            sent.append(t.numel() * t.element_size())
            return real_all_reduce(t, *a, **k)
        dist.all_reduce = counting_all_reduce
        seen = []
        real_step = tr.opt_surfels.step
        tr.opt_surfels.step = lambda: (seen.append(tr.bucket.flat.clone()), real_step())[1]
        tr.step()
        dist.all_reduce = real_all_reduce
        tr.step()
        params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params])
        gathered = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            q.put((same, len(sent), sum(sent), tr.wire_bytes_per_step()["total"], seen[0].numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_views_per_rank_two_ranks_one_exchange_per_step():
    """Two ranks x two views per rank: ONE exchange per step (the same two all-reduces and bytes as a single-view step), replicas
    bit-identical, and the bucket the optimisers see is the mean of the FOUR views' gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_k_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, n_calls, n_bytes, wire, flat = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert same
    assert n_calls == 2 and n_bytes == wire, (n_calls, n_bytes, wire)
    render_mod.GaussianRasterizer = OracleRasterizer
    flats, n = _single_view_flats([0, 1, 2, 3])
    expect = torch.cat([sum(f[:n] for f in flats) / 4, sum(f[n:] for f in flats)])
    assert torch.allclose(torch.from_numpy(flat), expect, rtol=1e-4, atol=1e-7)


def test_trainer_lr_schedule_sets_the_reference_rates():
    """CPU path: step t runs at schedule(t - 1) for xyz and the deformation network; the node group keeps its initial rate."""
    from dgs_amd.train import expon_lr
    surfels, deform, cams, targets, bg = _build(P=60, S=32, nodes=16, views=2)
    tr = Trainer(surfels, deform, cams, targets, bg, lr_schedule=True)
    seen = []
    orig = tr.opt_surfels.step
    tr.opt_surfels.step = lambda: (seen.append((tr.opt_surfels.param_groups[0]["lr"], tr.opt_deform.param_groups[0]["lr"],
                                                 tr.opt_deform.param_groups[1]["lr"])), orig())[1]
    for _ in range(3):
        tr.step()
    for k, (lx, ld, ln) in enumerate(seen):
        assert abs(lx - expon_lr(k, 0.00016 * 5, 0.0000016 * 5, 30000)) <= 1e-12
        assert abs(ld - expon_lr(k, 0.00016 * 5, 0.0000016, 40000)) <= 1e-12
        assert ln == 0.00016 * 5
    assert seen[0][0] == 0.00016 * 5 and seen[2][0] < seen[1][0] < seen[0][0]


def test_trace_stages_are_noops_unless_enabled(monkeypatch):
    """dgs_amd.trace: roctx ranges around the step's stages only with DGS_ROCTX=1; without it (and without the library) the context
    manager does nothing and costs nothing."""
    import importlib
    from dgs_amd import trace
    assert not trace.enabled()
    with trace.stage("x"):
        pass
    monkeypatch.setenv("DGS_ROCTX", "1")
    t2 = importlib.reload(trace)
    try:
        with t2.stage("dgs.test"):      # pushes / pops a range if libroctx64.so loads here, stays a no-op otherwise
            pass
    finally:
        monkeypatch.delenv("DGS_ROCTX")
        importlib.reload(trace)


def test_list_promise_moves_up_one_tier_at_a_time():
    """Trainer._next_list_hint: the promise of the longest tile list (rasterizer option 6) follows the tiers of the library's sort
    launches -- 2048 entries (one launch), 57 344 = 28 segments of 2048 (three), none (four).  A view that breaks the promise moves
    the trainer ONE tier up; anything that is not a tier, and the last tier, lead to 'no promise'."""
    from dgs_amd.train import Trainer
    t = Trainer.__new__(Trainer)
    assert Trainer.LIST_HINT_TIERS == (2048, 57344, 0)
    seen = []
    t._list_hint = 2048
    for _ in range(4):
        t._list_hint = t._next_list_hint()
        seen.append(t._list_hint)
    assert seen == [57344, 0, 0, 0]
    t._list_hint = 1234
    assert t._next_list_hint() == 0
    del t._list_hint
    assert t._next_list_hint() == 0


def test_overflow_reason_bits_pick_the_next_configuration_in_one_recovery():
    """Trainer._after_overflow: the kernels leave the REASON of an overflow in the flag (kernels_preprocess.h overflow_reason: bit 0
    capacity, bit 1 promised list length, bit 2 list beyond the segmented sort's 57 344 entries), so one recovery lands on the
    configuration that fits -- a capacity overflow doubles the capacity and keeps the promise (rounds 3-4 walked the tiers first: up to
    three skipped steps), a broken promise moves the tier and keeps the capacity, a 60 000-entry list goes straight to 'no promise'."""
    from dgs_amd.train import Trainer

    def after(hint, cap, reason):
        t = Trainer.__new__(Trainer)
        t._list_hint, t._capacity = hint, cap
        t._after_overflow(reason)
        return t._list_hint, t._capacity

    assert after(2048, 1000, 1) == (2048, 2000)
    assert after(2048, 1000, 2) == (57344, 1000)
    assert after(2048, 1000, 6) == (0, 1000)
    assert after(57344, 1000, 6) == (0, 1000)
    assert after(2048, 1000, 3) == (57344, 2000)
    assert after(0, 1000, 1) == (0, 2000)
    # flag already consumed (reason 0): the promise is the first suspect, then the capacity
    assert after(2048, 1000, 0) == (57344, 1000)
    assert after(0, 1000, 0) == (0, 2000)


def test_alias_modules_share_storage_and_keep_their_own_leaves():
    """dgs_amd.train.alias_module (the lanes of Trainer(concurrent_views=True)): same class and configuration, NEW Parameter objects over
    the SAME storage, the same buffer tensors, none of the original's per-instance scratch -- and gradients that do not meet."""
    from dgs_amd.train import alias_module
    surfels, deform, cams, targets, bg = _build()
    deform._knn_seed = torch.zeros(3)            # per-instance scratch: must not travel
    for m in (surfels, deform):
        a = alias_module(m)
        assert type(a) is type(m) and "_knn_seed" not in a.__dict__
        pm, pa = dict(m.named_parameters()), dict(a.named_parameters())
        assert list(pm) == list(pa) and len(pm) > 0
        for n in pm:
            assert pa[n] is not pm[n] and pa[n].data_ptr() == pm[n].data_ptr() and pa[n].requires_grad == pm[n].requires_grad
        for (_, b), (_, c) in zip(m.named_buffers(), a.named_buffers()):
            assert b is c
    a = alias_module(surfels)
    with torch.no_grad():
        surfels._xyz.add_(1.0)                   # an in-place update of the original is the alias's value at once
    assert torch.equal(a._xyz, surfels._xyz)
    a._xyz.sum().backward()
    assert surfels._xyz.grad is None and a._xyz.grad is not None    # own autograd leaves
    assert a.active_sh_degree == surfels.active_sh_degree and a.packed_sh == surfels.packed_sh


def _reason_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        render_mod.GaussianRasterizer = OracleRasterizer
        surfels, deform, cams, targets, bg = _build()
        tr = Trainer(surfels, deform, cams, targets, bg)
        # what each rank's own overflow flag says: rank 0 nothing, rank 1 "capacity" (1); then rank 0 "promise" (2), rank 1 "beyond the segmented sort" (4 | 2)
        got = [tr._agree_reason((0, 1)[rank]), tr._agree_reason((2, 6)[rank]), tr._agree_reason(0)]
        tr.no_collectives = True
        got.append(tr._agree_reason((0, 1)[rank]))      # the diagnostic mode issues no collective: the local value stands
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_overflow_reason_is_the_or_over_the_ranks():
    """the reason bits of every rank's flag, OR-ed (three 0/1 words through a MAX all-reduce), so that all ranks pick the
    same next capacity / promise -- max(1, 2) = 2 would have dropped the capacity bit."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_reason_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get() for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][:3] == res[1][:3] == [1, 6, 0]
    assert res[0][3] == 0 and res[1][3] == 1
