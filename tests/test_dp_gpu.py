"""-m gpu: the data-parallel step on the HIP path (fused kernels, in-place gradient sinks, one flat all-reduce), two ranks
sharing the one GPU of the test box over gloo (the collective's transport is irrelevant to what is checked: replicas stay
identical, and the reduced bucket is the sum of the two views' single-process gradients, which Adam averages on the fly)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2, slots=30000)
        tr.enable_graph(capacity=40 * 20000)   # graphs 1a / 1b = forward + backward halves, eager all-reduces, graphs 2a / 2 = updates
        losses = [float(tr.step()) for _ in range(3)]
        # in-place densification between replays: every rank must perform the same surgery (summed statistics, seeded noise)
        counts = tr.densify_and_prune(max_grad=2e-5, min_opacity=0.02, extent=5.0, max_screen_size=20, seed=11)
        assert sum(counts) > 0, counts
        tr.reset_opacity()
        losses += [float(tr.step()) for _ in range(2)]
        tr.settle_shards()   # (sharded SH update: the all-gather of the last step's rows is waited for by the next READER; raw access asks for it)
        torch.cuda.synchronize()
        assert not _C.read_overflow()
        params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params] + [tr.surfels.alive.float()]).cpu()
        gp = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(gp, params)
        if rank == 0:
            q.put((all(torch.equal(gp[0], g) for g in gp), bool(torch.isfinite(params).all()), losses))
    finally:
        dist.destroy_process_group()


def test_data_parallel_graph_replay_keeps_replicas_identical():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, finite, losses = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert same and finite and all(0.0 < l < 10.0 for l in losses), losses


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        out = {}
        for bf16 in (False, True):
            for split3 in (False, True):
                tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
                tr.wire_bf16, tr.split3 = bf16, split3
                tr.enable_graph(capacity=40 * 20000)
                assert tr._split, "the split step is the one with separate slices"
                losses = [float(tr.step()) for _ in range(3)]
                tr.settle_shards()
                torch.cuda.synchronize()
                params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
                gp = [torch.zeros_like(params) for _ in range(world)]
                dist.all_gather(gp, params)
                out[(bf16, split3)] = (all(torch.equal(gp[0], g) for g in gp), bool(torch.isfinite(params).all()), losses, tr.wire_bytes_per_step(), tr.n_sh, tr._n_mid())
                del tr
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


def test_bfloat16_wire_keeps_replicas_identical_and_halves_the_big_slices():
    """Trainer.wire_bf16 (the lever for slow links, off by default): the SH slice -- and with split3 the per-surfel slice -- cross the
    wire as bfloat16.  Every rank receives the same reduced values: replicas stay BIT-identical; the bytes handed to the collectives
    move as wire_bytes_per_step states; the loss trajectory stays next to the fp32 one (it is a change of the numerics, bounded here)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for key, (same, finite, losses, wire, n_sh, n_mid) in out.items():
        bf16, split3 = key
        assert same and finite, key
        assert wire["sh"] == (2 if bf16 else 4) * n_sh, (key, wire)
        assert ("mid" in wire) == split3 and (not split3 or wire["mid"] == (2 if bf16 else 4) * n_mid), (key, wire)
        assert wire["total"] == sum(v for k, v in wire.items() if k != "total")
    n_sh, n_mid = out[(True, False)][4], out[(True, False)][5]
    assert out[(True, False)][3]["total"] == out[(False, False)][3]["total"] - 2 * n_sh
    assert out[(True, True)][3]["total"] == out[(False, True)][3]["total"] - 2 * (n_sh + n_mid)
    for split3 in (False, True):
        for a, b in zip(out[(False, split3)][2], out[(True, split3)][2]):
            assert abs(a - b) <= 2e-3 * abs(a), (out[(False, split3)][2], out[(True, split3)][2])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
        assert tr.world == world and tr.view_for(0) == rank
        tr.shard_optimizer = False          # this test looks at the REDUCED bucket: with the sharded SH update a rank holds the sum on its own rows only
        tr.opt_surfels.zero_grads = False   # keep the reduced gradients of the step for the comparison below (the update can
        tr.step()                           # clear them behind its reads)
        flat_after = tr.bucket.flat.detach().cpu().numpy().copy()
        tr.step()
        torch.cuda.synchronize()
        params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
        stats = torch.cat([tr.surfels.xyz_gradient_accum.reshape(-1), tr.surfels.denom.reshape(-1), tr.surfels.max_radii2D.float()]).cpu()
        gp = [torch.zeros_like(params) for _ in range(world)]
        gs = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(gp, params)
        dist.all_gather(gs, stats)
        if rank == 0:
            q.put((all(torch.equal(gp[0], g) for g in gp), all(torch.equal(gs[0], g) for g in gs), flat_after))
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_on_the_hip_path():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, same_stats, flat_dp = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert same and same_stats
    # single-process gradients of the two views
    import bench
    dev = torch.device("cuda:0")
    flats = []
    for view in range(world):
        tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
        tr.view_for = lambda it, v=view: v
        tr.opt_surfels.step = lambda *a, **k: None
        tr.step()
        flats.append(tr.bucket.flat.detach().cpu().clone())
    n = tr.bucket.n_grad
    # the bucket keeps the SUM over the ranks: the flat Adam kernel reads grad / world (FlatAdam.grad_scale), there is no
    # separate averaging pass on the HIP path
    expect = flats[0] + flats[1]
    got = torch.from_numpy(flat_dp)
    # fp32 atomics in the rasterizer backward: compare robustly
    err = (got - expect).abs()
    scale = expect.abs().max()
    assert float(err.max()) <= 2e-3 * float(scale), (float(err.max()), float(scale))
    assert float(err.median()) <= 1e-6 * float(scale)


def _rccl_worker(port, q):
    """One rank over the REAL backend (nccl = RCCL) with the trainer told it is one of two: every mechanism of the multi-GPU
    step runs -- communicator next to graph capture, asynchronous all-reduce of the SH slice between graph 1a and 1b, the
    MAX reduction of the radii -- except that the sums have a single contribution."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        res = {}
        for graph in (False, True):
            tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
            tr.world = 2                       # take the data-parallel code path
            assert tr._split_ok()
            if graph:
                tr.enable_graph(capacity=24 * 20000)
                assert tr._g1b is not None and tr._g2 is not None
            res[graph] = [float(tr.step()) for _ in range(4)]
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            _C.set_capacity(0)
        q.put((res[False], res[True]))
    finally:
        dist.destroy_process_group()


def test_split_step_over_rccl_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    proc = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    proc.start()
    eager, graph = q.get()
    proc.join(180)
    assert proc.exitcode == 0
    for a, b in zip(eager, graph):
        assert 0.0 < a < 10.0 and abs(a - b) <= 1e-3 * abs(a), (eager, graph)


def _split3_worker(rank, world, port, q, graph):
    """Third split of the data-parallel step (Trainer.split3): per-surfel non-SH gradients leave after the skinning backward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        out = {}
        for key, split3 in (("base", False), ("again", False), ("split3", True)):   # the same run twice: the yardstick for run-to-run noise
            tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
            tr.split3 = split3
            assert tr._split_ok()
            wire = tr.wire_bytes_per_step()
            if graph:
                tr.enable_graph(capacity=40 * 20000)
                assert (tr._g1c is not None) == split3 and (tr._g2b is not None) == split3
            losses = [float(tr.step()) for _ in range(4)]
            tr.settle_shards()
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            _C.set_capacity(0)
            params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
            gp = [torch.zeros_like(params) for _ in range(world)]
            dist.all_gather(gp, params)
            out[key] = (all(torch.equal(gp[0], g) for g in gp), losses, params, wire, tr.P, tr.bucket.flat.numel())
        if rank == 0:
            q.put({k: (v[0], v[1], v[2].numpy(), v[3], v[4], v[5]) for k, v in out.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_third_split_keeps_replicas_identical_and_moves_the_wire_bytes(graph):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_split3_worker, args=(r, world, port, q, graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    (same2, l2, p2, w2, P, nflat), (same3, l3, p3, w3, _, _), (same2b, _, p2b, _, _, _) = res["base"], res["split3"], res["again"]
    assert same2 and same3 and same2b                                   # replicas bit-identical with and without the third split
    # same training run either way (float atomics in the rasterizer's backward: equal to rounding, not bitwise)
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(l2, l3)), (l2, l3)
    # ... and the same parameters, up to what the order of the float atomics (rasterizer and skinning backward) does to a
    # gradient that is zero to rounding: Adam turns its sign into a full step, so some elements move by ~lr per step from one
    # run to the next.  The yardstick is the same configuration run twice.
    import numpy as np
    scale = float(np.abs(p2).max())
    d, noise = np.abs(p2 - p3), np.abs(p2 - p2b)
    assert float(np.median(d)) <= 1e-7 * scale
    for qt in (0.99, 0.999, 1.0):
        assert float(np.quantile(d, qt)) <= 2.0 * float(np.quantile(noise, qt)) + 1e-6 * scale, (qt, float(np.quantile(d, qt)), float(np.quantile(noise, qt)))
    # what leaves when: 48 SH floats per surfel first, then (third split) the 18 other per-surfel floats, then the rest
    assert w2["sh"] == w3["sh"] == 4 * 48 * P and "mid" not in w2
    assert w3["mid"] == 4 * (3 + 2 + 4 + 1 + 8) * P and w3["rest"] == w2["rest"] - w3["mid"] and w3["total"] == w2["total"] == 4 * (nflat + P + 4)


def _shard_worker(rank, world, port, q, graph):
    """Sharded SH update (Trainer.shard_optimizer, the default of the split step): reduce-scatter of the SH gradients, Adam on the rows
    this rank owns, all-gather of the rows into the parameter under the next step's deformation head."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        out = {}
        for key, shard in (("allreduce", False), ("again", False), ("shard", True)):   # the same run twice: the yardstick for run-to-run noise
            tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2, slots=30000)
            tr.shard_optimizer = shard
            assert tr._split_ok() and tr._shard_ok() == shard
            wire = tr.wire_bytes_per_step()
            if graph:
                tr.enable_graph(capacity=40 * 20000)
                assert (tr._g0 is not None) == shard
            losses = [float(tr.step()) for _ in range(4)]
            local = bool(tr._sh_moments_local)
            tr.settle_shards()
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            n_sh = tr.n_sh
            state = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params] + [tr.opt_surfels.exp_avg[:n_sh], tr.opt_surfels.exp_avg_sq[:n_sh]]).cpu()
            gp = [torch.zeros_like(state) for _ in range(world)]
            dist.all_gather(gp, state)
            same = all(torch.equal(gp[0], g) for g in gp)
            # density control on top of the sharded state: the surgery reads and permutes moment rows, so it must have gathered them
            counts = tr.densify_and_prune(max_grad=2e-5, min_opacity=0.02, extent=5.0, max_screen_size=20, seed=11)
            tr.sort_surfels()
            losses += [float(tr.step()) for _ in range(2)]
            tr.settle_shards()
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            _C.set_capacity(0)
            state2 = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params] + [tr.opt_surfels.exp_avg[:n_sh], tr.opt_surfels.exp_avg_sq[:n_sh],
                                                                                      tr.surfels.alive.float()]).cpu()
            gp = [torch.zeros_like(state2) for _ in range(world)]
            dist.all_gather(gp, state2)
            same2 = all(torch.equal(gp[0], g) for g in gp)
            out[key] = (same, same2, losses, state.numpy(), wire, local, sum(counts), n_sh, bool(torch.isfinite(state2).all()))
            del tr
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_sharded_sh_update_matches_the_all_reduce_step(graph):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    import numpy as np
    for key, (same, same2, losses, state, wire, local, n_densified, n_sh, finite) in res.items():
        # replicas bit-identical: parameters AND (after settle_shards) both SH moments, before and after density control + reordering
        assert same and same2 and finite, key
        assert local == (key == "shard")      # the sharded step leaves the SH moments current on their owners' rows only
        assert n_densified > 0 and all(0.0 < l < 10.0 for l in losses), (key, losses)
    # the same bytes are handed to the collectives either way (reduce-scatter + all-gather of the SH segment = its all-reduce)
    assert res["shard"][4] == res["allreduce"][4]
    # the same training run: losses equal to rounding, parameters and moments within the run-to-run noise of the all-reduce step
    l_ar, l_sh = res["allreduce"][2], res["shard"][2]
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(l_ar[:4], l_sh[:4])), (l_ar, l_sh)
    a, b, c = res["allreduce"][3], res["shard"][3], res["again"][3]
    scale = float(np.abs(a).max())
    d, noise = np.abs(a - b), np.abs(a - c)
    assert float(np.median(d)) <= 1e-7 * scale
    for qt in (0.99, 0.999, 1.0):
        assert float(np.quantile(d, qt)) <= 2.0 * float(np.quantile(noise, qt)) + 1e-6 * scale, (qt, float(np.quantile(d, qt)), float(np.quantile(noise, qt)))


def _one_rank_overflow_worker(rank, world, port, q):
    """ONE rank's frame breaks the promised list length (reason bit 1), the other rank's does not: both must recover to the SAME
    configuration (the reason was read from the rank-local flag, the clean rank took the 'reason 0' fallback)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        tr = bench.build_trainer(20000, 128, 128, dev, n_views=4, n_targets=2)
        tr.enable_graph(capacity=40 * 20000)
        cap0, hint0 = tr._capacity, tr._list_hint
        for _ in range(2):
            tr.step()
        torch.cuda.synchronize()
        if rank == 1:
            tr._oflag.fill_(1)    # what the rasterizer of THIS rank would have left: "list capacity exceeded"; rank 0's flag stays clean
        for _ in range(tr.GUARD_LAG + 3):
            tr.step()
        tr.settle_shards()
        torch.cuda.synchronize()
        cfg = torch.tensor([tr._capacity, tr._list_hint, tr.iteration, tr.overflow_recoveries], dtype=torch.int64)
        gc = [torch.zeros_like(cfg) for _ in range(world)]
        dist.all_gather(gc, cfg)
        params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
        gp = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(gp, params)
        if rank == 0:
            q.put(([g.tolist() for g in gc], all(torch.equal(gp[0], g) for g in gp), cap0, hint0))
    finally:
        dist.destroy_process_group()


def test_overflow_on_one_rank_recovers_every_rank_to_the_same_configuration():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_one_rank_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    cfgs, same, cap0, hint0 = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert cfgs[0] == cfgs[1], cfgs                 # same capacity, same promise, same iteration, same number of recoveries
    assert cfgs[0][3] >= 1 and cfgs[0][0] == 2 * cap0 and cfgs[0][1] == hint0, (cfgs, cap0, hint0)   # reason 1 on rank 1 only: capacity doubled on BOTH, promise kept
    assert same


def _concurrent_dp_worker(rank, world, port, q, graph):
    """Two ranks x two CONCURRENT views per rank (Trainer.concurrent_views): every rank folds its lanes' buckets into one before the
    step's single exchange; replicas stay bit-identical and the step trains on four views."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from diff_surfel_rasterization import _C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        out = {}
        for conc in (False, True):
            tr = bench.build_trainer(20000, 128, 128, dev, n_views=8, n_targets=2, views_per_rank=2, concurrent_views=conc)
            if graph:
                tr.enable_graph(capacity=40 * 20000)
            losses = [float(tr.step()) for _ in range(3)]
            tr.settle_shards()
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            _C.set_capacity(0)
            assert (tr._lanes is not None) == conc and int(tr.surfels.denom.max()) == 3 * 2 * world
            params = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
            gp = [torch.zeros_like(params) for _ in range(world)]
            dist.all_gather(gp, params)
            out[conc] = (all(torch.equal(gp[0], g) for g in gp), bool(torch.isfinite(params).all()), losses, params.numpy())
            del tr
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_concurrent_views_under_data_parallelism(graph):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_concurrent_dp_worker, args=(r, world, port, q, graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    import numpy as np
    for conc, (same, finite, losses, _) in res.items():
        assert same and finite and all(0.0 < l < 10.0 for l in losses), (conc, losses)
    # the same four views per step either way: sequential and concurrent lanes train alike (float atomics aside)
    for a, b in zip(res[False][2], res[True][2]):
        assert abs(a - b) <= 1e-4 * abs(a), (res[False][2], res[True][2])
    assert float(np.median(np.abs(res[False][3] - res[True][3]))) < 1e-6
