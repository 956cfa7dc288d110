"""-m gpu: the fused SSIM and KNN kernels (csrc/train_ops.hip) against the PyTorch formulations they replace
(which tests/ verify on the CPU against the reference's formula / brute force)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 64, 80), (3, 37, 53), (1, 16, 16), (3, 800, 800)])
def test_fused_ssim_matches_torch(shape):
    from dgs_amd.losses import ssim, ssim_torch
    g = torch.Generator().manual_seed(0)
    a = torch.rand(*shape, generator=g).cuda().requires_grad_(True)
    b = (a.detach() * 0.7 + 0.3 * torch.rand(*shape, generator=g).cuda()).clamp(0, 1)
    v = ssim(a, b)
    (3.0 * v).backward()
    ga = a.grad.clone()
    a.grad = None
    vr = ssim_torch(a, b)
    (3.0 * vr).backward()
    assert abs(float(v) - float(vr)) < 2e-6
    assert torch.allclose(ga, a.grad, rtol=2e-4, atol=2e-9 + 1e-4 * float(a.grad.abs().max()))


@pytest.mark.parametrize("N,M,D,K", [(1000, 64, 11, 3), (200000, 1024, 11, 3), (513, 1500, 3, 4), (77, 5, 16, 1)])
def test_knn_kernel_matches_bruteforce(N, M, D, K):
    from dgs_amd import _ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, D, generator=g).cuda()
    n = torch.randn(M, D, generator=g).cuda()
    idx = _ops.knn_indices(x, n, K)
    dsel = ((x[:, None, :] - n[idx]) ** 2).sum(-1)
    ref = torch.cat([torch.topk(((x[s:s + 8192, None, :] - n[None]) ** 2).sum(-1), K, dim=-1, largest=False).values
                     for s in range(0, N, 8192)])
    assert torch.allclose(dsel, ref, rtol=1e-5, atol=1e-6)          # same K nearest distances, ascending
    assert bool((dsel[:, 1:] >= dsel[:, :-1] - 1e-6).all())


@pytest.mark.parametrize("mode,fscale", [("box", 0.05), ("mfma", 0.05), ("mfma", 0.7)])
def test_knn_refine_is_exact_for_any_seed(mode, fscale):
    """dgs_knn_refine_mode == plain scan whatever the seed holds: last step's answer, random indices, duplicates, garbage -- for the
    3-D culling kernel and for the matrix-core filter, the latter also with hyper coordinates that dominate the distance."""
    from dgs_amd import _ops
    g = torch.Generator().manual_seed(9)
    N, M = 30001, 1000
    x = (torch.rand(N, 3, generator=g) * 2 - 1).cuda()
    f = (fscale * torch.randn(N, 8, generator=g)).cuda()
    nodes = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, 0.05 * torch.randn(M, 8, generator=g)], 1).cuda()
    want = _ops.knn_indices2(x, f, nodes, 3)
    assert torch.equal(want, _ops.knn_indices(torch.cat([x, f], 1), nodes, 3))
    moved = _ops.knn_indices2(x + 0.01 * torch.randn(N, 3, generator=g).cuda(), f, nodes, 3)
    seeds = {"previous": moved.clone(), "exact": want.clone(),
             "random": torch.randint(0, M, (N, 3), generator=g).cuda(),
             "duplicates": want[:, :1].repeat(1, 3).contiguous(),
             "garbage": torch.randint(-5, 3 * M, (N, 3), generator=g).cuda()}
    for name, seed in seeds.items():
        got = _ops.knn_indices2(x, f, nodes, 3, seed=seed, mode=mode)
        assert got.data_ptr() == seed.data_ptr()
        assert torch.equal(got, want), name


def test_knn_refine_matrix_core_filter_sizes_and_ties():
    """Node counts that are not multiples of 32 / 64, fewer coordinates than 11, points that coincide with nodes (distance 0,
    exact ties between duplicated nodes: the lower index wins like in the plain scan), a point count that leaves a partial group."""
    from dgs_amd import _ops
    g = torch.Generator().manual_seed(4)
    for N, M, H in ((4099, 77, 8), (20000, 512, 8), (3000, 200, 2), (999, 33, 5)):
        nodes = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, 0.3 * torch.randn(M, H, generator=g)], 1)
        nodes[M // 2] = nodes[3]                       # duplicated node: ties for every point
        nodes = nodes.cuda()
        x = (torch.rand(N, 3, generator=g) * 2 - 1).cuda()
        f = (0.3 * torch.randn(N, H, generator=g)).cuda()
        x[:M], f[:M] = nodes[:, :3], nodes[:, 3:]      # points ON nodes
        want = _ops.knn_indices2(x, f, nodes, 3)
        for name, seed in (("exact", want.clone()), ("shifted", torch.roll(want, 1, 0).contiguous())):
            got = _ops.knn_indices2(x, f, nodes, 3, seed=seed, mode="mfma")
            assert torch.equal(got, want), (N, M, H, name)


def test_pick_knn_refine_follows_the_hyper_coordinates():
    """ControlNodes.pick_knn_refine: 3-D culling while the K-th neighbour distance is spatial, the matrix-core filter once the
    (trained) hyper coordinates dominate it; fused deformation gives the same result under either."""
    from dgs_amd.deform import ControlNodes
    from dgs_amd.model import SurfelModel
    from dgs_amd.synthetic import make_scene
    dev = torch.device("cuda:0")
    surfels = SurfelModel(make_scene(20000, seed=3)).to(dev)
    d = ControlNodes(node_num=256, K=3, hyper_dim=8, local_frame=True).to(dev)
    d.init_from_points(surfels.get_xyz.detach(), fps=True)
    t = torch.tensor([0.3], device=dev)
    with torch.no_grad():
        a = [v.clone() for v in d.forward_assembled(surfels, t)]      # plain scan (no seed yet)
        assert not d.pick_knn_refine(surfels) and d.knn_refine_mode == "box" and d.knn_spatial_share > 0.9
        surfels.feature.add_(0.8 * torch.randn_like(surfels.feature))
        ref = [v.clone() for v in d.forward_assembled(surfels, t)]    # refine with the 3-D culling kernel
        assert d.pick_knn_refine(surfels) and d.knn_refine_mode == "mfma" and d.knn_spatial_share < 0.5
        got = d.forward_assembled(surfels, t)                         # ... with the matrix-core filter
    assert all(torch.equal(u, v) for u, v in zip(ref, got))
    assert not torch.equal(a[0], ref[0])


def test_knn_refine_block_culling_is_exact_on_sorted_inputs():
    """The 32-node block culling of dgs_knn_refine only fires when a wave's points and a block's nodes are spatially coherent:
    nodes along a Morton curve, points in the order of their nearest node (the storage order of the trainer).  Same answer as
    the plain scan, for a good seed (small search spheres, most blocks skipped), a stale one, and a node count that is not a
    multiple of 32 (partial last block)."""
    from dgs_amd import _ops
    g = torch.Generator().manual_seed(21)
    for N, M in ((40000, 1024), (20011, 1000), (5000, 77)):
        nodes3 = torch.rand(M, 3, generator=g) * 2 - 1
        q = ((nodes3 + 1) * 511.5).long().clamp(0, 1023)
        code = torch.zeros(M, dtype=torch.long)
        for b in range(10):
            for c in range(3):
                code |= ((q[:, c] >> b) & 1) << (3 * b + c)
        nodes3 = nodes3[torch.argsort(code)]
        nodes = torch.cat([nodes3, 0.01 * torch.randn(M, 8, generator=g)], 1).cuda()
        x = torch.rand(N, 3, generator=g) * 2 - 1
        x = x[torch.argsort(torch.cdist(x, nodes3).argmin(1), stable=True)].cuda()
        f = (0.01 * torch.randn(N, 8, generator=g)).cuda()
        want = _ops.knn_indices2(x, f, nodes, 3)
        stale = _ops.knn_indices2(x + 0.02 * torch.randn(N, 3, generator=g).cuda(), f, nodes, 3)
        for name, seed in (("exact", want.clone()), ("stale", stale), ("far", torch.flip(want, (0,)).contiguous())):
            got = _ops.knn_indices2(x, f, nodes, 3, seed=seed)
            assert torch.equal(got, want), (N, M, name)
            assert torch.equal(_ops.knn_indices2(x, f, nodes, 3, seed=seed.clone(), mode="mfma"), want), (N, M, name, "mfma")


def test_graph_captured_step_matches_eager():
    """The whole-step HIP graph (rasterizer in capacity mode, no host synchronisation) must train like the eager
    step: same loss trajectory (enable_graph() warms up on a snapshot, it does not train).  Parameters are compared through the losses and a robust
    statistic only: the backward's fp32 atomics make near-zero gradients flip sign under Adam."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    res = {}
    for graph in (False, True):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2)
        losses = []
        try:
            if graph:
                before = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).clone()
                tr.enable_graph(capacity=24 * 20000)
                assert torch.equal(before, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]))
                assert float(tr.opt_surfels.t) == 0.0 and float(tr.surfels.denom.sum()) == 0.0
            for _ in range(3):
                losses.append(float(tr.step()))
            torch.cuda.synchronize()
            assert not _C.read_overflow()
        finally:
            _C.set_capacity(0)
        res[graph] = (losses, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu())
    (le, pe), (lg, pg) = res[False], res[True]
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
    assert torch.isfinite(pg).all()
    assert float((pe - pg).abs().median()) < 1e-6


def test_views_per_rank_on_the_fused_path():
    """Trainer(views_per_rank=2) on the HIP path: the first view of a step STORES its gradients (nothing is cleared), the second one
    goes through the ADD flavour of every producer (SH sink, skinning backward, node-table fold, weight gradients) and reuses the
    first view's neighbour search -- the bucket the Adam kernel then reads is the SUM of the two single-view gradients (it divides by
    k itself), the statistics tail their sum; and the captured form (one graph per kind of view) trains like the eager one."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")

    def bucket_of(views_per_rank, views):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2, views_per_rank=views_per_rank)
        order = list(views)
        tr.view_for = lambda it, j=0: order[j]
        rec = []
        tr.opt_surfels.step = lambda *a, **k: rec.append(tr.bucket.flat.clone())
        tr.step()
        torch.cuda.synchronize()
        return rec[-1], tr.bucket.n_grad, float(tr.opt_surfels.grad_scale)

    f0, n, s0 = bucket_of(1, [0])
    f1, _, _ = bucket_of(1, [1])
    f0b, _, _ = bucket_of(1, [0])     # the yardstick: two runs of ONE configuration differ by the float atomics of the backward
    f1b, _, _ = bucket_of(1, [1])     # (the hyper-coordinate gradients are small differences of large terms: ~50 % noise)
    fk, _, sk = bucket_of(2, [0, 1])
    assert s0 == 1.0 and sk == 0.5
    off = 0
    tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2)
    bad = []
    for i, m in enumerate([p.numel() for p in tr.bucket.params] + [fk.numel() - n]):
        a, b = fk[off:off + m], (f0 + f1)[off:off + m]
        noise = float((f0 - f0b)[off:off + m].norm()) + float((f1 - f1b)[off:off + m].norm())
        if not float((a - b).norm()) <= 4.0 * noise + 2e-4 * float(b.norm()) + 1e-12:
            bad.append((i, m, float((a - b).norm()), noise, float(b.norm())))
        off += m
    assert not bad, bad
    res = {}
    for graph in (False, True):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2, views_per_rank=2)
        try:
            if graph:
                tr.enable_graph(capacity=24 * 20000)
            losses = [float(tr.step()) for _ in range(3)]
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            assert tr.iteration == 3 and int(tr.surfels.denom.max()) == 6     # three steps of two views
        finally:
            _C.set_capacity(0)
        res[graph] = (losses, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu())
    (le, pe), (lg, pg) = res[False], res[True]
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
    assert torch.isfinite(pg).all() and float((pe - pg).abs().median()) < 1e-6


def test_device_side_view_selection_follows_the_host_order():
    """The view of a replayed step is chosen by the graph's first node (dgs_select_row: device step counter, default order
    (i * world + rank) mod V).  It must render what the host's order asks for: in the default order without any host copy, under a
    view order of the caller's own (the override word), and after the host's iteration counter was moved (a rewind after skipped
    steps, a tool that pins the view) -- every replayed step is compared with the eager step of the same state and the same view."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")

    def run(graph, order):
        tr = bench.build_trainer(8000, 128, 160, dev, n_views=8, n_targets=8)
        views = []
        try:
            if graph:
                tr.enable_graph(capacity=40 * 8000)
                assert tr._dev_select
            if order == "custom":
                tr.view_for = lambda it: (5 * it + 3) % 8
            out = []
            for k in range(6):
                if order == "rewind" and k == 3:
                    tr.iteration = 1                      # steps 1, 2 are rendered again
                views.append(tr.view_for(tr.iteration))
                out.append(float(tr.step()))
            torch.cuda.synchronize()
            if graph:
                assert int(tr._vctr.item()) == tr._vctr_host == tr.iteration and int(tr._vovr.item()) == -1
        finally:
            _C.set_capacity(0)
        return out, views

    for order in ("default", "custom", "rewind"):
        (le, ve), (lg, vg) = run(False, order), run(True, order)
        assert ve == vg and len(set(ve)) > 2
        # different views give losses that differ in the second digit; the same view agrees to the atomics' noise
        for a, b in zip(le, lg):
            assert abs(a - b) <= 2e-4 * abs(a), (order, le, lg)


def test_capacity_overflow_is_flagged_not_fatal():
    from diff_surfel_rasterization import _C
    from gpu_utils import run_hip
    from scene_utils import small_case
    case = small_case(P=3000, H=96, W=80, seed=9, view=4)
    _C.set_capacity(100)  # far too small
    try:
        out = run_hip(case, debug=False)
        torch.cuda.synchronize()
        assert _C.read_overflow() == 1  # reason bit 0: the lists exceed the capacity (include/dgs_surfel_rasterizer.h, dgs_set_overflow_flag)
        bg = case["bg"].numpy()[:, None, None]
        assert abs(out["color"] - bg).max() == 0.0  # the frame rendered as background, nothing was written out of bounds
        assert not _C.read_overflow()  # flag was reset by the read
        # a broken promise of the longest list (option 6) says so: bit 1; together with a full buffer: both bits
        _C.set_capacity(1 << 20)
        _C.set_option(6, 4)
        out = run_hip(case, debug=False)
        assert _C.read_overflow() == 2 and abs(out["color"] - bg).max() == 0.0
        _C.set_capacity(100)
        _C.set_option(6, 4)
        run_hip(case, debug=False)
        assert _C.read_overflow() == 3
        _C.set_capacity(1 << 20)            # (set_capacity keeps no promise across a 0; here the promise is withdrawn explicitly)
        _C.set_option(6, 0)
        out = run_hip(case, debug=False)
        assert not _C.read_overflow() and abs(out["color"] - bg).max() > 0.0
    finally:
        _C.set_option(6, 0)
        _C.set_capacity(0)
    out = run_hip(case, debug=False)
    assert abs(out["color"]).max() > 0


def test_capacity_overflow_skips_the_step_and_recovers():
    """A view whose tile lists do not fit the capacity must train NOTHING (parameters, moments, statistics, step count),
    on the device, without any host read in the step; the trainer notices two steps later, doubles the capacity,
    re-captures and redoes the skipped views -- the run ends where a run with enough capacity ends."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    P, steps = 20000, 5

    def flat(tr):
        return torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).clone()

    try:
        ref = bench.build_trainer(P, 256, 256, dev, n_views=8, n_targets=2)
        ref.enable_graph(capacity=24 * P)
        ref_losses = [float(ref.step()) for _ in range(steps)]
        torch.cuda.synchronize()
        assert ref.overflow_recoveries == 0 and float(ref.opt_surfels.status[1]) == 0
        want, want_stats = flat(ref).cpu(), ref.surfels.denom.clone().cpu()

        tr = bench.build_trainer(P, 256, 256, dev, n_views=8, n_targets=2)
        start = flat(tr)
        tr.enable_graph(capacity=4000, validate=False)        # every view overflows this
        tr.step()
        tr.step()
        torch.cuda.synchronize()
        # the two overflowing steps changed nothing at all
        assert torch.equal(flat(tr), start) and float(tr.opt_surfels.t) == 0.0 and float(tr.surfels.denom.sum()) == 0.0
        assert float(tr.opt_surfels.exp_avg.abs().sum()) == 0.0 and float(tr.opt_surfels.status[1]) == 2.0
        losses = {}
        calls = 0
        while float(tr.opt_surfels.t) < steps and calls < 200:   # the host reads here only because the test wants to stop
            it = tr.iteration
            l = tr.step()
            calls += 1
            losses[it] = l.clone()
        torch.cuda.synchronize()
        assert float(tr.opt_surfels.t) == steps and tr.overflow_recoveries >= 3 and tr._capacity >= 32000
        assert tr.iteration == steps + (int(tr.opt_surfels.status[1]) - tr._skipped_seen)
        got = flat(tr).cpu()
        assert torch.isfinite(got).all() and float((got - want).abs().median()) < 1e-6
        assert float((tr.surfels.denom.cpu() - want_stats).abs().sum()) <= 4   # every view counted exactly once (a borderline radius may flip: atomics order)
    finally:
        _C.set_capacity(0)
        _C.set_overflow_flag(None)


def test_fused_lbs_matches_torch_autograd():
    """dgs_lbs_forward/backward against the PyTorch formulation of ControlNodes.forward (itself golden-pinned
    against the reference on the CPU): outputs and every gradient that leaves the deformation module."""
    from dgs_amd.deform import ControlNodes
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(3)
    N, Mn = 5000, 256
    x = ((torch.rand(N, 3, generator=g) * 2 - 1) * 1.3).cuda()
    res = {}
    for fused in (False, True):
        torch.manual_seed(1)
        m = ControlNodes(node_num=Mn, K=3, hyper_dim=8, local_frame=True).cuda()
        m.init_from_points(x)
        with torch.no_grad():
            m.nodes[:, 3:] += 0.02 * torch.randn(Mn, 8, generator=g).cuda()
            m._node_weight += 0.5 * torch.randn(Mn, 1, generator=g).cuda()
            m._node_radius += 0.2 * torch.randn(Mn, generator=g).cuda()
            for head in (m.network.gaussian_warp, m.network.gaussian_rotation, m.network.gaussian_scaling, m.network.local_rotation):
                head.weight.mul_(3e3)
        g = torch.Generator().manual_seed(3)  # same draws for both variants
        _ = torch.rand(N, 3, generator=g)
        m.use_fused, m.use_fused_mlp = fused, False
        feature = (0.05 * torch.randn(N, 8, generator=torch.Generator().manual_seed(5))).cuda().requires_grad_(True)
        mask = torch.sigmoid(torch.randn(N, 1, generator=torch.Generator().manual_seed(6))).cuda()
        t = torch.full((Mn, 1), 0.37).cuda()
        out = m(x, t, feature, mask)
        cot = [torch.randn(N, c, generator=torch.Generator().manual_seed(7 + c)).cuda() for c in (3, 4, 2)]
        loss = (out['d_xyz'] * cot[0]).sum() + (out['d_rotation'] * cot[1]).sum() + (out['d_scaling'] * cot[2]).sum()
        loss.backward()
        res[fused] = dict(out={k: v.detach() for k, v in out.items()}, feature=feature.grad.clone(), nodes=m.nodes.grad.clone(),
                          radius=m._node_radius.grad.clone(), weight=m._node_weight.grad.clone(),
                          net={n: p.grad.clone() for n, p in m.network.named_parameters()})
    a, b = res[False], res[True]

    def close(u, v, name, tol=2e-4):
        scale = max(float(u.abs().max()), 1e-12)
        err = float((u - v).abs().max())
        assert err <= tol * scale, "%s: err %.3e scale %.3e" % (name, err, scale)

    for k in a["out"]:
        close(a["out"][k], b["out"][k], k, 1e-5)
    for k in ("feature", "nodes", "radius", "weight"):
        close(a[k], b[k], "grad " + k)
    assert float(b["nodes"][:, :3].abs().max()) == 0.0
    for n in a["net"]:
        close(a["net"][n], b["net"][n], "grad net." + n, 5e-4)


def test_flat_adam_rate_pattern_matches_two_parameters():
    """One [P,16,3] SH parameter with the periodic two-rate pattern == the reference's (_features_dc, _features_rest) pair
    under torch.optim.Adam with their two rates."""
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket
    torch.manual_seed(3)
    P = 3001
    full = torch.randn(P, 16, 3, device="cuda")
    packed = torch.nn.Parameter(full.clone())
    other = torch.nn.Parameter(torch.randn(777, device="cuda"))
    dc, rest = torch.nn.Parameter(full[:, :1].clone()), torch.nn.Parameter(full[:, 1:].clone())
    other_ref = torch.nn.Parameter(other.detach().clone())
    ref = torch.optim.Adam([{'params': [dc], 'lr': 4e-3}, {'params': [rest], 'lr': 2e-4}, {'params': [other_ref], 'lr': 1e-2}], eps=1e-15)
    bucket = FlatGradBucket([other, packed])
    opt = _ops.FlatAdam(bucket.params, [1e-2, 4e-3], bucket.flat, patterns={1: (48, 3, 2e-4)})
    for it in range(4):
        g = torch.randn(P, 16, 3, device="cuda") * (0.1 + it)
        go = torch.randn(777, device="cuda")
        bucket.zero()
        packed.grad.copy_(g); other.grad.copy_(go)
        dc.grad, rest.grad, other_ref.grad = g[:, :1].clone(), g[:, 1:].clone(), go.clone()
        opt.step(); ref.step()
    assert torch.allclose(packed[:, :1], dc, rtol=1e-5, atol=1e-7)
    assert torch.allclose(packed[:, 1:], rest, rtol=1e-5, atol=1e-7)
    assert torch.allclose(other, other_ref, rtol=1e-5, atol=1e-7)


def test_sh_gradient_sink_matches_returned_gradient():
    """set_sh_grad_sink: dL/dSH written in place == the gradient autograd would have accumulated; culled rows untouched."""
    import diff_surfel_rasterization as dsr
    from gpu_utils import run_hip
    from scene_utils import small_case
    case = small_case(P=3000, H=80, W=96, seed=5, view=2, scale_mul=1.5, radius=3.0)
    g = np.random.default_rng(1)
    gc = g.standard_normal((3, 80, 96)).astype(np.float32)
    go = g.standard_normal((8, 80, 96)).astype(np.float32)
    ref = run_hip(case, gc, go)
    sink = torch.full((3000, 16, 3), 7.0, device="cuda")
    dsr.set_sh_grad_sink(sink)
    try:
        got = run_hip(case, gc, go, allow_missing_sh_grad=True)
    finally:
        dsr.set_sh_grad_sink(None)
    vis = torch.from_numpy(ref["radii"] > 0).cuda()
    assert bool((sink[~vis] == 7.0).all()) and int((~vis).sum()) > 0
    a, b = torch.from_numpy(ref["dL_dsh"]).cuda()[vis], sink[vis]
    assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
    assert got["dL_dsh"] is None


@pytest.mark.parametrize("M,per_node_t", [(1024, False), (64, True)])
def test_fused_node_mlp_matches_torch_autograd(M, per_node_t):
    """dgs_mlp_forward/backward (fp32 MFMA) against DeformMLP.forward + torch.autograd: the attribute table, every
    weight/bias gradient, in both gradient modes (returned to autograd / added into existing .grad tensors)."""
    from dgs_amd import _ops
    from dgs_amd.deform import DeformMLP
    torch.manual_seed(4)
    net = DeformMLP().cuda()
    with torch.no_grad():
        for head in (net.gaussian_warp, net.gaussian_rotation, net.gaussian_scaling, net.local_rotation):
            head.weight.normal_(0, 0.05)
            head.bias.normal_(0, 0.1)
        for lin in list(net.linear) + [net.timenet[0], net.timenet[2]]:
            lin.bias.normal_(0, 0.05)
    nodes = torch.randn(M, 11, device="cuda") * 0.8
    t = torch.rand(M, 1, device="cuda") if per_node_t else torch.full((1, 1), 0.37, device="cuda").expand(M, 1)
    cot = torch.randn(M, 13, device="cuda")
    rot_bias = torch.tensor([1.0, 0.0, 0.0, 0.0], device="cuda")

    o = net(nodes[:, :3], t)
    ref = torch.cat([o['local_rotation'] + rot_bias, o['d_xyz'], o['d_rotation'], o['d_scaling']], -1)
    (ref * cot).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in net.named_parameters()}

    def close(u, v, name, tol):
        scale = max(float(u.abs().max()), 1e-12)
        err = float((u - v).abs().max())
        assert err <= tol * scale, "%s: err %.3e scale %.3e" % (name, err, scale)

    for sink in (False, True):
        for p in net.parameters():
            p.grad = torch.full_like(p, 0.25) if sink else None
        net_out = _ops.fused_node_mlp(net, nodes, t, grad_sink=sink)
        close(ref.detach(), net_out.detach(), "attrs", 2e-5)
        (net_out * cot).sum().backward()
        for n, p in net.named_parameters():
            got = p.grad - 0.25 if sink else p.grad
            assert got is not None, n
            close(ref_grads[n], got, "grad %s (sink=%s)" % (n, sink), 2e-4)


def test_fused_deform_assembled_matches_torch_autograd():
    """ControlNodes.forward_assembled (KNN on split inputs, MFMA node MLP, skinning + surfel activations in one kernel
    per direction, gradients written or added in place) against the PyTorch formulation feeding render()."""
    from dgs_amd.deform import ControlNodes
    from dgs_amd.model import SurfelModel
    from dgs_amd.synthetic import make_scene
    N, Mn = 6001, 256
    cot_g = torch.Generator().manual_seed(11)
    cot = [torch.randn(N, c, generator=cot_g).cuda() for c in (3, 2, 4, 1)]
    res = {}
    # "coherent": the wave-sum + global-atomic backward for node-sorted storage order -- on the unsorted cloud (worst case:
    # ~64 distinct nodes per wave) and on the cloud sorted by nearest node ("sorted": the torch result is permuted to compare)
    near_perm = None
    for mode in ("torch", "fused", "sink", "coherent", "sorted", "fixed", "fixed2"):
        torch.manual_seed(1)
        scene = make_scene(N, seed=3)
        if mode == "sorted":
            scene = type(scene)(*[t_[near_perm.cpu()].contiguous() for t_ in scene])
        pc = SurfelModel(scene).cuda()
        m = ControlNodes(node_num=Mn, K=3, hyper_dim=8, local_frame=True).cuda()
        m.init_from_points(SurfelModel(make_scene(N, seed=3)).cuda()._xyz.detach())   # same nodes whatever the storage order
        if near_perm is None:
            near_perm = torch.argsort(torch.cdist(pc._xyz.detach(), m.nodes.detach()[:, :3]).argmin(1), stable=True)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            m.nodes[:, 3:] += 0.02 * torch.randn(Mn, 8, generator=g).cuda()
            m._node_weight += 0.5 * torch.randn(Mn, 1, generator=g).cuda()
            m._node_radius += 0.2 * torch.randn(Mn, generator=g).cuda()
            df = 0.05 * torch.randn(N, 8, generator=g).cuda()
            pc.feature += df[near_perm] if mode == "sorted" else df
            for head in (m.network.gaussian_warp, m.network.gaussian_rotation, m.network.gaussian_scaling, m.network.local_rotation):
                head.weight.mul_(3e3)
        params = dict(list(pc.named_parameters()) + [("deform." + n, p) for n, p in m.named_parameters()])
        t = torch.full((1,), 0.37).cuda()
        if mode == "torch":
            m.use_fused = m.use_fused_mlp = False
            dv = m(pc.get_xyz.detach(), t, pc.feature, pc.motion_mask)
            out = (pc.get_xyz + dv['d_xyz'], pc.get_scaling + dv['d_scaling'], pc.get_rotation_bias(dv['d_rotation']), pc.get_opacity)
        else:
            assert m.can_assemble(pc)
            m.grad_sink = mode == "sink"
            m.coherent_surfels = mode in ("coherent", "sorted", "fixed", "fixed2")
            m.fixed_point_tables = mode in ("fixed", "fixed2")   # the node table as 64-bit fixed-point sums (order-free integer atomics)
            if m.grad_sink:
                for p in params.values():
                    p.grad = torch.full_like(p, 0.5)
            out = m.forward_assembled(pc, t)
        cot_m = [c[near_perm] for c in cot] if mode == "sorted" else cot
        sum((o * c).sum() for o, c in zip(out, cot_m)).backward()
        if mode == "sorted":   # back to the original order for the comparison
            inv = torch.empty_like(near_perm)
            inv[near_perm] = torch.arange(N, device=near_perm.device)
            out = tuple(o[inv] for o in out)
            for n_, p_ in pc.named_parameters():
                if p_.grad is not None:
                    p_.grad = p_.grad[inv]
        grads = {}
        for n, p in params.items():
            if p.grad is None:
                assert n.startswith("_features"), n
                continue
            grads[n] = (p.grad - 0.5) if mode == "sink" and not n.startswith("_features") else p.grad.clone()
        res[mode] = ([o.detach() for o in out], grads)

    def close(u, v, name, tol):
        scale = max(float(u.abs().max()), 1e-12)
        err = float((u - v).abs().max())
        assert err <= tol * scale, "%s: err %.3e scale %.3e" % (name, err, scale)

    for n, ga in res["fixed"][1].items():   # two runs of the fixed-point table: bit-identical node (and every other) gradient
        assert torch.equal(ga, res["fixed2"][1][n]), n
    for mode in ("fused", "sink", "coherent", "sorted", "fixed"):
        for i, (a, b) in enumerate(zip(res["torch"][0], res[mode][0])):
            close(a, b, "%s out %d" % (mode, i), 2e-5)
        for n, ga in res["torch"][1].items():
            if n.startswith("_features"):
                continue
            close(ga, res[mode][1][n], "%s grad %s" % (mode, n), 5e-4)


def test_fused_train_loss_matches_separate_terms():
    """dgs_photo_* + dgs_regloss_* + dgs_loss_combine as one autograd node == L1 + D-SSIM + regulariser ops."""
    from dgs_amd import losses
    from dgs_amd.cameras import orbit_cameras
    torch.manual_seed(2)
    H, W = 96, 144
    cam = orbit_cameras(1, W, H)[0].to("cuda")
    gt = torch.rand(3, H, W, device="cuda")
    vals = {}
    for fused in (False, True):
        gen = torch.Generator(device="cuda").manual_seed(5)
        image = (0.6 * torch.rand(3, H, W, device="cuda", generator=gen) + 0.4 * gt).requires_grad_(True)
        allmap = torch.rand(8, H, W, device="cuda", generator=gen)
        allmap[5] += 2.0
        allmap.requires_grad_(True)
        losses.FUSE_PHOTOMETRIC = fused
        try:
            loss = losses.training_loss_from_allmap(image, allmap, cam, gt)
        finally:
            losses.FUSE_PHOTOMETRIC = True
        loss.backward()
        vals[fused] = (float(loss), image.grad.clone(), allmap.grad.clone())
    assert abs(vals[0][0] - vals[1][0]) <= 1e-5 * abs(vals[0][0])
    for a, b in ((vals[0][1], vals[1][1]), (vals[0][2], vals[1][2])):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12


@pytest.mark.parametrize("H,W", [(96, 144), (61, 47), (800, 800)])
def test_unit_gradient_loss_node_matches_two_pass_path(H, W):
    """unit_grad=True (what the Trainer uses): regularisers' value and gradient from ONE kernel with the depth gradient gathered
    instead of added atomically (dgs_regloss_fused), SSIM backward launched in the forward -- against the forward/backward kernel
    pairs, and (small sizes) against the PyTorch ops.  Depth plane with NaN / +-inf pixels and a background region of zeros."""
    from dgs_amd import losses
    from dgs_amd.cameras import orbit_cameras
    torch.manual_seed(2)
    cam = orbit_cameras(1, W, H)[0].to("cuda")
    gt = torch.rand(3, H, W, device="cuda")
    vals = {}
    modes = ("unit", "pair") + (("ops",) if H * W < 100_000 else ())
    for mode in modes:
        gen = torch.Generator(device="cuda").manual_seed(5)
        image = (0.6 * torch.rand(3, H, W, device="cuda", generator=gen) + 0.4 * gt).requires_grad_(True)
        allmap = torch.rand(8, H, W, device="cuda", generator=gen)
        allmap[5] += 2.0
        allmap[5, H // 3: H // 3 + 7, : W // 2] = 0.0                 # background: depth 0, degenerate normals (|v| = 0)
        allmap[5, 5, 7], allmap[5, 9, 3], allmap[5, 11, 11] = float("nan"), float("inf"), float("-inf")
        allmap[5, 0, 0], allmap[5, H - 1, W - 1] = float("nan"), float("inf")
        allmap.requires_grad_(True)
        losses.FUSE_PHOTOMETRIC = mode != "ops"
        try:
            loss = losses.training_loss_from_allmap(image, allmap, cam, gt, unit_grad=(mode == "unit"))
        finally:
            losses.FUSE_PHOTOMETRIC = True
        loss.backward(torch.ones((), device="cuda"))
        vals[mode] = (float(loss), image.grad.clone(), allmap.grad.clone())
    for other in modes[1:]:
        la, lb = vals["unit"][0], vals[other][0]
        if la == la or lb == lb:          # (-inf depth makes the reference's own loss non-finite on some layouts)
            assert abs(la - lb) <= 1e-5 * abs(lb), (other, la, lb)
        for a, b in ((vals["unit"][1], vals[other][1]), (vals["unit"][2], vals[other][2])):
            fin = torch.isfinite(b)
            assert torch.equal(torch.isfinite(a), fin), other
            # the depth gradient is a sum of four neighbour terms that cancel at the rim of the degenerate region (values up to
            # ~170 there): gather order vs atomic order differ by a few ulp of the largest TERM
            assert float((a[fin] - b[fin]).abs().max()) <= 5e-5 * float(b[fin].abs().max()) + 1e-12, other
    assert torch.equal(vals["unit"][1], vals["pair"][1])       # the SSIM backward is the same kernel on the same inputs


def test_stored_gradients_equal_cleared_and_added_ones():
    """Trainer.store_grads (no gradient-bucket fill: every producer overwrites) against the cleared bucket the producers add to:
    after three steps from the same state the bucket holds the same gradients segment by segment -- a producer that still added
    into the never-cleared buffer would show up as a multiple -- and the parameters agree like two runs of one configuration."""
    import bench
    dev = torch.device("cuda:0")
    res = {}
    for store in (True, False, False):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=4, n_targets=2)
        tr.store_grads = store
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        assert tr._store_now == store
        res.setdefault(store, []).append((tr.bucket.flat.clone(), [p.detach().clone() for p in tr.bucket.params], [p.numel() for p in tr.bucket.params]))
    (fa, pa, sizes), (fb, pb, _), (fc, pc, _) = res[True][0], res[False][0], res[False][1]
    off = 0
    for i, n in enumerate(sizes + [fa.numel() - sum(sizes)]):
        a, b, c = fa[off:off + n], fb[off:off + n], fc[off:off + n]
        off += n
        noise = float((b - c).norm()) + 1e-12 * float(b.norm()) + 1e-30        # two cleared runs: float atomics + three Adam steps apart
        # (1e-3 of the segment's norm: a producer that added would be off by a multiple of it; the 4-element segments -- head biases --
        # have been seen 4e-4 apart between two runs whose own pair happened to agree to 6e-5: one pair's noise is a draw, not a bound)
        assert float((a - b).norm()) <= 4.0 * noise + 1e-3 * float(b.norm()), (i, n, float((a - b).norm()), noise, float(b.norm()))
    # Parameters after three Adam steps: Adam turns a gradient that is zero to rounding into a full step of either sign, so a few
    # elements per run land 2 lr apart between ANY two runs (float atomics); the maximum difference of one pair of runs is a draw
    # from that tail, not a yardstick for another pair (tools/diag/store_probe.py: pairwise maxima of one configuration spread over
    # a factor of ten).  The 99.9th percentile is stable; a producer that added into the uncleared bucket would move every element.
    q = lambda x: float(torch.quantile(x.abs().reshape(-1)[:4_000_000].float(), 0.999))
    for a, b, c in zip(pa, pb, pc):
        assert q(a - b) <= 4.0 * q(b - c) + 1e-6, (tuple(a.shape), q(a - b), q(b - c))
        assert float((a - b).abs().max()) <= 50.0 * float((b - c).abs().max()) + 1e-4


def test_fused_step_matches_unfused_step():
    """Trainer with every fused stage (assembled deformation, MFMA node MLP with gradient sinks, fused loss, fused
    statistics) against the same Trainer on the PyTorch formulations: loss trajectory and statistics."""
    import bench
    from dgs_amd import losses
    dev = torch.device("cuda:0")
    res = {}
    for fused in (False, True):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=4, n_targets=2, packed_sh=fused)
        tr.fuse_deform = fused
        tr.deform.use_fused_mlp = fused
        losses.FUSE_PHOTOMETRIC = fused
        try:
            ls = [float(tr.step()) for _ in range(4)]
        finally:
            losses.FUSE_PHOTOMETRIC = True
        res[fused] = (ls, tr.surfels.xyz_gradient_accum.clone(), tr.surfels.denom.clone(), tr.surfels.max_radii2D.clone())
    for a, b in zip(res[False][0], res[True][0]):
        assert abs(a - b) <= 2e-4 * abs(a), (res[False][0], res[True][0])
    assert torch.equal(res[False][2], res[True][2])
    # four Adam steps on fp32-atomics gradients: trajectories agree statistically, not element by element
    assert float((res[False][1] - res[True][1]).abs().median()) <= 1e-2 * float(res[False][1].abs().median())
    assert int((res[False][3] != res[True][3]).sum()) <= 20


def test_flat_adam_matches_torch_adam():
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket
    torch.manual_seed(0)
    shapes = [(1000, 3), (1000, 15, 3), (7,), (256, 93), (1, 1), (5000,)]
    lrs = [8e-6, 2e-4, 0.05, 1.6e-6, 0.01, 0.002]
    a = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    bucket = FlatGradBucket(a)
    flat = _ops.FlatAdam(a, lrs, bucket.flat)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    for it in range(4):
        g = [torch.randn(*s, generator=torch.Generator().manual_seed(10 * it + i)).cuda() * (10.0 ** (i - 3)) for i, s in enumerate(shapes)]
        for p, q, gi in zip(a, b, g):
            p.grad.copy_(gi)
            q.grad = gi.clone()
        flat.step()
        ref.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), float((p - q).abs().max())


def test_flat_adam_late_parameters_start_their_own_step_count():
    """The reference un-detaches the deformation after the warm-up (train_gui.py:281-285): until then `feature` and the deformation
    parameters have no gradient and torch.optim.Adam skips them, so their per-parameter step count starts at 1 when they join.
    FlatAdam keeps ONE device counter; set_origin gives the late segments t - t0 in the bias corrections (dgs_adam_step_origin).
    Without it the first updates of moments that start from zero run at 1 / (1 - beta1^t) ~ 1 instead of 10 x the raw moment."""
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket
    torch.manual_seed(1)
    shapes = [(500, 3), (500, 8), (64, 30)]
    lrs = [1e-3, 2e-3, 5e-4]
    a = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    bucket = FlatGradBucket(a)
    flat = _ops.FlatAdam(a, lrs, bucket.flat)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    warm = 30
    for it in range(warm + 6):
        late = it >= warm
        if it == warm:
            flat.set_origin(1, None, float(flat.t.item()))
        g = [torch.randn(*s, generator=torch.Generator().manual_seed(100 * it + i)).cuda() for i, s in enumerate(shapes)]
        for i, (p, q, gi) in enumerate(zip(a, b, g)):
            p.grad.copy_(gi)
            q.grad = gi.clone() if (i == 0 or late) else None   # torch skips parameters without a gradient
        flat.step(0, None if late else 1)
        ref.step()
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), (it, float((p - q).abs().max()))


def test_two_trainers_interleaved_on_one_device_do_not_share_state():
    """Per-trainer state (round 4): the SH gradient sink is keyed by the trainer's own SH parameter and the persistent node table of
    the coherent skinning backward belongs to the trainer's deformation module.  Two trainers on ONE device, stepped alternately,
    must train exactly as each does alone -- with the per-device state of round 3 the second trainer's backward removed or
    replaced the first one's sink (dL/dSH then came back through autograd into a bucket that store mode never clears)."""
    import bench
    import diff_surfel_rasterization as dsr
    dev = torch.device("cuda:0")

    def make(P):
        tr = bench.build_trainer(P, 96, 128, dev, n_views=8, n_targets=2)
        tr.set_regime(warmup=False, lambda_normal=0.05, lambda_dist=100.0)
        return tr

    def run_alone(P, n):
        tr = make(P)
        losses = [float(tr.step()) for _ in range(n)]
        return losses, tr.surfels._features.detach().clone(), tr.deform.nodes.detach().clone()

    la, fa, na = run_alone(3000, 4)
    lb, fb, nb = run_alone(5000, 4)
    A, B = make(3000), make(5000)
    ia, ib = [], []
    for _ in range(4):
        ia.append(float(A.step()))
        ib.append(float(B.step()))
    assert not dsr._SINKS, "a trainer left its sink behind"
    # float atomics in the backward: steps agree to rounding, not bit for bit
    for x, y in zip(la + lb, ia + ib):
        assert abs(x - y) <= 2e-4 * abs(x), (la, ia, lb, ib)
    for alone, both in ((fa, A.surfels._features), (fb, B.surfels._features), (na, A.deform.nodes), (nb, B.deform.nodes)):
        d = float((alone - both).abs().max())
        assert d <= 1e-4 * max(float(alone.abs().max()), 1e-6), d


def test_fused_reg_loss_matches_render_postprocess():
    """dgs_regloss_forward/backward against the PyTorch post-processing of dgs_amd.render.render + the two
    regulariser terms of training_loss, on the rasterizer's real allmap."""
    import bench
    from dgs_amd.losses import training_loss, training_loss_from_allmap
    from dgs_amd.render import render
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(20000, 200, 264, dev, n_views=8, n_targets=2)
    s, d = tr.surfels, tr.deform
    cam, gt = tr.cameras[3], tr.targets[1]
    grads = []
    vals = []
    for fused in (False, True):
        tr.bucket.zero()
        dv = d(s.get_xyz.detach(), d.expand_time(cam.fid), s.feature, s.motion_mask)
        pkg = render(cam, s, tr.bg, dv['d_xyz'], dv['d_rotation'], dv['d_scaling'], postprocess=not fused)
        pkg["allmap"].retain_grad()
        loss = training_loss_from_allmap(pkg["render"], pkg["allmap"], cam, gt) if fused else training_loss(pkg, gt)
        loss.backward()
        vals.append(float(loss))
        grads.append(torch.nan_to_num(pkg["allmap"].grad.clone(), 0.0, 0.0, 0.0))
    assert abs(vals[0] - vals[1]) <= 1e-5 * abs(vals[0]), vals
    a, b = grads
    # channels 0 and 1 carry 0 * inf = NaN in the PyTorch path at empty pixels (expected depth / alpha with
    # depth_ratio = 1); the rasterizer ignores them, the fused path never produces them
    for c in (2, 3, 4, 5, 6, 7):
        scale = max(float(a[c].abs().max()), 1e-20)
        assert float((a[c] - b[c]).abs().max()) <= 2e-4 * scale, (c, float((a[c] - b[c]).abs().max()), scale)


def test_flat_adam_device_schedule_matches_host_schedule():
    """dgs_adam_step_sched: the exponential rate computed in the kernel from the device step counter against torch Adam whose
    group rates are set on the host before every step (what the reference does, one step later: step t sees schedule(t-1))."""
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket, expon_lr
    torch.manual_seed(1)
    shapes = [(700, 3), (300, 16, 3), (256, 93), (11,)]
    lrs = [8e-4, 0.0025, 8e-4, 0.05]
    sched = {0: (8e-6, 50.0), 2: (1.6e-6, 20.0)}      # short horizons: the clip at max_steps is exercised too
    a = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    bucket = FlatGradBucket(a)
    flat = _ops.FlatAdam(a, lrs, bucket.flat, patterns={1: (48, 3, 0.0025 / 20)}, schedules=sched)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    mask = (torch.arange(300 * 48).cuda() % 48 >= 3).view(300, 16, 3)
    for it in range(30):
        g = [torch.randn(*s, generator=torch.Generator().manual_seed(100 * it + i)).cuda() * 1e-2 for i, s in enumerate(shapes)]
        for i, (p, q, gi) in enumerate(zip(a, b, g)):
            p.grad.copy_(gi)
            q.grad = gi.clone()
        for i, (lr_final, n) in sched.items():
            ref.param_groups[i]["lr"] = expon_lr(it, lrs[i], lr_final, n)
        before = b[1].detach().clone()
        flat.step()
        ref.step()
        with torch.no_grad():   # the reference keeps the two SH rates in two parameters: emulate the second rate
            b[1].copy_(torch.where(mask, before + (b[1] - before) / 20, b[1]))
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), (i, float((p - q).abs().max()))


def test_flat_adam_grad_scale_is_the_mean_of_a_summed_bucket():
    """grad_scale = 1 / world on a bucket holding the sum over `world` ranks == plain Adam on the mean."""
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket
    torch.manual_seed(2)
    shapes = [(500, 16, 3), (500, 3), (33,)]
    a = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ba, bb = FlatGradBucket(a), FlatGradBucket(b)
    oa = _ops.FlatAdam(a, [1e-3] * 3, ba.flat, patterns={0: (48, 3, 5e-5)})
    ob = _ops.FlatAdam(b, [1e-3] * 3, bb.flat, patterns={0: (48, 3, 5e-5)})
    oa.grad_scale = 0.25
    for it in range(5):
        g = torch.randn(ba.flat.numel(), generator=torch.Generator().manual_seed(it)).cuda()
        ba.flat.copy_(4.0 * g)       # "sum over four ranks"
        bb.flat.copy_(g)
        oa.step(0, 1)                # split launches as in the data-parallel step
        oa.step(1, None, advance=False)
        ob.step()
    for p, q in zip(a, b):
        assert torch.equal(p, q)     # 4 g * 0.25 is exact in binary floating point


def test_flat_adam_clears_the_gradients_behind_its_reads():
    """FlatAdam.zero_grads (dgs_adam_step_zero): optimizer.step() + zero_grad() in one pass -- same update, every gradient element
    of the launched range zero afterwards, the others untouched; on a guarded step that is skipped the gradients are cleared
    too (the next backward must not add to a stale buffer) while parameters, moments and the step count stay."""
    from dgs_amd import _ops
    from dgs_amd.train import FlatGradBucket
    torch.manual_seed(4)
    shapes = [(700, 16, 3), (700, 3), (1030,)]
    a = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ba, bb = FlatGradBucket(a, extra=64), FlatGradBucket(b, extra=64)
    oa = _ops.FlatAdam(a, [1e-3] * 3, ba.flat)
    ob = _ops.FlatAdam(b, [1e-3] * 3, bb.flat)
    oa.zero_grads = True
    skip = torch.zeros(1, dtype=torch.int32, device="cuda")
    oa.skip = skip
    for it in range(3):
        g = torch.randn(ba.flat.numel(), generator=torch.Generator().manual_seed(it)).cuda()
        ba.flat.copy_(g)
        bb.flat.copy_(g)
        oa.step(0, 1)
        assert not a[0].grad.any() and torch.equal(a[1].grad, b[1].grad) and torch.equal(a[2].grad, b[2].grad)
        oa.step(1, None, advance=False)
        ob.step()
        assert not ba.flat[:ba.n_grad].any() and torch.equal(ba.extra, g[ba.n_grad:])   # the tail is not a gradient
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    before = [p.detach().clone() for p in a], oa.exp_avg.clone(), oa.exp_avg_sq.clone(), float(oa.t)
    skip.fill_(1)
    ba.flat.copy_(torch.randn(ba.flat.numel(), generator=torch.Generator().manual_seed(9)).cuda())
    oa.step()
    torch.cuda.synchronize()
    assert not ba.flat[:ba.n_grad].any()
    assert all(torch.equal(p, q) for p, q in zip(a, before[0])) and torch.equal(oa.exp_avg, before[1]) and torch.equal(oa.exp_avg_sq, before[2])
    assert float(oa.t) == before[3]


def test_trainer_without_the_bucket_fill_trains_the_same():
    """Trainer with FlatAdam.zero_grads: the update clears the gradients, no fill is launched in front of the next step
    (Trainer._bucket_clean) -- same training as with the fill in front of the step (the default)."""
    import bench
    dev = torch.device("cuda:0")
    res = {}
    for fused_zero in (False, True):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2)
        tr.opt_surfels.zero_grads = fused_zero
        losses = [float(tr.step()) for _ in range(4)]
        torch.cuda.synchronize()
        assert tr._bucket_clean == fused_zero
        if fused_zero:
            assert not tr.bucket.flat[:tr.bucket.n_grad].any()
        res[fused_zero] = (losses, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu())
    for x, y in zip(res[False][0], res[True][0]):
        assert abs(x - y) <= 1e-4 * abs(x), (res[False][0], res[True][0])
    assert float((res[False][1] - res[True][1]).abs().median()) < 1e-6


def test_graph_capture_checks_every_view_not_only_the_first():
    """enable_graph(validate=True) renders all views (forward only) before the run: a promise of the longest tile list that only
    a LATER view breaks is withdrawn at capture, not by the step guard in the middle of training (which skips a step and
    re-captures).  View 0 is zoomed far in (a handful of surfels, short lists: the promise of 256 entries holds for it), the others are not."""
    import bench
    from dgs_amd.cameras import orbit_cameras
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    res = {}
    for validate in (False, True):
        tr = bench.build_trainer(20000, 128, 128, dev, n_views=8, n_targets=2)
        tr.cameras[0] = orbit_cameras(8, 128, 128, fov=0.02)[0].to(dev)   # zoomed far in: a handful of surfels in view
        tr._list_hint = 256
        try:
            before = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).clone()
            tr.enable_graph(capacity=24 * 20000, validate=validate)
            assert torch.equal(before, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]))   # capture trains nothing
            hint_after_capture = tr._list_hint
            losses = [float(tr.step()) for _ in range(12)]
            torch.cuda.synchronize()
            res[validate] = (hint_after_capture, tr.overflow_recoveries, float(tr.opt_surfels.t), losses)
        finally:
            _C.set_capacity(0)
    # without the check: captured with the promise, the guard finds out at the first other view and re-captures
    assert res[False][0] == 256 and res[False][1] >= 1
    # with it: found at capture, nothing skipped, every step trained
    assert res[True][0] == 0 and res[True][1] == 0 and res[True][2] == 12.0
    assert all(0.0 < l < 10.0 for l in res[True][3])
