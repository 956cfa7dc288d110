"""-m gpu: Trainer(views_per_rank=2, concurrent_views=True) -- the two views of a step in flight at the same time, a lane each
(alias modules over the same parameter storage, own gradient bucket, own rasterizer context, own stream / captured graph), one update
that reads the sum of the two buckets (dgs_adam_step_sum2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_alias_modules_share_values_and_keep_their_own_gradients():
    import bench
    from dgs_amd.train import alias_module
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(2000, 64, 64, dev, n_views=4, n_targets=1)
    for m in (tr.surfels, tr.deform):
        a = alias_module(m)
        pm, pa = dict(m.named_parameters()), dict(a.named_parameters())
        assert list(pm) == list(pa) and len(pm) > 0
        for n in pm:
            assert pa[n] is not pm[n] and pa[n].data_ptr() == pm[n].data_ptr() and pa[n].grad is None
        for (n, b), (_, c) in zip(m.named_buffers(), a.named_buffers()):
            assert b is c
        with torch.no_grad():
            next(iter(pm.values())).add_(1.0)
        assert torch.equal(next(iter(pa.values())), next(iter(pm.values())))


def test_concurrent_views_bucket_is_the_sum_of_the_single_view_gradients_and_graph_equals_eager():
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")

    def bucket_of(views_per_rank, views, concurrent=False):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2, views_per_rank=views_per_rank, concurrent_views=concurrent)
        order = list(views)
        tr.view_for = lambda it, j=0: order[j]
        rec = []

        def grab(*a, **k):
            g2 = tr.opt_surfels.grad2
            rec.append(tr.bucket.flat.clone() if g2 is None else tr.bucket.flat + g2)
        tr.opt_surfels.step = grab
        tr.step()
        torch.cuda.synchronize()
        assert (tr._lanes is not None) == concurrent
        return rec[-1], tr.bucket.n_grad, float(tr.opt_surfels.grad_scale)

    f0, n, s0 = bucket_of(1, [0])
    f1, _, _ = bucket_of(1, [1])
    f0b, _, _ = bucket_of(1, [0])     # the yardstick: two runs of ONE configuration differ by the float atomics of the backward
    f1b, _, _ = bucket_of(1, [1])
    fc, _, sc = bucket_of(2, [0, 1], concurrent=True)
    assert s0 == 1.0 and sc == 0.5
    tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2)
    off, bad = 0, []
    for i, m in enumerate([p.numel() for p in tr.bucket.params] + [fc.numel() - n]):
        a, b = fc[off:off + m], (f0 + f1)[off:off + m]
        noise = float((f0 - f0b)[off:off + m].norm()) + float((f1 - f1b)[off:off + m].norm())
        if not float((a - b).norm()) <= 4.0 * noise + 2e-4 * float(b.norm()) + 1e-12:
            bad.append((i, m, float((a - b).norm()), noise, float(b.norm())))
        off += m
    assert not bad, bad
    res = {}
    for mode in ("sequential", "eager", "graph"):
        tr = bench.build_trainer(20000, 256, 256, dev, n_views=8, n_targets=2, views_per_rank=2, concurrent_views=mode != "sequential")
        try:
            if mode == "graph":
                tr.enable_graph(capacity=24 * 20000)
                assert tr._glanes is not None and (tr._gall is not None or len(tr._glanes) == 2)
            losses = [float(tr.step()) for _ in range(3)]
            torch.cuda.synchronize()
            assert not _C.read_overflow()
            assert tr.iteration == 3 and int(tr.surfels.denom.max()) == 6     # three steps of two views
        finally:
            _C.set_capacity(0)
        res[mode] = (losses, torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu())
    for other in ("eager", "graph"):
        for a, b in zip(res["sequential"][0], res[other][0]):
            assert abs(a - b) <= 1e-4 * abs(a), (other, res["sequential"][0], res[other][0])
        assert torch.isfinite(res[other][1]).all() and float((res["sequential"][1] - res[other][1]).abs().median()) < 1e-6


def test_concurrent_lanes_follow_in_place_densification_and_are_rebuilt_after_growth():
    """The lanes alias the trainer's parameter STORAGE: in-place density control and reordering are visible to them at once (same
    captured graph keeps replaying); growth replaces the parameters, so the lanes are rebuilt over the new ones and re-captured."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(20000, 128, 128, dev, n_views=8, n_targets=2, slots=40000, views_per_rank=2, concurrent_views=True)
    try:
        tr.enable_graph(capacity=40 * 40000)
        lanes0 = tr._lanes
        losses = [float(tr.step()) for _ in range(3)]
        counts = tr.densify_and_prune(max_grad=2e-5, min_opacity=0.02, extent=5.0, max_screen_size=20, seed=3)
        assert sum(counts) > 0 and tr.P == 40000        # (room enough: no growth here)
        tr.sort_surfels()
        assert tr._lanes is lanes0                      # nothing was re-allocated: same lanes, same graph
        for ln in tr._lanes:
            for p, q in zip(ln.bucket.params, tr.bucket.params):
                assert p.data_ptr() == q.data_ptr() and p.grad.data_ptr() != q.grad.data_ptr()
        losses += [float(tr.step()) for _ in range(2)]
        tr.grow(45056)
        assert tr.P == 45056 and tr._lanes is not lanes0 and tr._lanes[0].P == 45056   # rebuilt over the new parameters by the re-capture
        losses += [float(tr.step()) for _ in range(2)]
        torch.cuda.synchronize()
        assert not _C.read_overflow() and all(np.isfinite(l) and 0.0 < l < 10.0 for l in losses), losses
        assert int(tr.surfels.denom.max()) <= 2 * 7 and all(bool(torch.isfinite(p).all()) for p in tr.bucket.params)
        # back to one view per step (bench.py's lever windows do this): the module gets its inner forks back
        assert tr.deform.overlap_streams is False
        tr.views_per_rank, tr.concurrent_views = 1, False
        tr._graph = None
        tr.enable_graph(tr._capacity)
        assert tr.deform.overlap_streams is True and tr._glanes is None
        assert np.isfinite(float(tr.step()))
    finally:
        _C.set_capacity(0)


def test_fit_with_concurrent_lanes_learns(tmp_path):
    """fit() with two views of every step in flight at once: the lanes go through everything a run does -- the stages (re-captures), the
    SH degree steps, in-place densification and reordering, opacity resets, growth (lanes rebuilt) -- and the scene is learnt."""
    import math
    from dgs_amd import io as dio
    from dgs_amd.fit import fit
    from dgs_amd.render import render
    from dgs_amd.synthetic import write_dynamic_dnerf
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    data = str(tmp_path / "scene")
    write_dynamic_dnerf(data, n_train=60, n_test=8, H=200, W=200, device=dev)
    test = dio.load_dnerf(data, num_pts=20_000)["test"]
    bg = torch.zeros(3, device=dev)

    def heldout_psnr(tr):
        vals = []
        with torch.no_grad():
            for f in test:
                cam = f.camera.to(dev)
                dv = tr.deform(tr.surfels.get_xyz.detach(), tr.deform.expand_time(cam.fid), tr.surfels.feature, tr.surfels.motion_mask)
                img = render(cam, tr.surfels, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])["render"]
                vals.append(-10.0 * math.log10(max(float(((img.clamp(0, 1).cpu() - f.image) ** 2).mean()), 1e-12)))
        return float(np.mean(vals))

    probes, lanes_seen = {}, []

    def hook(it, tr):
        if it in (1, 3000):
            probes[it] = heldout_psnr(tr)
        if it % 500 == 0:
            lanes_seen.append((tr._lanes is not None and len(tr._lanes) == 1, getattr(tr, "_gall", None) is not None, id(tr._lanes[0])))
    try:
        tr, losses = fit(data, str(tmp_path / "model"), iterations=3000, device=dev, num_pts=20_000, node_num=256, seed=0, warm_up=900,
                         regularize_from=2400, on_iteration=hook, views_per_rank=2, concurrent_views=True)
        torch.cuda.synchronize()
    finally:
        _C.set_capacity(0)
    print("concurrent lanes: held-out PSNR", {k: round(v, 2) for k, v in probes.items()}, "live", tr.surfels.num_surfels, "slots", tr.P,
          "distinct lane objects over the run", len({x[2] for x in lanes_seen}))
    assert all(a and b for a, b, _ in lanes_seen), lanes_seen       # the one-graph lanes ran at every probe
    assert len({x[2] for x in lanes_seen}) >= 2                      # ... and were rebuilt along the way (stages, SH degree, growth)
    assert probes[3000] >= probes[1] + 8.0 and probes[3000] >= 17.0, probes
    losses = np.asarray(losses)
    assert np.isfinite(losses).all() and losses[-300:].mean() < 0.6 * losses[:300].mean()
    assert all(bool(torch.isfinite(p).all()) for p in tr.bucket.params)
