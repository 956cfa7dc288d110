"""-m gpu: does the train step LEARN?  Every gradient is parity-checked in isolation elsewhere; this is the composition: a hidden
dynamic scene (dgs_amd.synthetic.DynamicTruth: a bobbing sphere and a swinging plate) is rendered into a D-NeRF-format dataset
with this package's own rasterizer, and fit() -- reader, random point cloud, the reference's stages (deformation detached during
warm-up, train_gui.py:282-285; regularisers off until iteration 8000, :292-293), learning-rate schedules, densification and
opacity resets on the reference's intervals, whole-step HIP graphs -- has to recover it: PSNR on HELD-OUT views and times must rise
by a stated margin, and the loss must fall."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return -10.0 * math.log10(max(float(((a - b) ** 2).mean()), 1e-12))


def test_fit_recovers_a_hidden_dynamic_scene(tmp_path):
    from dgs_amd import io as dio
    from dgs_amd.fit import fit
    from dgs_amd.render import render
    from dgs_amd.synthetic import write_dynamic_dnerf
    dev = torch.device("cuda:0")
    data = str(tmp_path / "scene")
    write_dynamic_dnerf(data, n_train=60, n_test=12, H=200, W=200, device=dev)
    test = dio.load_dnerf(data, num_pts=20_000)["test"]
    assert len(test) == 12 and test[0].image.shape == (3, 200, 200)
    bg = torch.zeros(3, device=dev)

    def heldout_psnr(tr):
        vals = []
        with torch.no_grad():
            for f in test:
                cam = f.camera.to(dev)
                dv = tr.deform(tr.surfels.get_xyz.detach(), tr.deform.expand_time(cam.fid), tr.surfels.feature, tr.surfels.motion_mask)
                img = render(cam, tr.surfels, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])["render"]
                vals.append(_psnr(img.clamp(0, 1).cpu(), f.image))
        return float(np.mean(vals))

    probes = {}
    # deterministic=True (round 5): order-free sums instead of float atomics -- the run is the same trajectory every time, so the
    # margins below are checked on ONE known outcome instead of on a draw from a 5-dB spread (which failed this test once in ~15 runs)
    tr, losses = fit(data, str(tmp_path / "model"), iterations=9000, device=dev, num_pts=20_000, node_num=256, seed=0,
                     deterministic=os.environ.get("DGS_LEARNING_TEST_ATOMICS", "0") != "1",
                     on_iteration=lambda it, t: probes.__setitem__(it, heldout_psnr(t)) if it in (1, 3000, 6000, 9000) else None)
    tr.set_deterministic(False)
    print("held-out PSNR by iteration:", {k: round(v, 2) for k, v in sorted(probes.items())})
    losses = np.asarray(losses)
    blocks = losses[:9000].reshape(18, 500).mean(1)
    print("mean loss per 500 iterations:", np.round(blocks, 4))
    # ---- the stated margins
    # (the run is chaotic -- float atomics, 9000 Adam steps, densification decisions: seven runs of this test on one code state
    # ended between 23.3 and 28.2 dB, the level of an untrained model is 7.7; the test below shows that the spread is the order of the
    # float atomics and nothing else)
    assert probes[9000] >= probes[1] + 10.0, probes                    # held-out PSNR rises by >= 10 dB over the run
    assert probes[9000] >= 21.0, probes                                # ... to a level at which the scene is recognisably recovered
    assert probes[6000] >= probes[3000] + 1.0, probes                  # the deformation (trained from iteration 3000) adds to the static fit
    # the loss falls: block means decrease over the run (opacity resets at 3000 / 6000 / 9000 and the regularisers switched on at
    # 8001 may bump one block), and the exponential moving average ends far below where it started
    ema, e = [], losses[0]
    for l in losses:
        e = 0.99 * e + 0.01 * l
        ema.append(e)
    assert ema[7999] <= 0.4 * ema[200] and ema[-1] <= 0.6 * ema[200], (ema[200], ema[7999], ema[-1])   # (the regularisers join the loss at 8001)
    assert blocks[5] < 0.6 * blocks[0] and blocks[15] < blocks[5], blocks
    assert sum(1 for a, b in zip(blocks[:-1], blocks[1:]) if b > 1.05 * a) <= 4, blocks
    assert os.path.exists(os.path.join(str(tmp_path / "model"), "point_cloud/iteration_9000/point_cloud.ply"))


def test_the_run_to_run_spread_is_the_float_atomics(tmp_path, monkeypatch):
    """The learning test above ends anywhere between 23 and 28 dB from run to run.  Is that chaos seeded by the order of float
    atomics, or a race?  With every atomic sum replaced by an ordered one -- the deterministic backward blend (rasterizer option
    7) and the skinning backward's per-workgroup tables instead of its wave-level atomics (coherent_surfels off) -- two fits with
    the same seed, through warm-up, densification, pruning and an opacity reset, must agree BIT FOR BIT in every loss and every
    parameter; a third fit in the default configuration (atomics, captured step) starts from the same state and drifts away."""
    from diff_surfel_rasterization import _C
    from dgs_amd import train as dtrain
    from dgs_amd.fit import fit
    from dgs_amd.synthetic import write_dynamic_dnerf
    dev = torch.device("cuda:0")
    data = str(tmp_path / "scene")
    write_dynamic_dnerf(data, n_train=24, n_test=2, H=128, W=128, device=dev)
    kw = dict(iterations=900, device=dev, num_pts=6000, node_num=128, seed=0, warm_up=300, regularize_from=600, densify_from=200,
              opacity_reset_interval=500)

    def run(tag, deterministic):
        if deterministic:
            sort = dtrain.Trainer.sort_surfels

            def sort_ordered(self):
                out = sort(self)
                self.deform.coherent_surfels = False     # per-workgroup LDS tables + ordered reduction instead of wave-level float atomics
                return out
            monkeypatch.setattr(dtrain.Trainer, "sort_surfels", sort_ordered)
            _C.set_option(7, 1)
        try:
            tr, losses = fit(data, str(tmp_path / tag), graph=not deterministic, **kw)
        finally:
            _C.set_option(7, 0)
            monkeypatch.undo()
        alive = tr.surfels.alive
        state = [t.detach()[alive].clone() for t in (tr.surfels._xyz, tr.surfels._features, tr.surfels._opacity, tr.surfels._scaling, tr.surfels._rotation)]
        state += [tr.deform.nodes.detach().clone()] + [p.detach().clone() for p in tr.deform.network.parameters()]
        return np.asarray(losses, dtype=np.float64), state

    la, sa = run("a", True)
    lb, sb = run("b", True)
    assert len(la) == 900 and np.isfinite(la).all()
    assert np.array_equal(la, lb), int(np.argmax(la != lb))
    assert len(sa) == len(sb) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(sa, sb))
    lc, _ = run("c", False)
    assert lc[0] == pytest.approx(la[0], rel=1e-4)                       # the same first step, up to the summation order ...
    assert np.abs(lc[:100] - la[:100]).max() <= 0.02 * la[:100].max()    # ... then the trajectories separate slowly
    assert not np.array_equal(lc, la)
    print("|delta loss| deterministic vs atomic run, mean over iterations 1-100 / 801-900: %.2e / %.2e; mean loss 801-900: %.4f vs %.4f"
          % (np.abs(lc[:100] - la[:100]).mean(), np.abs(lc[800:] - la[800:]).mean(), la[800:].mean(), lc[800:].mean()))
    # the same optimisation, not the same trajectory: ten atomic runs ended this window between 0.067 and 0.091 (one beyond 0.085:
    # the window starts at a densification) against the deterministic run's 0.0674
    assert lc[800:].mean() == pytest.approx(la[800:].mean(), rel=0.45)


def test_deterministic_captured_fits_are_bit_identical(tmp_path):
    """fit(deterministic=True): order-free sums (fixed-point integer atomics in the backward blend, ordered skinning tables) INSIDE the
    captured step -- two fits from the same seed through warm-up, densification, pruning and an opacity reset agree bit for bit in
    every loss and every parameter, at the speed of the default step (bench.py --workload trained pre-fits this way so that every
    run times the same scene).  A third fit with the float atomics starts at the same loss and ends elsewhere."""
    from dgs_amd.fit import fit
    from dgs_amd.synthetic import write_dynamic_dnerf
    dev = torch.device("cuda:0")
    data = str(tmp_path / "scene")
    write_dynamic_dnerf(data, n_train=24, n_test=2, H=128, W=128, device=dev)
    kw = dict(iterations=900, device=dev, num_pts=6000, node_num=128, seed=0, warm_up=300, regularize_from=600, densify_from=200,
              opacity_reset_interval=500)

    def run(tag, deterministic):
        tr, losses = fit(data, str(tmp_path / tag), deterministic=deterministic, **kw)
        assert tr._graph, "the step was captured"
        alive = tr.surfels.alive
        state = [t.detach()[alive].clone() for t in (tr.surfels._xyz, tr.surfels._features, tr.surfels._opacity, tr.surfels._scaling, tr.surfels._rotation)]
        state += [tr.deform.nodes.detach().clone()] + [p.detach().clone() for p in tr.deform.network.parameters()]
        tr.set_deterministic(False)
        return np.asarray(losses, dtype=np.float64), state

    la, sa = run("a", True)
    lb, sb = run("b", True)
    assert len(la) == 900 and np.isfinite(la).all()
    assert np.array_equal(la, lb), int(np.argmax(la != lb))
    assert len(sa) == len(sb) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(sa, sb))
    lc, _ = run("c", False)
    assert lc[0] == pytest.approx(la[0], rel=1e-4)
    assert not np.array_equal(lc, la)
    assert la[800:].mean() < 0.5 * la[:50].mean()                       # it learns like the default does
    assert lc[800:].mean() == pytest.approx(la[800:].mean(), rel=0.4)   # (the float-atomic run is a draw: same optimisation, other trajectory)
