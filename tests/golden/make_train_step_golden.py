"""Generates tests/golden/train_step_golden.npz by running the REFERENCE's joint-stage iteration -- the body of GUI.train_step
(train_gui.py:215-439), compiled from the reference's file in the build container -- on the CPU for 12 consecutive iterations that
cross everything the joint stage does: the end of the deformation warm-up (opt.warm_up), the regularisers switching on behind the
hard-wired iteration 8000 (train_gui.py:292-293), an SH degree step, densification statistics, the forced node densification /
pruning, clone / split / prune of the surfels, an opacity reset, both optimisers with their learning-rate schedules, the ARAP
regulariser of the control nodes with its landmark weight.

The reference's own code here: train_step, GaussianModel (density control, optimiser surgery, schedules), DeformModel /
ControlNodeWarp / DeformNetwork (skinning, node densification, ARAP), render(), l1_loss / ssim, landmark_interpolate.  Stand-ins as in
make_node_pretrain_golden.py (knn_points, distCUDA2, the oracle operator as rasterizer, `.cuda()` -> CPU) plus inert objects for what
a training step touches but this comparison does not read: network_gui (no connection), the CUDA timing events, the progress bar,
training_report (metrics on test views), Scene.save.  Random draws (view choice, ARAP times, split noise) are recorded per iteration.
Run from the repo root:  python tests/golden/make_train_step_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_deform_golden import fill_params  # noqa: E402
from make_densify_golden import cuda_to_cpu  # noqa: E402
from make_node_pretrain_golden import Recorder, import_reference_stack, reference_method  # noqa: E402

CASE = dict(P=260, nodes=24, views=6, S=44, seed=4, width=32, first=7995, last=8006, warm_up=7997, oneup=7996, densify_from=7990,
            densify_interval=5, opacity_reset_interval=8000, node_force=7999, densify_grad_threshold=0.003, extent=18.0)
# iterations 7995 .. 8006: deformation detached for 7995-7996 (< warm_up), ARAP joins the loss at 7998 (> warm_up), SH degree 0 -> 1 at 7996,
# normal / distortion regularisers from 8001, node densification at 7999, surfel densification at 7995, 8000, 8005, opacity reset at 8000 (so that the call at 8005 prunes)


def scene_inputs():
    """Cameras, targets, the initial point cloud with colours -- formulas shared with the test."""
    from make_node_pretrain_golden import CASES, scene_inputs as blobs
    CASES["_train_step"] = dict(views=CASE["views"], S=CASE["S"], seed=CASE["seed"], n_points=CASE["P"])
    cams, targets, pts = blobs("_train_step")
    del CASES["_train_step"]
    g = torch.Generator().manual_seed(CASE["seed"] + 1)
    cols = torch.rand(CASE["P"], 3, generator=g)
    return cams, targets, pts.numpy().astype(np.float64), cols.numpy().astype(np.float64)


def main(variant="default"):
    c = CASE
    masks = variant == "masks"
    tu, gm, dm, ref_renderer, lu = import_reference_stack()
    cams, targets, pts, cols = scene_inputs()
    views = [SimpleNamespace(**cam._asdict(), original_image=targets[k], gt_alpha_mask=alpha_masks(targets)[k] if masks else None, image_name="v%d" % k,
                             flow_dirs=[], load2device=lambda *a: None) for k, cam in enumerate(cams)]
    opt = SimpleNamespace(
        iterations=80_000, warm_up=c["warm_up"], dynamic_color_warm_up=20_000, oneupSHdegree_step=c["oneup"], progressive_train=False,
        progressive_stage_steps=3000, progressive_stage_ratio=0.2, random_bg_color=masks, gt_alpha_mask_as_scene_mask=masks,
        gt_alpha_mask_as_dynamic_mask=masks, lambda_dssim=0.2, lambda_optical_landmarks=[1e-1, 1e-1, 1e-3, 0],
        lambda_optical_steps=[0, 15_000, 25_000, 25_001], lambda_motion_mask_landmarks=[5e-1, 1e-2, 0], lambda_motion_mask_steps=[0, 10_000, 10_001],
        no_motion_mask_loss=False, densify_until_iter=50_000, densify_from_iter=c["densify_from"], densification_interval=c["densify_interval"],
        opacity_reset_interval=c["opacity_reset_interval"], densify_grad_threshold=c["densify_grad_threshold"],
        node_densify_from_iter=1000, node_densification_interval=5000, node_densify_until_iter=25_000, node_force_densify_prune_step=c["node_force"],
        node_enable_densify_prune=False, percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
        position_lr_max_steps=30_000, deform_lr_max_steps=40_000, feature_lr=0.004, opacity_lr=0.05, scaling_lr=0.002, rotation_lr=0.002,
        deform_lr_scale=1.0, no_arap_loss=False)
    rec = Recorder(c["seed"] + 200)
    saved = (torch.rand, torch.normal, torch.randint, torch.Tensor.to)
    rand_like = torch.rand_like
    torch.rand, torch.normal, torch.randint = rec.rand, rec.normal, rec.randint
    torch.rand_like = lambda t, **k: rec.rand(*t.shape)        # the random background of render() (gaussian_renderer/__init__.py:58)
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else saved[3](self, *a, **k)
    losses, marks = [], []
    backward = torch.Tensor.backward

    def recording_backward(self, *a, **k):
        losses.append(float(self.detach()))
        return backward(self, *a, **k)
    out = {}
    try:
        with cuda_to_cpu():
            torch.manual_seed(0)
            init = tu.DeformNetwork.__init__
            defaults = init.__defaults__
            init.__defaults__ = (8, c["width"]) + defaults[2:]        # see make_node_pretrain_golden.py: a narrower reference network
            deform = dm.DeformModel(K=3, deform_type="node", is_blender=True, skinning=False, hyper_dim=8, node_num=c["nodes"], pred_opacity=False,
                                    pred_color=False, use_hash=False, hash_time=False, d_rot_as_res=True, local_frame=True,
                                    progressive_brand_time=False, with_arap_loss=True, max_d_scale=-1, enable_densify_prune=False,
                                    is_scene_static=False)
            init.__defaults__ = defaults
            fill_params(deform.deform.network)
            with torch.no_grad():
                deform.deform.network.gaussian_warp.weight.mul_(4.0)
                deform.deform.network.gaussian_rotation.weight.mul_(2.0)
            deform.train_setting(opt)
            gaussians = gm.GaussianModel(3, fea_dim=8, with_motion_mask=masks)
            pcd = gm.BasicPointCloud(points=pts, colors=cols, normals=np.zeros_like(pts))
            gaussians.create_from_pcd(pcd, print_info=False)
            with torch.no_grad():                  # surfels of a few pixels, some opaque enough to matter, non-trivial higher-order SH
                gaussians._scaling += torch.tensor([0.2, 0.35])      # anisotropic: an isotropic surfel's in-plane rotation is a gauge direction
                gaussians._opacity += 2.0
                g = torch.Generator().manual_seed(c["seed"] + 2)
                gaussians._features_rest += 0.05 * torch.randn(gaussians._features_rest.shape, generator=g)
                gaussians._rotation += 0.3 * torch.randn(gaussians._rotation.shape, generator=g)
            gaussians.training_setup(opt)
            deform.deform.init(init_pcl=gaussians.get_xyz, force_init=True, opt=opt, as_gs_force_with_motion_mask=False, force_gs_keep_all=False)
            deform.deform.train()
            # a state like that of a run in progress: at initialisation every node has the same hyper coordinates and radius, so the
            # gradient of the surfels' hyper coordinates cancels exactly in exact arithmetic -- rounding noise that Adam amplifies
            with torch.no_grad():
                g = torch.Generator().manual_seed(c["seed"] + 3)
                deform.deform.nodes.data[:, 3:] += 0.02 * saved[0](c["nodes"], 8, generator=g)   # (torch.rand itself is the recorder here)
                deform.deform._node_radius.data += 0.1 * torch.randn(c["nodes"], generator=g)
                deform.deform._node_weight.data += 0.3 * torch.randn(c["nodes"], 1, generator=g)
                gaussians.feature.data[:, :8] += 0.01 * torch.randn(c["P"], 8, generator=g)
                if masks:
                    gaussians.feature.data[:, 8] += 0.5 * torch.randn(c["P"], generator=g)
            out["nodes0"] = deform.deform.nodes.detach().numpy().copy()
            # the state a run that reached iteration `first` would be in: both schedules evaluated by the iteration before, and Adam
            # step counts of that size (with zero moments: no history is invented) -- at step counts of 1, 2, 3 the bias corrections
            # differ by tens of percent from one step to the next, which turns the one-step lag between a per-parameter count (torch)
            # and a global one (the flat Adam kernel of the device path) after a held update into a visible difference; at the
            # counts of a run in progress it is 1e-7
            gaussians.update_learning_rate(c["first"] - 1)
            deform.update_learning_rate(c["first"] - 1)
            for opt_ in (gaussians.optimizer,):     # (the deformation's parameters start counting when the warm-up ends, at 7997: in both)
                for grp_ in opt_.param_groups:
                    for p_ in grp_["params"]:
                        opt_.state[p_] = {"step": torch.tensor(float(c["first"] - 1)), "exp_avg": torch.zeros_like(p_.data), "exp_avg_sq": torch.zeros_like(p_.data)}
            bar = SimpleNamespace(set_postfix=lambda *a, **k: None, update=lambda *a, **k: None, close=lambda: None, set_description=lambda *a, **k: None)
            zero = torch.zeros(())
            gui = SimpleNamespace(
                viewpoint_stack=None, opt=opt, iteration=c["first"], deform=deform, gaussians=gaussians, background=torch.zeros(3), args=SimpleNamespace(model_path="/nonexistent"),
                pipe=SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=1.0),
                dataset=SimpleNamespace(load2gpu_on_the_fly=False, is_blender=True, white_background=False, source_path=""),
                scene=SimpleNamespace(getTrainCameras=lambda: list(views), cameras_extent=c["extent"], save=lambda it: None),
                iter_start=SimpleNamespace(record=lambda: None, elapsed_time=lambda e: 0.0), iter_end=SimpleNamespace(record=lambda: None),
                tb_writer=None, testing_iterations=[], saving_iterations=[], progress_bar=bar, ema_loss_for_log=0.0, best_psnr=0.0, best_iteration=0,
                best_ssim=0.0, best_ms_ssim=0.0, best_lpips=1e10, best_alex_lpips=1e10, smooth_term=None)
            ns = dict(randint=rec.randint_py, torch=torch, np=np, render=ref_renderer.render, l1_loss=lu.l1_loss, ssim=lu.ssim, os=os,
                      network_gui=SimpleNamespace(conn=None, try_connect=lambda: None), landmark_interpolate=tu.landmark_interpolate,
                      training_report=lambda *a, **k: (zero, zero, zero, zero, zero), render_flow=None, imageio=None)
            exec(reference_method("train_step"), ns)
            step = ns["train_step"]
            # density-control bookkeeping (as in make_node_pretrain_golden.py) + the node densification's counts
            calls, cur = [], {}
            post, prune, dap = gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune

            def postfix(self, new_xyz, *a, **k):
                cur.setdefault("added", []).append(int(new_xyz.shape[0]))
                return post(self, new_xyz, *a, **k)

            def prune_points(self, mask):
                cur.setdefault("pruned", []).append(int(mask.sum()))
                return prune(self, mask)

            def densify_and_prune(self, *a, **k):
                cur.clear()
                r = dap(self, *a, **k)
                calls.append((gui.iteration, cur["added"][0], cur["pruned"][0], cur["pruned"][1], int(self.get_xyz.shape[0])))
                return r
            das, parents = gm.GaussianModel.densify_and_split, {}

            def densify_and_split(self, grads=None, grad_threshold=None, scene_extent=None, N=2, **k):
                if self is not gaussians or grads is None:      # (the node densification splits the nodes' own surfels, N = 1)
                    return das(self, grads, grad_threshold, scene_extent, N, **k)
                # the same selection the method makes (scene/gaussian_model.py:418-425), to record WHICH surfels the noise rows belong to
                padded = torch.zeros(self.get_xyz.shape[0])
                padded[:grads.shape[0]] = grads.squeeze()
                sel = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
                parents[gui.iteration] = self.get_xyz.detach()[sel].numpy().copy()
                return das(self, grads, grad_threshold, scene_extent, N, **k)
            gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune = postfix, prune_points, densify_and_prune
            gm.GaussianModel.densify_and_split = densify_and_split
            torch.Tensor.backward = recording_backward
            per_it = []
            while gui.iteration <= c["last"]:
                it = gui.iteration
                marks.append((it, len(rec.log)))
                step(gui)
                per_it.append((it, gaussians.get_xyz.shape[0], deform.deform.nodes.shape[0], gaussians.active_sh_degree,
                               float(gaussians.get_xyz.detach().abs().sum()), float(gaussians.get_opacity.detach().sum()),
                               float(deform.deform.nodes.detach().abs().sum()), float(gaussians.max_radii2D.sum())))
            torch.Tensor.backward = backward
            gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune = post, prune, dap
            gm.GaussianModel.densify_and_split = das
            for it_, v_ in parents.items():
                out["split_parents_%d" % it_] = v_
            srt = lambda t: t.detach().reshape(t.shape[0], -1).numpy()
            order = np.lexsort(srt(gaussians._xyz).T[::-1])
            out.update(
                calls=np.array(calls), per_it=np.array(per_it, dtype=np.float64), losses=np.array(losses), marks=np.array(marks),
                final_xyz=srt(gaussians._xyz)[order], final_opacity=srt(gaussians._opacity)[order], final_scaling=srt(gaussians._scaling)[order],
                final_f_dc=srt(gaussians._features_dc)[order], final_feature=srt(gaussians.feature)[order],
                final_nodes=deform.deform.nodes.detach().numpy().copy(), final_warp_w=deform.deform.network.gaussian_warp.weight.detach().numpy().copy(),
                final_lr_xyz=np.array([g_["lr"] for g_ in gaussians.optimizer.param_groups if g_["name"] == "xyz"]),
                final_lr_deform=np.array([g_["lr"] for g_ in deform.optimizer.param_groups]))
    finally:
        torch.rand, torch.normal, torch.randint, torch.Tensor.to = saved
        torch.rand_like = rand_like
        torch.Tensor.backward = backward
    out["draw_kinds"] = np.array([k for k, _ in rec.log])
    for i, (_, v) in enumerate(rec.log):
        out["draw_%03d" % i] = np.asarray(v)
    print("losses", ["%.5f" % v for v in out["losses"]])
    print("per iteration (it, surfels, nodes, sh degree):", out["per_it"][:, :4].astype(int).tolist())
    print("density control (iteration, cloned, split, pruned, rows):", out["calls"].tolist())
    kinds = [k for k, _ in rec.log]
    print("draws:", {k: kinds.count(k) for k in sorted(set(kinds))})
    np.savez_compressed(os.path.join(HERE, "train_step_golden.npz" if variant == "default" else "train_step_golden_%s.npz" % variant), **out)


from make_node_pretrain_golden import alpha_masks  # noqa: E402,F401  (ground-truth masks of the views, shared with the node stage's golden)


if __name__ == "__main__":
    main()
    main("masks")
