"""Fit a dynamic scene end to end on the pieces of this package: D-NeRF reader -> point-cloud initialisation -> captured
train step -> in-place densification -> checkpoint files the reference can load.

The loop is the part of GUI.train_step (train_gui.py:272-432) this build covers: the joint surfel + node-deformation step
with the normal and distortion regularisers on, densification every `densify_interval` iterations between `densify_from`
and `densify_until` (size threshold 20 after the first opacity reset), opacity reset every `opacity_reset_interval`
(arguments/__init__.py:115-122), the one forced node densification / pruning at iteration 10000, the SH degree ramp
(one degree per 1000 iterations from 0).  `node_pretrain` runs the reference's node pre-training stage first
(GUI.train_node_rendering_step: dgs_amd/node_pretrain.py).  Not here: the flow losses and the GUI.  `arap=True` adds the control nodes' ARAP regulariser with the reference's weight schedule
(the step then runs eagerly until the weight reaches zero at iteration 20001, and captured from there).  Learning rates follow the reference's exponential schedules (Trainer(lr_schedule=True)).
"""
import os

import torch

from . import io as dio
from .deform import ControlNodes
from .model import SurfelModel
from .train import Trainer


def fit(data_path, model_path, iterations, device="cuda:0", white_background=False, densify_from=500, densify_interval=100,
        densify_until=50_000, opacity_reset_interval=3000, densify_grad_threshold=0.0002, slots=None, node_num=512, num_pts=100_000,
        graph=None, list_capacity=None, rasterizer_cls=None, seed=0, log=None, node_densify_at=10_000, oneup_sh_degree_step=1000,
        arap=False, warm_up=3000, regularize_from=8000, on_iteration=None, deterministic=False, views_per_rank=1, concurrent_views=False,
        node_pretrain=None, reference_update_order=False, mask_as_scene=False, mask_as_dynamic=False, random_bg_color=False,
        with_motion_mask=False):
    """Returns (trainer, losses).  slots: surfel slots to allocate (default 1.25x the initial point count; grown on demand).
    list_capacity: rasterizer list entries for the captured step (default 96 per slot).  warm_up / regularize_from: the
    reference's stages (train_gui.py:282-285: deformation detached while iteration < opt.warm_up; :292-293: normal and
    distortion regularisers off until iteration 8000).  on_iteration(it, trainer): optional hook after every iteration.
    deterministic=True (HIP path): Trainer.set_deterministic -- order-free sums instead of float atomics; two fits with the same
    arguments end bit-identical.  The trainer is returned in that mode (set_deterministic(False) restores the float atomics).
    views_per_rank / concurrent_views: k views per step and rank, added before one update -- back to back, or in flight at the same
    time, a lane each (Trainer); an iteration is then a step of k views.
    node_pretrain: None / False = the control nodes start as a farthest-point sample of the initial points (as before); True = the
    reference's default first stage (10 000 iterations: warm-up 2000, node sampling at 7500; arguments/__init__.py:128-131), or a dict
    of NodePretrainer keyword arguments (iterations, node_warm_up, sampling_at, densify_interval ...).  Data parallel: rank 0 runs the
    stage (it is a one-view-per-step loop over a few thousand small surfels), the others receive its result.
    reference_update_order: see run_iteration.
    mask_as_scene / mask_as_dynamic / random_bg_color / with_motion_mask: the reference's uses of the views' alpha channels
    (--gt_alpha_mask_as_scene_mask, --gt_alpha_mask_as_dynamic_mask, --random_bg_color, --gs_with_motion_mask; all off by default there
    too; Trainer).  A step with such a term runs eagerly: the motion-mask term ends at iteration 10001, the scene-mask compositing never."""
    device = torch.device(device)
    data = dio.load_dnerf(data_path, white_background=white_background, num_pts=num_pts, seed=seed)
    pc = data["point_cloud"]
    scene = dio.scene_from_point_cloud(pc.points, pc.colors, device=device if device.type == "cuda" else None)
    P = scene.xyz.shape[0]
    slots = int(slots or 1.25 * P)
    on_gpu = device.type == "cuda" and rasterizer_cls is None
    surfels = SurfelModel(scene, active_sh_degree=0 if oneup_sh_degree_step else 3, packed_sh=on_gpu, capacity=slots, with_motion_mask=with_motion_mask).to(device)
    torch.manual_seed(seed)
    deform = ControlNodes(node_num=min(node_num, P), K=3, hyper_dim=8, local_frame=True).to(device)
    cams = [f.camera.to(device) for f in data["train"]]
    targets = [f.image.to(device).contiguous() for f in data["train"]]
    bg = torch.tensor([1.0, 1.0, 1.0] if white_background else [0.0, 0.0, 0.0], device=device)
    extent = float(data["normalization"]["radius"])
    pre = None
    if node_pretrain:
        pre = pretrain_nodes(deform, cams, targets, bg, surfels.get_xyz.detach()[surfels.alive], extent, seed=seed, log=log,
                             white_background=white_background, densify_grad_threshold=densify_grad_threshold, rasterizer_cls=rasterizer_cls,
                             **(node_pretrain if isinstance(node_pretrain, dict) else {}))
    else:
        deform.init_from_points(surfels.get_xyz.detach()[surfels.alive], fps=True)
    masks = [f.alpha.to(device).contiguous() for f in data["train"]] if (mask_as_scene or mask_as_dynamic) else None
    tr = Trainer(surfels, deform, cams, targets, bg, rasterizer_cls=rasterizer_cls, fused_adam=None if on_gpu else False, lr_schedule=True, arap=arap,
                 views_per_rank=views_per_rank, concurrent_views=concurrent_views, alpha_masks=masks, mask_as_scene=mask_as_scene,
                 mask_as_dynamic=mask_as_dynamic, random_bg_color=random_bg_color, white_background=white_background)
    if pre is not None:
        tr.adopt_deform_state(pre.opt_deform)
    if graph is None:
        graph = on_gpu
    if deterministic:
        tr.set_deterministic(True)
    if on_gpu:
        tr.sort_surfels()
    tr.arap_from = warm_up                                             # opt.warm_up (arguments/__init__.py:102)
    tr.set_regime(warmup=1 < warm_up, lambda_normal=0.0 if 1 <= regularize_from else 0.02, lambda_dist=0.0 if 1 <= regularize_from else 1000.0)
    from .arap import LAMBDA_ARAP_STEPS
    graph_from = LAMBDA_ARAP_STEPS[-1] if (arap and graph) else 0      # the regulariser runs eagerly while its weight is non-zero
    if graph and mask_as_scene and (white_background or random_bg_color):
        graph = False                                                  # the target is re-composited in every step of the run
    elif graph and mask_as_dynamic:
        graph_from = max(graph_from, Trainer.MOTION_MASK_STEPS[-1])    # the motion-mask term's weight reaches zero there
    if graph and not graph_from:
        tr.enable_graph(int(list_capacity or 96 * slots))
    losses = []
    # A captured step returns the SAME device tensor every time (it lives in the graph's pool and is rewritten by every replay), so
    # the history comes from the step guard's pinned ring instead (Trainer.loss_history: no copy kernel per step, one
    # synchronisation per RING/2 iterations); the eager CPU path returns a fresh tensor per step.
    ring = tr.opt_deform is None and getattr(tr, "_oflag", None) is not None
    pending = 0
    sch = Schedule(warm_up=warm_up, regularize_from=regularize_from, oneup_sh_degree_step=oneup_sh_degree_step, densify_from=densify_from,
                   densify_interval=densify_interval, densify_until=densify_until, opacity_reset_interval=opacity_reset_interval,
                   densify_grad_threshold=densify_grad_threshold, node_densify_at=node_densify_at, extent=extent,
                   white_background=white_background, seed=seed, reference_update_order=reference_update_order)
    for it in range(1, iterations + 1):
        loss = run_iteration(tr, it, sch, log=log, on_gpu=on_gpu, after_step=None if on_iteration is None else (lambda: on_iteration(it, tr)))
        if ring:
            pending += 1
            if pending == tr.GUARD_RING // 2 or it == iterations:
                losses += tr.loss_history(pending)
                pending = 0
        else:
            losses.append(float(loss))
        if graph_from and it == graph_from - 1:
            tr.enable_graph(int(list_capacity or 96 * tr.P))
    save(tr, model_path, iterations)
    return tr, losses


class Schedule:
    """The joint stage's iteration schedule (the reference's defaults: arguments/__init__.py:101-122)."""

    def __init__(self, warm_up=3000, regularize_from=8000, oneup_sh_degree_step=1000, densify_from=500, densify_interval=100, densify_until=50_000,
                 opacity_reset_interval=3000, densify_grad_threshold=0.0002, node_densify_at=10_000, extent=1.0, white_background=False, seed=0,
                 reference_update_order=False):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def run_iteration(tr, it, sch, log=None, on_gpu=False, after_step=None, noise=None):
    """Iteration `it` (1-based) of the joint stage, in the order of GUI.train_step (train_gui.py:215-439): SH degree step -> regime
    (deformation detached below warm_up, regularisers behind regularize_from) -> the step (forward, backward, statistics, update)
    -> node densification -> clone / split / prune -> opacity reset.  Returns the step's loss.
    reference_update_order (off by default): the reference's density control REPLACES every surfel parameter before its optimiser steps
    (cat_tensors_to_optimizer / _prune_optimizer, scene/gaussian_model.py:327-387), the new tensors have no gradient, and
    torch.optim.Adam skips parameters without one: in an iteration that densifies, the surfels are not updated (values and moments
    stay; the deformation model is).  The trainer's step contains its update, so the surfels' state is held across such a step
    (Trainer.hold_surfels: two copies of the surfel rows per densification, every 100th iteration); likewise the node densification
    reads the surfels as they were before the iteration's update (in the default schedule it falls on a densifying iteration anyway).
    False (default): update in every iteration, density control behind it -- the order every measured number and learning test of
    this package was produced with; True reproduces the reference's trajectory (tests/test_train_step_golden.py: every loss of twelve
    iterations of its GUI.train_step, every density-control count) at the cost of those copies; on the device the flat Adam kernel
    counts steps globally, so a held parameter's bias correction runs one step ahead of torch's per-parameter count per densification
    (relative effect on the update: 3e-4 at step 600, 1e-7 at 8000).  noise: the split's standard-normal draws (tests)."""
    surfels, deform = tr.surfels, tr.deform
    if sch.oneup_sh_degree_step and it % sch.oneup_sh_degree_step == 0:        # train_gui.py:233-235
        tr.oneup_sh_degree()
    on = it > sch.regularize_from                                              # train_gui.py:292-293
    tr.set_regime(warmup=it < sch.warm_up, lambda_normal=0.02 if on else 0.0, lambda_dist=1000.0 if on else 0.0)
    densifies = it < sch.densify_until and it > sch.densify_from and it % sch.densify_interval == 0
    nodes_too = it < sch.densify_until and it == sch.node_densify_at
    held = tr.hold_surfels() if ((densifies or nodes_too) and sch.reference_update_order) else None
    held_nodes = tr.hold_surfels(nodes=True) if (nodes_too and sch.reference_update_order) else None   # replaced by the node densification
    loss = tr.step()
    updated = None
    if held_nodes is not None:
        tr.release_surfels(held_nodes)
    if held is not None:
        if not densifies:   # a node densification alone: it reads the surfels as they were BEFORE this iteration's update, which then applies
            updated = tr.hold_surfels()
        tr.release_surfels(held)
    if after_step is not None:
        tr._wait_gather()   # (data parallel, sharded SH update: the hook sees complete parameters)
        after_step()
    if it < sch.densify_until:                                                 # train_gui.py:410-423
        if it == sch.node_densify_at:       # node_force_densify_prune_step; the periodic variant is off by default in the reference
            counts = tr.densify_nodes(sch.densify_grad_threshold)
            if log and counts:
                log("[%d] nodes: added %d, pruned %d -> %d" % ((it,) + tuple(counts) + (deform.node_num,)))
            if updated is not None:
                tr.release_surfels(updated)
        if densifies:
            size_threshold = 20 if it > sch.opacity_reset_interval else None
            counts = tr.densify_and_prune(sch.densify_grad_threshold, 0.01, sch.extent, size_threshold, seed=sch.seed, noise=noise)
            if on_gpu:
                tr.sort_surfels()   # children landed in free slots anywhere: restore the node order (in place, no re-capture)
                if tr.refresh_knn_mode() and log:   # the hyper coordinates train: the neighbour search may need its other kernel
                    log("[%d] neighbour search: %s (spatial share of the K-th distance %.2f)" % (it, deform.knn_refine_mode, deform.knn_spatial_share))
            if log:
                log("[%d] cloned %d, split %d, pruned %d -> %d surfels (%d slots)" % ((it,) + tuple(counts) + (surfels.num_surfels, tr.P)))
        if it % sch.opacity_reset_interval == 0 or (sch.white_background and it == sch.densify_from):
            tr.reset_opacity()
    return loss


def pretrain_nodes(deform, cams, targets, bg, points, extent, seed=0, log=None, **kw):
    """The node pre-training stage in front of the joint stage (train_gui.py:207-213: node rendering steps until
    iterations_node_rendering, then train_step).  Returns the NodePretrainer (its `opt_deform` carries on: Trainer.adopt_deform_state).
    With more than one rank, rank 0 runs the stage and broadcasts the deformation parameters and their Adam state: the stage's float
    atomics are not bit-reproducible across ranks, and replicas must start identical."""
    import torch.distributed as dist
    from .node_pretrain import Draws, NodePretrainer
    pre = NodePretrainer(deform, cams, targets, bg, points, extent, draws=Draws(seed), log=log, **kw)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1 or dist.get_rank() == 0:
        pre.run()
    if world > 1:
        for grp in pre.opt_deform.param_groups:
            for p in grp["params"]:
                dist.broadcast(p.data, 0)
                st = pre.opt_deform.state.get(p)
                has = torch.tensor([1.0 if st else 0.0], device=p.device)
                dist.broadcast(has, 0)
                if not bool(has.item()):
                    continue
                if not st:
                    st = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
                    pre.opt_deform.state[p] = st
                step = st["step"].to(p.device).reshape(1).float()
                dist.broadcast(step, 0)
                st["step"] = step.reshape(()).cpu()
                dist.broadcast(st["exp_avg"], 0)
                dist.broadcast(st["exp_avg_sq"], 0)
    return pre


def save(trainer, model_path, iteration):
    """Scene.save + DeformModel.save_weights (scene/__init__.py, scene/deform_model.py:41-44): the two files of a checkpoint."""
    if hasattr(trainer, "_flush_guard"):
        for _ in range(8):   # steps the guard skipped in the last iterations are redone before the state is written
            if not trainer._flush_guard():
                break
            while trainer.iteration < iteration:   # the recovery rewound the iteration counter by the number of skipped steps
                trainer.step()
    dio.save_surfels(trainer.surfels, os.path.join(model_path, "point_cloud/iteration_{}".format(iteration), "point_cloud.ply"))
    dio.save_deform(trainer.deform, model_path, iteration)


def restore(model_path, iteration=-1, device="cpu", packed_sh=False, slots=None, node_num=512):
    """Surfels + deformation of a checkpoint directory written by `save` (or by the reference)."""
    it = dio.search_for_max_iteration(os.path.join(model_path, "point_cloud")) if iteration == -1 else iteration
    scene = dio.load_surfels(os.path.join(model_path, "point_cloud/iteration_{}".format(it), "point_cloud.ply"))
    surfels = SurfelModel(scene, packed_sh=packed_sh, capacity=slots, with_motion_mask=scene.feature.shape[1] == 9).to(device)
    deform = ControlNodes(node_num=node_num, K=3, hyper_dim=8, local_frame=True).to(device)
    if not dio.load_deform(deform, model_path, iteration, pad_to=64 if (packed_sh and torch.device(device).type == "cuda") else 1):
        raise FileNotFoundError("no deform.pth under %s" % model_path)
    return surfels, deform.to(device)
