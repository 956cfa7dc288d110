"""Fit a dynamic scene end to end on the pieces of this package: D-NeRF reader -> point-cloud initialisation -> captured
train step -> in-place densification -> checkpoint files the reference can load.

The loop is the part of GUI.train_step (train_gui.py:272-432) this build covers: the joint surfel + node-deformation step
with the normal and distortion regularisers on, densification every `densify_interval` iterations between `densify_from`
and `densify_until` (size threshold 20 after the first opacity reset), opacity reset every `opacity_reset_interval`
(arguments/__init__.py:115-122), the one forced node densification / pruning at iteration 10000, the SH degree ramp
(one degree per 1000 iterations from 0).  Not here: the node warm-up
stage, the flow losses and the GUI.  `arap=True` adds the control nodes' ARAP regulariser with the reference's weight schedule
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
        arap=False, warm_up=3000, regularize_from=8000, on_iteration=None, deterministic=False, views_per_rank=1, concurrent_views=False):
    """Returns (trainer, losses).  slots: surfel slots to allocate (default 1.25x the initial point count; grown on demand).
    list_capacity: rasterizer list entries for the captured step (default 96 per slot).  warm_up / regularize_from: the
    reference's stages (train_gui.py:282-285: deformation detached while iteration < opt.warm_up; :292-293: normal and
    distortion regularisers off until iteration 8000).  on_iteration(it, trainer): optional hook after every iteration.
    deterministic=True (HIP path): Trainer.set_deterministic -- order-free sums instead of float atomics; two fits with the same
    arguments end bit-identical.  The trainer is returned in that mode (set_deterministic(False) restores the float atomics).
    views_per_rank / concurrent_views: k views per step and rank, added before one update -- back to back, or in flight at the same
    time, a lane each (Trainer); an iteration is then a step of k views."""
    device = torch.device(device)
    data = dio.load_dnerf(data_path, white_background=white_background, num_pts=num_pts, seed=seed)
    pc = data["point_cloud"]
    scene = dio.scene_from_point_cloud(pc.points, pc.colors, device=device if device.type == "cuda" else None)
    P = scene.xyz.shape[0]
    slots = int(slots or 1.25 * P)
    on_gpu = device.type == "cuda" and rasterizer_cls is None
    surfels = SurfelModel(scene, active_sh_degree=0 if oneup_sh_degree_step else 3, packed_sh=on_gpu, capacity=slots).to(device)
    torch.manual_seed(seed)
    deform = ControlNodes(node_num=min(node_num, P), K=3, hyper_dim=8, local_frame=True).to(device)
    deform.init_from_points(surfels.get_xyz.detach()[surfels.alive], fps=True)
    cams = [f.camera.to(device) for f in data["train"]]
    targets = [f.image.to(device).contiguous() for f in data["train"]]
    bg = torch.tensor([1.0, 1.0, 1.0] if white_background else [0.0, 0.0, 0.0], device=device)
    tr = Trainer(surfels, deform, cams, targets, bg, rasterizer_cls=rasterizer_cls, fused_adam=None if on_gpu else False, lr_schedule=True, arap=arap,
                 views_per_rank=views_per_rank, concurrent_views=concurrent_views)
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
    if graph and not graph_from:
        tr.enable_graph(int(list_capacity or 96 * slots))
    extent = float(data["normalization"]["radius"])
    losses = []
    # A captured step returns the SAME device tensor every time (it lives in the graph's pool and is rewritten by every replay), so
    # the history comes from the step guard's pinned ring instead (Trainer.loss_history: no copy kernel per step, one
    # synchronisation per RING/2 iterations); the eager CPU path returns a fresh tensor per step.
    ring = tr.opt_deform is None and getattr(tr, "_oflag", None) is not None
    pending = 0
    for it in range(1, iterations + 1):
        if oneup_sh_degree_step and it % oneup_sh_degree_step == 0:                # train_gui.py:233-235
            tr.oneup_sh_degree()
        on = it > regularize_from                                                  # train_gui.py:292-293
        tr.set_regime(warmup=it < warm_up, lambda_normal=0.02 if on else 0.0, lambda_dist=1000.0 if on else 0.0)
        loss = tr.step()
        if ring:
            pending += 1
            if pending == tr.GUARD_RING // 2 or it == iterations:
                losses += tr.loss_history(pending)
                pending = 0
        else:
            losses.append(float(loss))
        if on_iteration is not None:
            tr._wait_gather()   # (data parallel, sharded SH update: the hook sees complete parameters)
            on_iteration(it, tr)
        if graph_from and it == graph_from - 1:
            tr.enable_graph(int(list_capacity or 96 * tr.P))
        if it < densify_until:                                                     # train_gui.py:410-423
            if it == node_densify_at:       # node_force_densify_prune_step; the periodic variant is off by default in the reference
                counts = tr.densify_nodes(densify_grad_threshold)
                if log and counts:
                    log("[%d] nodes: added %d, pruned %d -> %d" % ((it,) + tuple(counts) + (deform.node_num,)))
            if it > densify_from and it % densify_interval == 0:
                size_threshold = 20 if it > opacity_reset_interval else None
                counts = tr.densify_and_prune(densify_grad_threshold, 0.01, extent, size_threshold, seed=seed)
                if on_gpu:
                    tr.sort_surfels()   # children landed in free slots anywhere: restore the node order (in place, no re-capture)
                    if tr.refresh_knn_mode() and log:   # the hyper coordinates train: the neighbour search may need its other kernel
                        log("[%d] neighbour search: %s (spatial share of the K-th distance %.2f)" % (it, deform.knn_refine_mode, deform.knn_spatial_share))
                if log:
                    log("[%d] cloned %d, split %d, pruned %d -> %d surfels (%d slots)" % ((it,) + tuple(counts) + (surfels.num_surfels, tr.P)))
            if it % opacity_reset_interval == 0 or (white_background and it == densify_from):
                tr.reset_opacity()
    save(tr, model_path, iterations)
    return tr, losses


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
    surfels = SurfelModel(scene, packed_sh=packed_sh, capacity=slots).to(device)
    deform = ControlNodes(node_num=node_num, K=3, hyper_dim=8, local_frame=True).to(device)
    if not dio.load_deform(deform, model_path, iteration, pad_to=64 if (packed_sh and torch.device(device).type == "cuda") else 1):
        raise FileNotFoundError("no deform.pth under %s" % model_path)
    return surfels, deform.to(device)
