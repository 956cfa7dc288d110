#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): train views/sec (fwd+bwd), 800x800, 200k surfels, at 1/2/4/8 GPUs.
One "step" = one full training pass of the hot path over one synthetic view per rank:
  node deformation (PyTorch-ROCm) -> HIP surfel rasterizer forward -> L1 + D-SSIM + normal + distortion loss
  -> rasterizer backward -> deformation backward -> [N>1] one flat RCCL all-reduce -> Adam (surfels + deform).
Inputs (scene S(200k,800,800,seed 0)) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline"        : achieved algorithmic HBM GB/s of the dominant kernel (the backward blend) against the 8 TB/s HBM3E peak.
                      Timed AFTER the headline region, on the scene state the region ended with (snapshot / restore), with
                      device timestamps inside the replayed whole-step graph -- the launch mode of the headline -- and, as a
                      cross-check, with HIP events around eager launches;
  "roofline_fwd"    : the same for the forward blend;
  "roofline_valu"   : the two blend kernels against what actually bounds them -- VALU issue: wave instructions per launch (PMC)
                      / duration against 0.5 instructions per cycle per SIMD (MI355X_MICROARCH.md: 2 cycles per wave64 op);
  "roofline_kernels": preprocess_fwd, binning (count + scan + scatter + per-tile sort) and surfel_bwd with the bytes SURVEY.md
                      section 8(d) defines for them (B2, B4-6, B10), same clocks;
  "twin"            : the graph-replayed steps and the eager steps of the roofline legs start from the SAME snapshot and render
                      the same views: their per-step losses must agree (1e-5 relative over the first 3 steps; over all steps 1e-3 or 4x the
                      difference of two eager runs, whichever is larger), or the result is invalid;
  "drift"           : the scene trains on noise targets while it is timed and its splats grow: ms/step of a second window 100
                      steps after the headline window, and the mean screen radius at both (the headline is the FIRST window);
  "cpu_baseline"    : the CPU port (C oracle rasterizer with OpenMP + PyTorch-CPU deformation/loss/Adam) timed on
                      this box's host cores for a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

# ROCm 7.2 replays captured graphs through pre-recorded AQL packets by default; with this train step (memset nodes of
# PyTorch's multi-block reductions and of the rasterizer between kernel nodes) that path intermittently executes a
# zero-fill out of order (an L1 term of exactly 0, or 1e18 gradients, depending on host timing).  The runtime knob
# below selects the regular graph launch path; it must be in the environment before the HIP runtime initialises.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
# multi-process GPU work on this pool needs dmabuf IPC (RCCL, CUDA-tensor sharing): also read when the HIP runtime starts, so it
# is set here, before torch is imported, and not next to init_process_group
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (surfels, H, W)   -- "metric" is the configuration BASELINE.json's metric is quoted on
    "metric": (200_000, 800, 800),
    "c3": (150_000, 800, 800),
    "c4": (300_000, 800, 800),
    "c5": (1_000_000, 1600, 1600),
    "c2": (50_000, 800, 800),          # BASELINE.json configs[1]: static canonical render, FORWARD ONLY, no deformation
    "tiny": (5_000, 128, 128),
    # a scene WITH structure: dgs_amd.synthetic.DynamicTruth rendered at 800x800, fitted from 100k random points with densification
    # and the reference's stages (see trained_trainer) before the timed region -- ~100k surfels crowd onto two surfaces that cover a
    # third of the image, so the tile lists have the statistics of a trained scene (lists of 1000-2100 entries in the covered
    # tiles, none elsewhere) instead of the uniform cloud of "metric"
    "trained": (100_000, 800, 800),
}


def build_trainer(P, H, W, device, n_views=64, n_targets=8, rasterizer_cls=None, fused_adam=None, packed_sh=None, slots=None,
                  sort_surfels=None, views_per_rank=1, concurrent_views=False):
    from dgs_amd.cameras import orbit_cameras
    from dgs_amd.deform import ControlNodes
    from dgs_amd.model import SurfelModel
    from dgs_amd.synthetic import make_scene, target_image
    from dgs_amd.train import Trainer
    torch.manual_seed(0)
    scene = make_scene(P, seed=0)
    if packed_sh is None:  # HIP product path: SH coefficients as one parameter (no per-render concatenation)
        packed_sh = torch.device(device).type == "cuda" and rasterizer_cls is None and fused_adam is not False
    surfels = SurfelModel(scene, packed_sh=packed_sh, capacity=slots).to(device)   # slots > P: room for densification
    deform = ControlNodes(node_num=1024, K=3, hyper_dim=8, local_frame=True).to(device)
    deform.init_from_points(surfels.get_xyz.detach()[surfels.alive], fps=True)
    cams = [c.to(device) for c in orbit_cameras(n_views, W, H)]
    targets = [target_image(H, W, seed=1 + v).to(device) for v in range(n_targets)]
    bg = torch.zeros(3, device=device)
    tr = Trainer(surfels, deform, cams, targets, bg, rasterizer_cls=rasterizer_cls, fused_adam=fused_adam, views_per_rank=views_per_rank,
                 concurrent_views=concurrent_views)
    if sort_surfels is None:
        sort_surfels = torch.device(device).type == "cuda" and rasterizer_cls is None and os.environ.get("DGS_SORT_SURFELS", "1") != "0"
    if sort_surfels:
        tr.sort_surfels()   # storage order = nearest control node: part of the initialisation, like the reference's own
                            # Morton ordering inside simple-knn; the per-step kernels of the deformation rely on it for speed
    return tr


def blend_bytes(S_per_launch, ntiles, HW, backward):
    """Algorithmic bytes per launch of the blend kernels (SURVEY.md section 8d / DESIGN.md):
    fwd B7 = 76*S + 8*T + 64*H*W ; bwd B8 = B7 + 4*16*N_vis with N_vis bounded by S."""
    b = 76.0 * S_per_launch + 8.0 * ntiles + 64.0 * HW
    if backward:
        b += 64.0 * S_per_launch
    return b


def measured_hbm_ceiling(device, mib=1024, iters=20):
    """What this box's HBM actually delivers to a trivially streaming kernel (SURVEY.md section 8d asks for it next to the vendor
    peak): device-to-device copy (read + write) and the triad a = b + s*c (two reads + one write) over `mib` MiB buffers, far
    beyond the 256 MB Infinity Cache."""
    n = mib * (1 << 20) // 4
    a, b, c = (torch.empty(n, dtype=torch.float32, device=device).fill_(v) for v in (0.0, 1.0, 2.0))
    out = {}
    for name, fn, nbuf in (("copy", lambda: a.copy_(b), 2), ("triad", lambda: torch.add(b, c, alpha=0.5, out=a), 3)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(nbuf * n * 4 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b, c
    torch.cuda.empty_cache()
    return out


def cpu_baseline(P, H, W, budget_s=25.0):
    """The oracle-backed CPU port of the same train step, bounded sample: as many steps as fit 10-25 s of CPU work (at ~1.1 s per step
    on 32 threads: 10-20 steps; round 3 stopped after 3 steps / 3.3 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_raster_op import OracleRasterizer  # test-only operator: oracle/ is the checker, never shipped
    from oracle import surfel_oracle
    ncores = min(os.cpu_count() or 1, 32)  # THREADS used (more only add contention on the accumulations); `cores` reports what the box has
    torch.set_num_threads(ncores)
    surfel_oracle.set_threads(ncores)
    tr = build_trainer(P, H, W, torch.device("cpu"), n_views=8, n_targets=2, rasterizer_cls=OracleRasterizer, fused_adam=False)
    tr.step()  # untimed: the first call pays one-off BLAS/oneDNN initialisation (tens of seconds on some hosts)
    t0 = time.time()
    tr.step()
    n = 1
    per = time.time() - t0
    while n < 20 and ((time.time() - t0) + per < budget_s) and (n < 3 or (time.time() - t0) < 10.0):
        tr.step()
        n += 1
        per = (time.time() - t0) / n
    dt = time.time() - t0
    # `cores` is what the bench contract defines it as -- the threads actually used; `host_cores` is what the box has (the north star
    # asks for the core count to be stated: both are, under names that cannot be confused)
    return {"value": n / dt, "unit": "views/s", "cores": ncores, "threads": ncores, "host_cores": os.cpu_count() or 1, "kind": "port",
            "sample": "%d full train steps (views) of the same %dk-surfel %dx%d workload, %.1f s" % (n, P // 1000, W, H, dt)}


def concurrent_views_report(args, P, H, W, device, ms_headline):
    """Single GPU, after the headline: the same workload with TWO views per step IN FLIGHT AT THE SAME TIME (Trainer(views_per_rank=2,
    concurrent_views=True): a lane per view -- own stream, captured graph, gradient bucket, rasterizer context -- and one update that
    reads the sum of the two buckets), W warm-up + K timed steps from a fresh scene; next to it the same two views back to back
    (the sequential views_per_rank=2 of round 5).  The headline stays one view per step; this object says what the idle half of that
    step is worth when another view fills it."""
    out = {"k": 2}
    for name, conc, k in (("concurrent", True, 2), ("sequential", False, 2), ("concurrent_k3", True, 3)):
        tr = build_trainer(P, H, W, device, views_per_rank=k, concurrent_views=conc)
        tr.enable_graph(capacity=24 * P)
        for _ in range(args.warmup):
            tr.step()
        dt = timed_steps(tr, args.steps, 1)
        clean = not bool(tr._oflag.item()) and tr.overflow_recoveries == 0
        ms = dt / args.steps * 1e3
        out[name] = {"k": k, "ms_per_step": round(ms, 4), "ms_per_view": round(ms / k, 4), "views_per_s": round(k * args.steps / dt, 2), "clean": clean}
        del tr
        torch.cuda.empty_cache()
    out["ms_per_view"] = out["concurrent"]["ms_per_view"]
    out["views_per_s"] = out["concurrent"]["views_per_s"]
    out["vs_one_view_per_step"] = round(ms_headline / out["concurrent"]["ms_per_view"], 3)
    out["what_limits_it"] = ("the two blend kernels saturate every CU (5 workgroups of 96 VGPRs / 30 KB LDS each); a kernel of the other lane whose "
                             "workgroups are large (1024 threads, tens of KB of LDS: scatter, offsets, node MLP, weight gradients) is placed only "
                             "when the blend's grid drains: profiles/r06_concurrent_timeline.txt")
    return out


def trained_trainer(P, H, W, device, pre_iterations):
    """--workload trained: fit() on the synthetic D-NeRF-format dataset (dgs_amd.synthetic.DynamicTruth with 40k finely textured
    surfels, which the fit needs ~100k surfels to reproduce) for `pre_iterations` with the reference's stages: deformation
    detached for the first 3000 iterations, regularisers from 8000, densification every 100 iterations from 500, opacity resets
    every 3000 (train_gui.py:272-313, 410-423; shorter runs compress the stages to 0.3 / 0.8 of the run).  The caller then times the
    full late-regime step on what that produced -- against the dataset's real target views."""
    import shutil
    import tempfile
    from dgs_amd.fit import fit
    from dgs_amd.synthetic import DynamicTruth, write_dynamic_dnerf
    tmp = tempfile.mkdtemp(prefix="dgs_trained_")
    try:
        write_dynamic_dnerf(os.path.join(tmp, "scene"), n_train=48, n_test=2, H=H, W=W, device=device, truth=DynamicTruth(24000, 16000, detail=0.3))
        warm_up, reg_from = (3000, 8000) if pre_iterations >= 10000 else (3 * pre_iterations // 10, 8 * pre_iterations // 10)
        # deterministic pre-fit (order-free sums instead of float atomics, Trainer.set_deterministic): every run of this workload times the
        # SAME scene -- with the float atomics the fit's outcome, and with it the timed step, varied by +-6 % from run to run (round 4)
        det = os.environ.get("DGS_TRAINED_DETERMINISTIC", "1") != "0"
        tr, losses = fit(os.path.join(tmp, "scene"), os.path.join(tmp, "model"), iterations=pre_iterations, device=device, num_pts=P, node_num=512,
                         seed=0, warm_up=warm_up, regularize_from=reg_from, node_densify_at=10 ** 9, deterministic=det, reference_update_order=False)
        if det:
            tr.set_deterministic(False)   # the timed steps are the product's default kernels
        tr.pre_fit_mode = "deterministic" if det else "float atomics"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return tr, losses


def kernel_bytes(kind, P, Pv, R, M, ntiles):
    """Algorithmic bytes per launch as SURVEY.md section 8(d) defines them for the reference's algorithm (P surfels, Pv visible,
    R = num_rendered, M SH coefficients per channel), and the bytes THIS implementation's layout moves for the same work."""
    if kind == "preprocess_fwd":   # B2
        return {"survey": (12 + 8 + 16 + 4 + 12 * M) * P + 91 * Pv, "own": (12 + 8 + 16 + 4 + 12 * M) * P + 104 * Pv,
                "formula": "B2 = (12+8+16+4+12 M) P + 91 Pv; own: 96-B record + 8-B rectangle per visible surfel"}
    if kind == "binning":          # B4-6: 44-bit LSD radix sort of (tile | depth) keys = 6 passes over 12-byte pairs
        bits = max(1, (ntiles - 1).bit_length())
        passes = -(-(32 + bits) // 8)
        return {"survey": 12 * R + 2 * 12 * R * passes + 8 * R, "own": 12 * R + 12 * R + 4 * R + 8 * ntiles,
                "formula": "B4-6 = 12 R + 2*12 R ceil((32+bits)/8) + 8 R (%d radix passes); own: keys written once into tile buckets "
                           "(12 R), read once by the per-tile LDS sort (12 R), 4 R sorted ids out" % passes}
    if kind == "surfel_bwd":       # B10
        return {"survey": (232 + 96) * Pv + (12 + 8 + 16 + 12 * M) * Pv, "own": (96 + 80 + 12 + 8 + 16 + 12 * M) * Pv + (36 + 12 * M + 12 + 12 + 12 + 4 + 8 + 16) * Pv,
                "formula": "B10 = (232+96) Pv + (12+8+16+12 M) Pv; own: 96-B record + 80-B accumulator row + inputs in, gradient rows out"}
    raise KeyError(kind)


def forward_only(args, P, H, W, device):
    """configs[1] of BASELINE.json: static canonical surfels rendered forward-only through the operator surface
    (GaussianRasterizer), one view per step, eager launches.  Same JSON contract; the roofline object is the forward blend."""
    import math
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    from dgs_amd.cameras import orbit_cameras
    from dgs_amd.synthetic import activated, make_scene
    xyz, scales, rots, opac, shs = (t.to(device) for t in activated(make_scene(P, seed=0)))
    bg = torch.zeros(3, device=device)
    rasts = []
    for cam in orbit_cameras(64, W, H):
        cam = cam.to(device)
        rasts.append(GaussianRasterizer(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center,
            prefiltered=False, debug=False)))
    m2 = torch.zeros_like(xyz)

    def step(i):
        with torch.no_grad():
            return rasts[i % len(rasts)](means3D=xyz, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    _C.profile_enable(True)
    _C.profile_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = _C.profile_read()
    _C.profile_enable(False)
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    n, ms, S = prof["fwd_n"], prof["fwd_ms"], prof["fwd_S"]
    bytes_per = blend_bytes(S / max(n, 1), ntiles, H * W, backward=False)
    gbs = bytes_per / (ms / max(n, 1) * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"metric": "render views/sec (fwd only), %dx%d, %dk surfels" % (W, H, P // 1000), "value": round(args.steps / dt, 3), "unit": "views/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "c2: static canonical render of S(%d surfels, %dx%d, seed 0), forward only, no deformation" % (P, W, H),
                      "surfels": P, "image": "%dx%d" % (W, H), "sh_degree": 3, "launch": "eager, through GaussianRasterizer.forward"},
           "roofline": {"bound": "hbm", "kernel": "blend_fwd_rows_kernel", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None, "timing": "HIP events on the launch stream, the timed steps themselves",
                        "avg_kernel_ms": round(ms / max(n, 1), 4), "alg_bytes_per_launch": round(bytes_per), "S_per_launch": round(S / max(n, 1))}}
    print(json.dumps(out), flush=True)


def mean_radius(tr):
    """Mean screen radius (px) of the surfels the last step rendered (radii of this rank's view; single-GPU runs keep them raw)."""
    r = tr._radii[:tr.P].float()
    vis = r > 0
    return round(float(r[vis].mean()), 2) if bool(vis.any()) else None


def timed_steps(tr, k, world, densify_every=0, densify_log=None):
    """Exactly k steps between barrier + synchronize on both sides; returns seconds (this rank's clock)."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        tr.step()   # the loss of every step lands in the trainer's pinned report ring (guard kernel): no copy kernel, no sync here;
                    # the reference reads loss.item() -- a host synchronisation -- every step
        if densify_every and (i + 1) % densify_every == 0:
            torch.cuda.synchronize()
            td = time.perf_counter()
            # reference thresholds (arguments/__init__.py:115-122); extent = radius of the camera orbit
            counts = tr.densify_and_prune(0.0002, 0.01, 4.0, 20)
            torch.cuda.synchronize()
            densify_log.append(((time.perf_counter() - td) * 1e3, counts))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _max_over_ranks(x, device):
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def choose_split3(tr, args, world, device, use_graph):
    """N > 1: two or three pieces of the data-parallel step?  (Trainer.split3: the per-surfel gradients leave as a slice of their own
    under the node-MLP backward.)  Which one wins depends on what the links deliver on this box, so both are timed here, before the
    headline -- W warm-up + K steps each on the step as it will be run -- and every rank takes the faster one (the MAX over ranks of
    each timing decides, so all ranks agree)."""
    if world == 1 or not use_graph or not tr._split_ok():
        return None
    res = {}
    for flag in (False, True):
        tr.split3 = flag
        tr._graph = None
        tr.enable_graph(tr._capacity, validate=False)
        for _ in range(args.warmup):
            tr.step()
        res[flag] = _max_over_ranks(timed_steps(tr, args.steps, world), device) / args.steps * 1e3
    pick = res[True] < res[False]
    if tr.split3 != pick:
        tr.split3 = pick
        tr._graph = None
        tr.enable_graph(tr._capacity, validate=False)
    return {"ms_per_step_two_pieces": round(res[False], 4), "ms_per_step_three_pieces": round(res[True], 4), "chosen": "three" if pick else "two"}


def comm_report(tr, args, world, device, ms_headline, split_choice):
    """N > 1: what the step's collectives cost on THIS box, so that the first multi-GPU number explains itself.
      slices: every all-reduce of the step timed alone, back to back (same sizes, same dtype / op, scratch buffers): ms and bus
        bandwidth 2 (N-1)/N x bytes / time -- what a ring moves per link direction, comparable to the 153 GB/s of one xGMI link and to
        the ~230 GB/s the plan in DESIGN.md section 8 needs for >= 6x at N = 8;
      ms_per_step_no_collectives: W + K steps of the same captured step with every collective skipped (Trainer.no_collectives), from a
        snapshot that is restored afterwards (the replicas diverge without their all-reduce);
      exposed_ms_per_step: headline - that = communication the step does not hide."""
    wire = tr.wire_bytes_per_step()
    shard = bool(tr._shard_ok())
    out = {"world": world, "backend": dist.get_backend(), "wire_bytes_per_step": wire, "wire_bf16": bool(tr.wire_bf16), "split3": split_choice,
           "sh_collective": ("reduce_scatter(gradients) -> Adam on P/N rows -> all_gather(parameter), the all-gather under the next step's "
                             "deformation head" if shard else "all_reduce(gradients), Adam on all rows on every rank"), "slices": {}}
    reps = 10
    rank = dist.get_rank()
    for name, nbytes in wire.items():
        if name == "total" or not nbytes:
            continue
        is_radii = name == "radii"
        half = tr.wire_bf16 and name in ("sh", "mid")   # these slices cross as bfloat16 (Trainer.wire_bf16)
        buf = torch.zeros(nbytes // (2 if half else 4), dtype=torch.int32 if is_radii else (torch.bfloat16 if half else torch.float32), device=device)
        op = dist.ReduceOp.MAX if is_radii else dist.ReduceOp.SUM
        c = buf.numel() // world
        kinds = [(name, lambda: dist.all_reduce(buf, op=op), 2.0 * (world - 1) / world)]
        if name == "sh" and shard:   # the sharded SH update: the two halves of the all-reduce as collectives of their own, in place
            gbuf = torch.zeros(nbytes // 4, dtype=torch.float32, device=device) if half else buf   # (the parameter rows always cross as fp32)
            cg = gbuf.numel() // world
            kinds = [("sh_reduce_scatter", lambda: dist.reduce_scatter_tensor(buf[rank * c:(rank + 1) * c], buf, op=op), (world - 1.0) / world),
                     ("sh_all_gather", lambda: dist.all_gather_into_tensor(gbuf, gbuf[rank * cg:(rank + 1) * cg]), (world - 1.0) / world)]
        for kname, fn, factor in kinds:
            kbytes = int(nbytes) if kname != "sh_all_gather" else int(4 * (nbytes // (2 if half else 4)))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ms = _max_over_ranks(time.perf_counter() - t0, device) / reps * 1e3
            out["slices"][kname] = {"bytes": kbytes, "ms": round(ms, 4), "bus_GBs": round(factor * kbytes / (ms * 1e-3) / 1e9, 2)}
        del buf
    snap, it0 = tr._snapshot(), tr.iteration
    tr.no_collectives = True
    try:
        for _ in range(args.warmup):
            tr.step()
        ms_free = _max_over_ranks(timed_steps(tr, args.steps, world), device) / args.steps * 1e3
    finally:
        tr.no_collectives = False
        tr._restore(snap)
        tr.iteration = it0
        if getattr(tr, "_oflag", None) is not None:
            tr._oflag.zero_()
    out["ms_per_step_no_collectives"] = round(ms_free, 4)
    out["exposed_ms_per_step"] = round(ms_headline - ms_free, 4)
    out["views_per_rank"] = int(args.views_per_rank)
    out["exposed_ms_per_view"] = round((ms_headline - ms_free) / args.views_per_rank, 4)   # what Trainer.views_per_rank divides
    out["sum_of_slices_ms"] = round(sum(v["ms"] for v in out["slices"].values()), 4)
    out["levers"] = comm_levers(tr, args, world, device, ms_headline)
    return out


def comm_levers(tr, args, world, device, ms_headline):
    """N > 1: the two levers that trade something for less exposed communication, timed in THIS run (W + K steps each, from a snapshot
    that is restored afterwards -- the headline and everything after it are unaffected), so that "which of them pays on these links"
    needs no second run:  wire_bf16 -- the big slices cross as bfloat16 (a change of the numerics; eager collectives, no re-capture);
    views_per_rank 2 -- two views per rank added before ONE exchange and update (k x N views per step; re-captured and captured back)."""
    if args.views_per_rank != 1 or os.environ.get("DGS_NO_LEVERS", "0") == "1":
        return None
    res = {"ms_per_view_headline": round(ms_headline, 4)}

    def window():
        for _ in range(args.warmup):
            tr.step()
        return _max_over_ranks(timed_steps(tr, args.steps, world), device) / args.steps * 1e3

    def rewind(snap, it0):
        tr._restore(snap)
        tr.iteration = it0
        if getattr(tr, "_oflag", None) is not None:
            tr._oflag.zero_()

    snap, it0 = tr._snapshot(), tr.iteration
    was = tr.wire_bf16
    try:
        tr.wire_bf16 = not was
        ms = window()
        res["wire_bf16_%s" % ("off" if was else "on")] = {"ms_per_view": round(ms, 4), "wire_bytes_per_step": tr.wire_bytes_per_step()["total"]}
    finally:
        tr.wire_bf16 = was
        rewind(snap, it0)
    if tr._graph:
        cap = tr._capacity
        try:
            # back to back | in flight at the same time (a lane each).  The lanes' capture (alias modules, a rasterizer context and a stream per
            # lane) is the one piece of this file that has never met RCCL with more than one rank: opt-in (DGS_LEVER_CONCURRENT=1), so that a
            # first multi-GPU run cannot lose its headline line to the side measurement; the single-GPU line reports concurrent lanes anyway
            kinds = (("views_per_rank_2", False),) + ((("views_per_rank_2_concurrent", True),) if os.environ.get("DGS_LEVER_CONCURRENT", "0") == "1" else ())
            for name, conc in kinds:
                tr.views_per_rank, tr.concurrent_views = 2, conc
                tr._graph = None
                tr.enable_graph(cap, validate=False)
                ms = window()
                res[name] = {"ms_per_step": round(ms, 4), "ms_per_view": round(ms / 2, 4), "views_per_step": 2 * world}
                rewind(snap, it0)
        finally:
            tr.views_per_rank, tr.concurrent_views = 1, False
            rewind(snap, it0)
            tr._graph = None
            tr.enable_graph(cap, validate=False)
    return res


_WATCH = {"lock": threading.Lock(), "done": False, "timer": None}


def arm_watchdog(headline, rank, seconds):
    """Everything behind the timed region (communication report and its levers, drift window, roofline legs, twin check, CPU baseline) is a
    side measurement, and some of it is first-contact code on a multi-GPU box.  If it has not finished after `seconds`, rank 0 prints the
    HEADLINE line alone (same metric / value / config; the objects that were not measured are absent and `note` says why) and every rank
    leaves: a collective that never returns must not cost the run its number.  Every rank arms its own timer right behind the barrier that
    ends the timed region, so they fire together."""
    def bail():
        with _WATCH["lock"]:
            if _WATCH["done"]:
                return
            _WATCH["done"] = True
            if rank == 0:
                print(json.dumps(dict(headline, note="measurements behind the timed region did not finish within %d s: headline only" % seconds)), flush=True)
            sys.stderr.write("bench.py: watchdog after %d s (rank %d)\n" % (seconds, rank))
            sys.stderr.flush()
            os._exit(0)
    t = threading.Timer(seconds, bail)
    t.daemon = True
    t.start()
    _WATCH["timer"] = t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="metric", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-legs", action="store_true",
                    help="skip the extra kernel-timing passes after the timed region (PMC runs of tools/pmc_kernels.sh use this)")
    ap.add_argument("--tile-order", type=int, default=None,
                    help="dispatch order of the blend tiles (dgs_set_option(1, m): 3 longest first, 4 XCD-local groups): A/B runs only")
    ap.add_argument("--densify-every", type=int, default=0,
                    help="also run the in-place densification every N timed steps (off by default: the metric is the plain step)")
    ap.add_argument("--slots-factor", type=float, default=1.5, help="surfel slots per initial surfel when --densify-every is on")
    ap.add_argument("--pre-iterations", type=int, default=10000, help="--workload trained: fit() iterations before the timed region")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="opt-in lever of the data-parallel step (Trainer.views_per_rank): k views per rank added before ONE exchange and update; "
                         "the line then reports views_per_step = k * N and the metric counts views, not steps")
    ap.add_argument("--drift-gap", type=int, default=100, help="steps between the headline window and the second (drift) window; 0: skip")
    ap.add_argument("--concurrent-report", type=float, default=None, metavar="MS_HEADLINE",
                    help="internal: print only the concurrent_views object (the main run starts this as a child process, so that nothing in it can cost the headline line)")
    args = ap.parse_args()
    if args.concurrent_report is not None:
        torch.cuda.set_device(0)
        P_, H_, W_ = WORKLOADS[args.workload]
        print(json.dumps(concurrent_views_report(args, P_, H_, W_, torch.device("cuda", 0), args.concurrent_report)), flush=True)
        return
    # the per-step losses of the timed steps are read back from the step guard's pinned ring: size it for the run asked for
    from dgs_amd.train import Trainer as _Trainer
    _Trainer.GUARD_RING = max(_Trainer.GUARD_RING, 2 * (args.steps + args.warmup) + 64)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the surfel rasterizer has no CPU path")
    # one rank per GPU; on a box with fewer GPUs than ranks (tests: two ranks on the one GPU of the test box) ranks share devices
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        # RCCL ("nccl") is the product path; DGS_DIST_BACKEND=gloo is for the multi-rank test of THIS file on a one-GPU box
        # (RCCL refuses two ranks on one device)
        backend = os.environ.get("DGS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)

    from diff_surfel_rasterization import _C
    from dgs_amd import _ops as _train_ops
    if world > 1:
        # a stale or hash-less binary is rebuilt by whoever loads it (_C._refuse_stale); _dgs_build serialises that with a file lock
        # and an atomic rename, and here one rank per node does the loading first so that the others never wait inside hipcc
        if local_rank == 0:
            _C.load()
            _train_ops.load()
        dist.barrier()
    _C.load()
    _train_ops.load()
    if args.tile_order is not None:
        _C.set_option(1, args.tile_order)
    P, H, W = WORKLOADS[args.workload]
    if args.workload == "c2":
        if world > 1:
            raise SystemExit("--workload c2 is the single-GPU forward-only case")
        return forward_only(args, P, H, W, device)
    # wake the device up (clocks, allocator) before anything is timed: fresh boxes occasionally ran the first
    # second of work several times slower
    _a = torch.randn(4096, 4096, device=device)
    _t = time.perf_counter()
    while time.perf_counter() - _t < 0.5:
        _a = (_a @ _a).clamp_(-1, 1)
    torch.cuda.synchronize()
    del _a
    use_graph = os.environ.get("DGS_NO_GRAPHS", "0") != "1"
    pre_losses = None
    if args.workload == "trained":
        if world > 1:
            raise SystemExit("--workload trained is a single-GPU workload (its preparation is not sharded)")
        tr, pre_losses = trained_trainer(P, H, W, device, args.pre_iterations)
        tr.set_regime(warmup=False, lambda_normal=0.02, lambda_dist=1000.0)   # the timed step is the late-regime step, like "metric"
        if use_graph and not tr._graph:
            tr.enable_graph(capacity=96 * tr.P)
        P = tr.P
    else:
        tr = build_trainer(P, H, W, device, slots=int(args.slots_factor * P) if args.densify_every else None, views_per_rank=args.views_per_rank)
        if use_graph:
            # whole-step HIP graphs: the rasterizer runs in capacity mode (no device->host read), 24 list entries per
            # surfel is ~3x what this scene needs
            tr.enable_graph(capacity=24 * P)
    # W untimed steps, then exactly K timed ones.  The scene trains while it is timed (Adam moves every log-scale by ~lr per
    # step on the noise targets: the splats grow by ~10 % in screen radius over 50 steps), so a long run can outgrow what graph
    # capture promised the rasterizer (longest tile list, list capacity): the trainer's step guard then skips that step,
    # re-captures and renders the view again (Trainer._recover_overflow).  A timed region in which that happened contains a
    # re-capture and is not reported: warm-up + timed region are run again on the re-captured step (at most twice).
    views_first_step = [tr.view_for(0, j) for j in range(args.views_per_rank)]
    if world > 1:   # the views every rank renders in step 0 (the shared schedule must give every rank its own)
        vt = torch.tensor(views_first_step, dtype=torch.int64, device=device)
        vall = [torch.zeros_like(vt) for _ in range(world)]
        dist.all_gather(vall, vt)
        views_first_step = [int(x) for v in vall for x in v.tolist()]
    split_choice = choose_split3(tr, args, world, device, use_graph)
    attempts = 0
    while True:
        attempts += 1
        recoveries = tr.overflow_recoveries
        for _ in range(args.warmup):
            tr.step()
        densify_log = []
        dt = timed_steps(tr, args.steps, world, args.densify_every, densify_log)
        clean = not use_graph or (tr.overflow_recoveries == recoveries and not bool(tr._oflag.item() if getattr(tr, "_oflag", None) is not None else 0))
        if world > 1:   # one verdict for all ranks (a rank that overflowed in the last steps knows before the others)
            c = torch.tensor([1 if clean else 0], dtype=torch.int32, device=device)
            dist.all_reduce(c, op=dist.ReduceOp.MIN)
            clean = bool(c.item())
        if clean:
            break
        if attempts == 3:
            raise SystemExit("rasterizer capacity overflow in three consecutive timed regions: result invalid")
        if rank == 0:
            print("note: the step was re-captured during the timed region (tile lists outgrew the capture); timing again", file=sys.stderr)
        for _ in range(tr.GUARD_LAG + 1):   # let the guard see the last steps of the region (it polls with a lag)
            tr.step()
        torch.cuda.synchronize()
    timed_losses = tr.loss_history(args.steps)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    radius_headline = mean_radius(tr)
    headline = {"metric": "train views/sec (fwd+bwd), 800x800, 200k surfels" if args.workload == "metric" else "train views/sec (fwd+bwd), %dx%d, %dk surfels" % (W, H, P // 1000),
                "value": round(args.views_per_rank * world * args.steps / dt, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s: synthetic scene S(%d surfels, %dx%d, seed 0), full train step, 1 view per GPU per step" % (args.workload, P, W, H),
                           "views_per_step": args.views_per_rank * world, "views_first_step": views_first_step, "parallelism": "dp%d" % world}}
    arm_watchdog(headline, rank, int(os.environ.get("DGS_BENCH_WATCHDOG_S", "600")))
    comm = comm_report(tr, args, world, device, dt / args.steps * 1e3, split_choice) if world > 1 else None

    # ---- everything below is measured AFTER the headline and does not change it --------------------------------------------
    # The scene state the headline window ended with is kept: the drift leg trains on, the roofline legs and the twin check
    # are run from this snapshot again.
    snap, it0 = tr._snapshot(), tr.iteration
    drift = None
    if args.drift_gap > 0 and not args.no_roofline_legs and not args.densify_every:
        recoveries = tr.overflow_recoveries
        for _ in range(args.drift_gap):
            tr.step()
        dt2 = timed_steps(tr, args.steps, world)
        if world > 1:
            t2 = torch.tensor([dt2], dtype=torch.float64, device=device)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            dt2 = float(t2.item())
        drift = {"steps_between_windows": args.drift_gap, "ms_per_step_second_window": round(dt2 / args.steps * 1e3, 4),
                 "mean_screen_radius_px": [radius_headline, mean_radius(tr)], "recaptured_in_between": tr.overflow_recoveries != recoveries,
                 "note": ("real target views: the scene keeps training; the headline is the first window of the run" if args.workload == "trained" else
                          "noise targets: Adam grows every splat by ~lr per step; the headline is the first window of the run")}

    # ---- roofline legs = the twin check.  From the snapshot: K graph-replayed steps with the library's device-timestamp hook on
    # (one-thread kernels before and after each timed launch append the 100 MHz device counter to a ring -- legal inside a captured
    # graph, where HIP events are not); from the snapshot again: the same K steps launched eagerly with HIP events around the
    # launches.  Same parameters, same Adam state, same views: the two loss series must agree step by step.
    prof_graph = prof = None
    twin = None
    if not args.no_roofline_legs and not args.densify_every:
        def rewind():
            tr._restore(snap)
            tr.iteration = it0
            if getattr(tr, "_oflag", None) is not None:
                tr._oflag.zero_()
        replay_losses = None
        if use_graph:
            rewind()
            _C.profile_enable(2)
            tr._graph = None
            tr.enable_graph(capacity=tr._capacity)
            torch.cuda.synchronize()
            _C.profile_reset()      # drop the stamps of the capture's warm-up launches: only replays are counted
            for _ in range(args.steps):
                tr.step()
            replay_losses = tr.loss_history(args.steps)
            prof_graph = _C.profile_read()
            _C.profile_enable(0)
        rewind()
        tr._graph = None
        _C.set_capacity(0)
        _C.set_option(6, 0)
        _C.profile_enable(1)
        _C.profile_reset()
        eager_losses = [float(tr.step()) for _ in range(args.steps)]
        torch.cuda.synchronize()
        prof = _C.profile_read()
        _C.profile_enable(0)
        if replay_losses is not None:
            # yardstick: the same eager steps a second time.  Two eager runs differ in the order of the float atomics of the backward
            # only, and so do a replay and an eager run -- a correct replay is as close to eager as eager is to itself
            rewind()
            eager2_losses = [float(tr.step()) for _ in range(args.steps)]
            torch.cuda.synchronize()
            relf = lambda xs, ys: [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(xs, ys)]
            rels, noise = relf(replay_losses, eager_losses), max(relf(eager2_losses, eager_losses))
            rel, rel_first = max(rels), max(rels[:3])
            # identical to ~1e-6 at first, then the trajectories drift apart (Adam turns the sign of a gradient that is zero to
            # rounding into a full step): 2e-6 after 20 steps on the metric workload, 1e-4 .. 2e-3 on the fast-moving "trained" one,
            # eager against eager just the same.  A replay that mis-orders a node is off by O(1) from the first step
            # (the floor is 1e-2 on "trained": its two eager runs differ by 1.2e-3 .. 3.9e-3 themselves, a chaotic divergence with a heavy
            # tail -- a single draw of it is a poor yardstick for another draw, and round 6 lost one of four runs to a 1e-3 floor)
            floor = 1e-2 if args.workload == "trained" else 1e-3
            tol_all = max(floor, 4.0 * noise)
            twin = {"steps": args.steps, "max_rel_loss_difference": float("%.3g" % rel), "max_rel_loss_difference_first_3_steps": float("%.3g" % rel_first),
                    "eager_vs_eager_max_rel_loss_difference": float("%.3g" % noise),
                    "tolerance": {"first_3_steps": 1e-5, "all_steps": float("%.3g" % tol_all), "rule": "max(%g, 4 x eager-vs-eager)" % floor},
                    "what": "graph-replayed vs eager steps from the same snapshot (parameters, Adam state, views); float atomics in the backward are the only difference"}
            twin_bad = not (rel <= tol_all and rel_first <= 1e-5) or any(l != l for l in replay_losses + timed_losses)
            if world > 1:   # one verdict for all ranks: a rank that left alone would leave the others inside a collective
                twin_bad = _max_over_ranks(1.0 if twin_bad else 0.0, device) > 0.0
            if twin_bad:
                raise SystemExit("graph-replayed steps disagree with their eager twin (max relative loss difference %.3g, eager vs eager %.3g; replay %s, eager %s): result invalid"
                                 % (rel, noise, ["%.6f" % l for l in replay_losses], ["%.6f" % l for l in eager_losses]))

    if rank == 0:
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)
        M = 16

        # HBM traffic and instruction counts per launch from the committed PMC passes (tools/pmc_kernels.sh: separate rocprofv3
        # --pmc runs of THIS command, corrected as MI355X_MICROARCH.md prescribes).  Reported only when the file was taken on this
        # workload AND on the library sources that are loaded now (hash of sources + flags).
        pmc, pmc_src = {}, None
        # the newest round's file for this workload (profiles/rNN_pmc_<workload>.json); a file of other sources is named, not used
        import glob as _glob
        _cands = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % args.workload)))
        pmc_path = os.path.join("profiles", os.path.basename(_cands[-1])) if _cands else os.path.join("profiles", "r04_pmc_%s.json" % args.workload)
        try:
            with open(os.path.join(ROOT, pmc_path)) as f:
                pj = json.load(f)
            if pj.get("workload") == args.workload and pj.get("library_source_hash") == _C.source_hash():
                pmc = pj["kernels"]
                pmc_src = "%s (bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes of this command; library sources %s)" % (
                    pmc_path, pj["library_source_hash"])
            else:
                pmc_src = "%s is for workload %s / library sources %s, loaded library is %s: not reported" % (
                    pmc_path, pj.get("workload"), pj.get("library_source_hash"), _C.source_hash())
        except Exception as ex:
            pmc_src = "no PMC profile for this workload (%s)" % type(ex).__name__

        # the blend kernels as the library launches them (the forward walks one list per 16-lane row since round 4)
        KNAME = {"fwd": "blend_fwd_rows_kernel", "bwd": "blend_bwd_kernel"}

        def pmc_of(kernel, counter):
            return pmc.get("dgs::" + kernel, {}).get(counter)

        def duration(key):
            """(average ms per launch, launches, which clock) from the in-graph leg when there is one, else the eager leg"""
            for src, label in ((prof_graph, "device timestamps (100 MHz counter) around the kernel inside the replayed whole-step graph"),
                               (prof, "HIP events on the launch stream, eager launches")):
                if src and src[key + "_n"] and src[key + "_ms"] > 0:
                    return src[key + "_ms"] / src[key + "_n"], src[key + "_n"], src, label
            return None

        def roof(kind):
            d = duration(kind)
            if not d:
                return None
            ms, n, src, label = d
            S = src[kind + "_S"] / n
            bytes_per = blend_bytes(S, ntiles, H * W, backward=(kind == "bwd"))
            gbs = bytes_per / (ms * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": KNAME[kind], "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": pmc_of(KNAME[kind], "hbm_traffic_bytes_per_launch"),
                 "traffic_source": pmc_src, "timing": "%s, %d launches" % (label, n),
                 "avg_kernel_ms": round(ms, 4), "alg_bytes_per_launch": round(bytes_per), "S_per_launch": round(S)}
            if src is prof_graph and prof and prof[kind + "_n"]:
                r["avg_kernel_ms_eager_events"] = round(prof[kind + "_ms"] / prof[kind + "_n"], 4)
            return r

        def roof_valu(kind):
            d = duration(kind)
            insts = pmc_of(KNAME[kind], "SQ_INSTS_VALU")
            if not d or not insts:
                return None
            ms = d[0]
            peak = 0.5 * 1024 * 2.4e9   # MI355X_MICROARCH.md: a wave64 VALU op issues over 2 cycles on a SIMD-32; 1024 SIMDs; 2.4 GHz
            ach = insts / (ms * 1e-3)
            return {"bound": "valu_issue", "kernel": KNAME[kind], "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2),
                    "unit": "G wave64 VALU instructions/s", "frac": round(ach / peak, 4), "valu_instructions_per_launch": insts,
                    "salu_instructions_per_launch": pmc_of(KNAME[kind], "SQ_INSTS_SALU"), "avg_kernel_ms": round(ms, 4),
                    "cycles_per_instruction_per_simd": round(1024 * 2.4e9 * ms * 1e-3 / insts, 2),
                    "measured_issue_ceiling": "2.5 cycles per plain FMA / mul / add, 4.4-4.8 with an SGPR operand, for comparisons, min/max, "
                                              "selects, DPP; 8.5-12.7 for v_rcp / v_exp (profiles/r03_valu_issue_gfx950.txt)",
                    "source": pmc_src}

        def roof_kernel(name, key):
            d = duration(key)
            if not d:
                return None
            ms, n, src, label = d
            R, Pv = src["R"] / max(src["pre_n"], 1), src["Pv"] / max(src["pre_n"], 1)
            b = kernel_bytes(name, P, Pv, R, M, ntiles)
            # binning: the SURVEY figure prices the reference's 6-pass radix sort; this implementation buckets once and sorts in LDS, so
            # the fraction is taken on the bytes it actually moves (the SURVEY figure stays next to it)
            basis = "own" if name == "binning" else "survey"
            gbs = b[basis] / (ms * 1e-3) / 1e9
            kernels = {"preprocess_fwd": ["preprocess_fwd_kernel"], "surfel_bwd": ["surfel_bwd_kernel"],
                       "binning": ["count_tiles_lds_kernel", "bin_offsets_kernel", "scatter_keys_lds_kernel", "sort_tiles_radix_kernel"]}[name]
            traffic = [pmc_of(k, "hbm_traffic_bytes_per_launch") for k in kernels]
            return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                    "traffic": None if any(t is None for t in traffic) else int(sum(traffic)),
                    "avg_ms": round(ms, 4), "frac_basis": "own_bytes_per_launch" if basis == "own" else "alg_bytes_per_launch (SURVEY 8d)",
                    "alg_bytes_per_launch": round(b["survey"]), "own_bytes_per_launch": round(b["own"]),
                    "own_GBs": round(b["own"] / (ms * 1e-3) / 1e9, 2), "survey_GBs": round(b["survey"] / (ms * 1e-3) / 1e9, 2),
                    "survey_frac": round(b["survey"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "formula": b["formula"], "R": round(R), "P_visible": round(Pv),
                    "timing": "%s, %d launches" % (label, n)}

        out = {
            "metric": "train views/sec (fwd+bwd), 800x800, 200k surfels" if args.workload == "metric"
            else "train views/sec (fwd+bwd), %dx%d, %dk surfels" % (W, H, P // 1000),
            "value": round(args.views_per_rank * world * args.steps / dt, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s: synthetic scene S(%d surfels, %dx%d, seed 0), full train step (node deform + surfel "
                                    "raster fwd/bwd + L1/D-SSIM/normal/distortion loss + Adam), 1 view per GPU per step" % (args.workload, P, W, H))
                       if args.workload != "trained" else
                       ("trained: DynamicTruth (bobbing sphere + swinging plate, 40k textured surfels) rendered to a D-NeRF-format dataset at %dx%d "
                        "and fitted for %d iterations from 100k random points with the reference's stages, densification and opacity resets "
                        "(%d live surfels in %d slots now); the timed step is the same full late-regime train step as 'metric', "
                        "on the dataset's target views" % (W, H, args.pre_iterations, tr.surfels.num_surfels, tr.P)),
                       "surfels": P, "image": "%dx%d" % (W, H), "sh_degree": 3, "control_nodes": int(tr.deform.node_num),
                       "views_per_step": args.views_per_rank * world, "views_per_rank": args.views_per_rank, "views_first_step": views_first_step,
                       "parallelism": ("dp%d (views sharded; SH gradients reduce-scattered to the rows' owners, updated rows all-gathered; the rest all-reduced)"
                                       if (comm or {}).get("sh_collective", "").startswith("reduce_scatter") else "dp%d (views sharded, one flat all-reduce)") % world,
                       "launch": "whole-step HIP graph replay" if use_graph else "eager",
                       "neighbour_search": "%s (spatial share of the K-th neighbour distance %.2f)"
                                           % (tr.deform.knn_refine_mode, getattr(tr.deform, "knn_spatial_share", float("nan")))},
            "roofline": roof("bwd"), "roofline_fwd": roof("fwd"),
            "roofline_valu": {"bwd": roof_valu("bwd"), "fwd": roof_valu("fwd")},
            "roofline_kernels": {"preprocess_fwd": roof_kernel("preprocess_fwd", "pre"), "binning": roof_kernel("binning", "bin"),
                                 "surfel_bwd": roof_kernel("surfel_bwd", "sbw")},
            "twin": twin, "drift": drift,
        }
        if comm is not None:
            out["comm"] = comm
        if pre_losses is not None:
            k = max(len(pre_losses) // 10, 1)
            out["config"]["pre_training_loss"] = {"first_tenth_mean": round(sum(pre_losses[:k]) / k, 5), "last_tenth_mean": round(sum(pre_losses[-k:]) / k, 5)}
            out["config"]["pre_fit"] = getattr(tr, "pre_fit_mode", "float atomics")
            # fingerprint of the fitted scene: equal from run to run exactly when the pre-fit is reproducible
            out["config"]["pre_fit_fingerprint"] = {"live_surfels": int(tr.surfels.num_surfels), "last_loss": float("%.9g" % pre_losses[-1]),
                                                    "loss_sum": float("%.12g" % sum(pre_losses))}
        if args.densify_every:
            out["densify"] = {"every": args.densify_every, "calls": len(densify_log), "ms_per_call": [round(m, 3) for m, _ in densify_log],
                              "cloned_split_pruned": [list(map(int, c)) for _, c in densify_log], "slots": tr.P,
                              "surfels_after": tr.surfels.num_surfels, "recaptured": tr.P != int(args.slots_factor * P)}
        try:
            ceil = measured_hbm_ceiling(device)
            for r in (out["roofline"], out["roofline_fwd"]):
                if r:
                    r["measured_hbm_ceiling"] = {"copy_GBs": ceil["copy"], "triad_GBs": ceil["triad"],
                                                 "method": "torch d2d copy / triad over 1 GiB fp32 buffers, 20 iterations, HIP events"}
        except Exception as ex:   # never lose the result line over the side measurement
            print("warning: HBM ceiling measurement failed: %r" % (ex,), file=sys.stderr)
        if world == 1 and use_graph and args.views_per_rank == 1 and not args.no_roofline_legs and not args.densify_every and args.workload != "trained":
            del tr
            torch.cuda.empty_cache()
            tr = None
            # in a CHILD process: nothing in the side measurement -- an exception, or a crash inside graph instantiation, which the
            # nested-fork form of the lanes' graph produced on ROCm 7.2 -- can cost the line of the headline
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--concurrent-report", "%.6f" % (dt / args.steps * 1e3), "--workload", args.workload,
                                     "--steps", str(args.steps), "--warmup", str(args.warmup)], capture_output=True, text=True, timeout=600)
                lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                out["concurrent_views"] = json.loads(lines[-1]) if cp.returncode == 0 and lines else {"error": "child exited with %d: %s" % (cp.returncode, cp.stderr[-300:])}
            except Exception as ex:
                out["concurrent_views"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_cpu_baseline and args.workload != "trained":
            del tr
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(P, H, W)
        with _WATCH["lock"]:
            if _WATCH["done"]:
                return
            _WATCH["done"] = True
            print(json.dumps(out), flush=True)
    if _WATCH["timer"] is not None:
        _WATCH["timer"].cancel()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
