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
  "roofline"     : achieved algorithmic HBM GB/s of the dominant kernel (the backward blend), timed with HIP
                   events on the launch stream inside the timed region, against the 8 TB/s HBM3E peak;
  "roofline_fwd" : the same for the forward blend;
  "cpu_baseline" : the CPU port (C oracle rasterizer with OpenMP + PyTorch-CPU deformation/loss/Adam) timed on
                   this box's host cores for a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
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
}


def build_trainer(P, H, W, device, n_views=64, n_targets=8, rasterizer_cls=None, fused_adam=None, packed_sh=None, slots=None,
                  sort_surfels=None):
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
    tr = Trainer(surfels, deform, cams, targets, bg, rasterizer_cls=rasterizer_cls, fused_adam=fused_adam)
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
    """The oracle-backed CPU port of the same train step, bounded sample (about 10-30 s of CPU work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_raster_op import OracleRasterizer  # test-only operator: oracle/ is the checker, never shipped
    from oracle import surfel_oracle
    ncores = min(os.cpu_count() or 1, 32)  # more threads only add contention on the accumulations
    torch.set_num_threads(ncores)
    surfel_oracle.set_threads(ncores)
    tr = build_trainer(P, H, W, torch.device("cpu"), n_views=8, n_targets=2, rasterizer_cls=OracleRasterizer, fused_adam=False)
    tr.step()  # untimed: the first call pays one-off BLAS/oneDNN initialisation (tens of seconds on some hosts)
    t0 = time.time()
    tr.step()
    n = 1
    per = time.time() - t0
    while n < 3 and (time.time() - t0) + per < budget_s:
        tr.step()
        n += 1
        per = (time.time() - t0) / n
    dt = time.time() - t0
    return {"value": n / dt, "unit": "views/s", "cores": ncores, "kind": "port",
            "sample": "%d full train steps (views) of the same %dk-surfel %dx%d workload, %.1f s" % (n, P // 1000, W, H, dt)}


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
           "roofline": {"bound": "hbm", "kernel": "blend_fwd_kernel", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None, "timing": "HIP events on the launch stream, the timed steps themselves",
                        "avg_kernel_ms": round(ms / max(n, 1), 4), "alg_bytes_per_launch": round(bytes_per), "S_per_launch": round(S / max(n, 1))}}
    print(json.dumps(out), flush=True)


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
    args = ap.parse_args()

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
    tr = build_trainer(P, H, W, device, slots=int(args.slots_factor * P) if args.densify_every else None)
    use_graph = os.environ.get("DGS_NO_GRAPHS", "0") != "1"
    if use_graph:
        # whole-step HIP graphs: the rasterizer runs in capacity mode (no device->host read), 24 list entries per
        # surfel is ~3x what this scene needs
        tr.enable_graph(capacity=24 * P)
    # W untimed steps, then exactly K timed ones.  The scene trains while it is timed (Adam moves every log-scale by ~lr per
    # step on the noise targets: the splats grow by ~10 % in screen radius over 50 steps), so a long run can outgrow what graph
    # capture promised the rasterizer (longest tile list, list capacity): the trainer's step guard then skips that step,
    # re-captures and renders the view again (Trainer._recover_overflow).  A timed region in which that happened contains a
    # re-capture and is not reported: warm-up + timed region are run again on the re-captured step (at most twice).
    attempts = 0
    while True:
        attempts += 1
        recoveries = tr.overflow_recoveries
        for _ in range(args.warmup):
            tr.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        densify_ms, densify_counts = [], []
        for i in range(args.steps):
            tr.step()   # the loss of every step lands in the trainer's pinned report ring (guard kernel): no copy kernel, no sync here;
                        # the reference reads loss.item() -- a host synchronisation -- every step
            if args.densify_every and (i + 1) % args.densify_every == 0:
                torch.cuda.synchronize()
                td = time.perf_counter()
                # reference thresholds (arguments/__init__.py:115-122); extent = radius of the camera orbit
                densify_counts.append(tr.densify_and_prune(0.0002, 0.01, 4.0, 20))
                torch.cuda.synchronize()
                densify_ms.append((time.perf_counter() - td) * 1e3)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
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

    # ---- roofline legs (after the timed region; the headline value above is not affected) -------------------------------
    # (1) in-graph: the step is re-captured with the library's device-timestamp hook on (one-thread kernels before and after
    #     each blend launch append the 100 MHz device counter to a ring -- legal inside a captured graph, where HIP events
    #     are not) and the same K steps are replayed: the blend kernels are timed in the launch mode the headline uses;
    # (2) eager: the same K steps launched one by one with HIP events around the blend launches -- the classic clock, kept
    #     as a cross-check (the device idles between eager launches and the kernels run a few % faster there).
    prof_graph = prof = None
    eager_losses = []
    if not args.no_roofline_legs:
        if use_graph:
            _C.profile_enable(2)
            tr._graph = None
            tr.enable_graph(capacity=tr._capacity)
            torch.cuda.synchronize()
            _C.profile_reset()      # drop the stamps of the capture's warm-up launches: only replays are counted
            for _ in range(args.steps):
                tr.step()
            torch.cuda.synchronize()
            prof_graph = _C.profile_read()
            _C.profile_enable(0)
        tr._graph = None
        _C.set_capacity(0)
        _C.set_option(6, 0)
        _C.profile_enable(1)
        _C.profile_reset()
        eager_losses = [float(tr.step()) for _ in range(args.steps)]
        torch.cuda.synchronize()
        prof = _C.profile_read()
        _C.profile_enable(0)
        # self-check of the graph replay: the eager steps continue the same training run, so the two loss series must agree
        ref = sum(eager_losses) / len(eager_losses)
        if not all(l == l and 0.8 * ref <= l <= 1.25 * ref for l in timed_losses):
            raise SystemExit("graph-replayed steps disagree with eager steps (losses %s vs eager mean %.5f): result invalid"
                             % (["%.4f" % l for l in timed_losses], ref))

    if rank == 0:
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)

        # HBM traffic per launch from the committed PMC passes (tools/pmc_kernels.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate runs of THIS command, corrected as MI355X_MICROARCH.md prescribes).  Reported only when the file was
        # taken on this workload AND on the library sources that are loaded now (hash of sources + flags).
        traffic, traffic_src = {}, None
        pmc_path = os.path.join("profiles", "r02_pmc_blend_%s.json" % args.workload)
        try:
            with open(os.path.join(ROOT, pmc_path)) as f:
                pmc = json.load(f)
            if pmc.get("workload") == args.workload and pmc.get("library_source_hash") == _C.source_hash():
                traffic = {"fwd": pmc["kernels"]["dgs::blend_fwd_kernel"]["hbm_traffic_bytes_per_launch"],
                           "bwd": pmc["kernels"]["dgs::blend_bwd_kernel"]["hbm_traffic_bytes_per_launch"]}
                traffic_src = "%s (bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes of this command; library sources %s)" % (
                    pmc_path, pmc["library_source_hash"])
            else:
                traffic_src = "%s is for workload %s / library sources %s, loaded library is %s: not reported" % (
                    pmc_path, pmc.get("workload"), pmc.get("library_source_hash"), _C.source_hash())
        except Exception as ex:
            traffic_src = "no PMC profile for this workload (%s)" % type(ex).__name__

        def roof(kind):
            # the fraction is computed from the in-graph duration when the step is graph-replayed (the headline's launch mode)
            main, other = (prof_graph, prof) if prof_graph and prof_graph[kind + "_n"] else (prof, None)
            if not main or main[kind + "_n"] == 0 or main[kind + "_ms"] <= 0:
                return None
            n, ms, S = main[kind + "_n"], main[kind + "_ms"], main[kind + "_S"]
            bytes_per = blend_bytes(S / n, ntiles, H * W, backward=(kind == "bwd"))
            gbs = bytes_per / (ms / n * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": "blend_%s_kernel" % kind, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic.get(kind), "traffic_source": traffic_src,
                 "timing": ("device timestamps (100 MHz counter) around the kernel inside the replayed whole-step graph, %d launches" % n)
                 if main is prof_graph else ("HIP events on the launch stream, eager launches, %d launches" % n),
                 "avg_kernel_ms": round(ms / n, 4), "alg_bytes_per_launch": round(bytes_per), "S_per_launch": round(S / n)}
            if other and other[kind + "_n"]:
                r["avg_kernel_ms_eager_events"] = round(other[kind + "_ms"] / other[kind + "_n"], 4)
            return r

        out = {
            "metric": "train views/sec (fwd+bwd), 800x800, 200k surfels" if args.workload == "metric"
            else "train views/sec (fwd+bwd), %dx%d, %dk surfels" % (W, H, P // 1000),
            "value": round(world * args.steps / dt, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic scene S(%d surfels, %dx%d, seed 0), full train step (node deform + surfel "
                                   "raster fwd/bwd + L1/D-SSIM/normal/distortion loss + Adam), 1 view per GPU per step" % (args.workload, P, W, H),
                       "surfels": P, "image": "%dx%d" % (W, H), "sh_degree": 3, "control_nodes": 1024,
                       "views_per_step": world, "parallelism": "dp%d (views sharded, one flat all-reduce)" % world,
                       "launch": "whole-step HIP graph replay" if use_graph else "eager"},
            "roofline": roof("bwd"), "roofline_fwd": roof("fwd"),
        }
        if args.densify_every:
            out["densify"] = {"every": args.densify_every, "calls": len(densify_ms), "ms_per_call": [round(m, 3) for m in densify_ms],
                              "cloned_split_pruned": [list(map(int, c)) for c in densify_counts], "slots": tr.P,
                              "surfels_after": tr.surfels.num_surfels, "recaptured": tr.P != int(args.slots_factor * P)}
        try:
            ceil = measured_hbm_ceiling(device)
            for r in (out["roofline"], out["roofline_fwd"]):
                if r:
                    r["measured_hbm_ceiling"] = {"copy_GBs": ceil["copy"], "triad_GBs": ceil["triad"],
                                                 "method": "torch d2d copy / triad over 1 GiB fp32 buffers, 20 iterations, HIP events"}
        except Exception as ex:   # never lose the result line over the side measurement
            print("warning: HBM ceiling measurement failed: %r" % (ex,), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            del tr
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(P, H, W)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
