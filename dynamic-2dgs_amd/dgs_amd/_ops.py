"""ctypes binding of libdgs_train_ops.so (include/dgs_train_ops.h): fused SSIM and brute-force KNN kernels for
gfx950, used by dgs_amd.losses / dgs_amd.deform when their inputs live on a HIP device.  CPU tensors keep using
the PyTorch formulation (that is what the CPU tests and bench.py's cpu_baseline leg run)."""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.environ.get("DGS_TRAIN_OPS_LIB", os.path.join(_CSRC, "libdgs_train_ops.so"))  # override: development A/B builds only
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-munsafe-fp-atomics"]
_lib = None
_EXPORTS = ("dgs_train_ops_abi_version", "dgs_train_ops_last_error", "dgs_ssim_forward", "dgs_ssim_backward", "dgs_knn_points",
            "dgs_lbs_scratch_bytes", "dgs_lbs_forward", "dgs_lbs_backward", "dgs_adam_plan_bytes", "dgs_adam_plan", "dgs_adam_step",
            "dgs_regloss_forward", "dgs_regloss_backward", "dgs_mlp_packed_floats", "dgs_mlp_saved_floats", "dgs_mlp_scratch_floats",
            "dgs_mlp_forward", "dgs_mlp_backward", "dgs_knn_points2", "dgs_deform_forward", "dgs_deform_backward", "dgs_photo_forward",
            "dgs_photo_backward", "dgs_loss_combine", "dgs_densify_view", "dgs_densify_accumulate", "dgs_knn_refine", "dgs_photo_blocks", "dgs_regloss_blocks", "dgs_regloss_forward_partials", "dgs_adam_step_pattern", "dgs_adam_step_sched", "dgs_lbs_supported", "dgs_regloss_backward_slot",
            "dgs_step_guard", "dgs_adam_step_guarded", "dgs_adam_step_zero", "dgs_densify_accumulate_guarded", "dgs_regloss_forward_partials_z",
            "dgs_regloss_fused", "dgs_regloss_fused_blocks", "dgs_photo_backward_combine", "dgs_knn_refine_mode", "dgs_deform_reduce", "dgs_photo_backward_combine_guard",
            "dgs_adam_step_origin", "dgs_select_row", "dgs_loss_forward_merged", "dgs_mlp_forward_select", "dgs_mlp_backward_reduce",
            "dgs_adam_step_sum2")


def _deps():
    hdr = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include", "dgs_train_ops.h")
    return [os.path.join(_CSRC, "train_ops.hip"), hdr, os.path.join(_CSRC, "node_mlp.h"), os.path.join(_CSRC, "wave_reduce.h")]


def source_hash():
    import _dgs_build
    return _dgs_build.source_hash(_deps(), HIPCC_FLAGS)


def build(force=False, verbose=False):
    """hipcc, in-tree; rebuilt whenever the hash of sources + flags differs from the one recorded with the binary."""
    import _dgs_build
    cmd = ["hipcc"] + HIPCC_FLAGS + [os.path.join(_CSRC, "train_ops.hip"), "-o", LIB_PATH]
    return _dgs_build.build(LIB_PATH, cmd, _deps(), HIPCC_FLAGS, _CSRC, force=force, verbose=verbose)[0]


def exported_symbols():
    return _EXPORTS


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found; build it with __graft_entry__.build()" % LIB_PATH)
        from diff_surfel_rasterization._C import _refuse_stale
        _refuse_stale(LIB_PATH, source_hash, build)   # never run a binary built from other sources than the tree's
        lib = ctypes.CDLL(LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.dgs_train_ops_abi_version.restype = ci
        lib.dgs_train_ops_last_error.restype = ctypes.c_char_p
        lib.dgs_ssim_forward.restype = ci
        lib.dgs_ssim_forward.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_ssim_backward.restype = ci
        lib.dgs_ssim_backward.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_knn_points.restype = ci
        lib.dgs_knn_points.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp]
        lib.dgs_lbs_scratch_bytes.restype = ctypes.c_size_t
        lib.dgs_lbs_scratch_bytes.argtypes = [ci, ci]
        lib.dgs_lbs_forward.restype = ci
        lib.dgs_lbs_forward.argtypes = [ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_lbs_backward.restype = ci
        lib.dgs_lbs_backward.argtypes = [ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_regloss_forward.restype = ci
        lib.dgs_regloss_forward.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp]
        lib.dgs_regloss_backward.restype = ci
        lib.dgs_regloss_backward.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.dgs_adam_plan_bytes.restype = ctypes.c_size_t
        lib.dgs_adam_plan_bytes.argtypes = [ctypes.c_longlong]
        lib.dgs_adam_plan.restype = ci
        lib.dgs_adam_plan.argtypes = [ci, vp, vp, vp]
        lib.dgs_adam_step.restype = ci
        lib.dgs_adam_step.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]
        lib.dgs_adam_step_pattern.restype = ci
        lib.dgs_adam_step_pattern.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                              vp, vp]
        lib.dgs_adam_step_sched.restype = ci
        lib.dgs_adam_step_sched.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]
        for f in (lib.dgs_mlp_packed_floats, lib.dgs_mlp_saved_floats, lib.dgs_mlp_scratch_floats):
            f.restype = ctypes.c_size_t
        lib.dgs_mlp_packed_floats.argtypes = []
        lib.dgs_mlp_saved_floats.argtypes = [ci]
        lib.dgs_mlp_scratch_floats.argtypes = [ci]
        lib.dgs_mlp_forward.restype = ci
        lib.dgs_mlp_forward.argtypes = [ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp]
        lib.dgs_mlp_forward_select.restype = ci
        lib.dgs_mlp_forward_select.argtypes = [ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, ci, ci, vp, vp]
        lib.dgs_mlp_backward.restype = ci
        lib.dgs_mlp_backward.argtypes = [ci, vp, vp, vp, vp, vp, ci, vp]
        lib.dgs_mlp_backward_reduce.restype = ci
        lib.dgs_mlp_backward_reduce.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp]
        lib.dgs_knn_points2.restype = ci
        lib.dgs_knn_points2.argtypes = [ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp]
        lib.dgs_knn_refine.restype = ci
        lib.dgs_knn_refine.argtypes = [ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp]
        lib.dgs_deform_reduce.restype = ci
        lib.dgs_deform_reduce.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp]
        lib.dgs_knn_refine_mode.restype = ci
        lib.dgs_knn_refine_mode.argtypes = [ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp]
        lib.dgs_deform_forward.restype = ci
        lib.dgs_deform_forward.argtypes = [ci, ci, ci, vp, vp, ci] + [vp] * 14
        lib.dgs_deform_backward.restype = ci
        lib.dgs_deform_backward.argtypes = [ci, ci, ci, vp, vp, ci] + [vp] * 22 + [ci, vp, vp]
        lib.dgs_photo_forward.restype = ci
        lib.dgs_photo_forward.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_photo_backward.restype = ci
        lib.dgs_photo_backward.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp]
        lib.dgs_regloss_backward_slot.restype = ci
        lib.dgs_regloss_backward_slot.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, ci, vp]
        lib.dgs_loss_combine.restype = ci
        lib.dgs_loss_combine.argtypes = [vp, ctypes.c_longlong, vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_float, vp, vp]
        lib.dgs_photo_blocks.restype = ctypes.c_size_t
        lib.dgs_photo_blocks.argtypes = [ci, ci, ci]
        lib.dgs_regloss_blocks.restype = ctypes.c_size_t
        lib.dgs_regloss_blocks.argtypes = [ci, ci]
        lib.dgs_regloss_forward_partials.restype = ci
        lib.dgs_regloss_forward_partials.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.dgs_densify_view.restype = ci
        lib.dgs_densify_view.argtypes = [ci, vp, vp, vp, vp, vp, vp]
        lib.dgs_densify_accumulate.restype = ci
        lib.dgs_densify_accumulate.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_step_guard.restype = ci
        lib.dgs_step_guard.argtypes = [vp, vp, vp, vp, ci, vp, vp]
        lib.dgs_photo_backward_combine.restype = ci
        lib.dgs_photo_backward_combine.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, ctypes.c_longlong, vp, ctypes.c_longlong, vp, vp]
        lib.dgs_photo_backward_combine_guard.restype = ci
        lib.dgs_photo_backward_combine_guard.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, ctypes.c_longlong, vp,
                                                         ctypes.c_longlong, vp, vp, vp, vp, vp, ci, vp]
        lib.dgs_regloss_fused_blocks.restype = ctypes.c_size_t
        lib.dgs_regloss_fused_blocks.argtypes = [ci, ci]
        lib.dgs_regloss_fused.restype = ci
        lib.dgs_regloss_fused.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        lib.dgs_loss_forward_merged.restype = ci
        lib.dgs_loss_forward_merged.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        lib.dgs_regloss_forward_partials_z.restype = ci
        lib.dgs_regloss_forward_partials_z.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        lib.dgs_adam_step_guarded.restype = ci
        lib.dgs_adam_step_guarded.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.dgs_adam_step_zero.restype = ci
        lib.dgs_adam_step_zero.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, ci, vp, vp, vp,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.dgs_select_row.restype = ci
        lib.dgs_select_row.argtypes = [vp, ci, ci, vp, vp, ci, ci, vp, vp]
        lib.dgs_adam_step_origin.restype = ci
        lib.dgs_adam_step_origin.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_float, vp, ci, vp, vp, vp,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.dgs_densify_accumulate_guarded.restype = ci
        lib.dgs_densify_accumulate_guarded.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.dgs_adam_step_sum2.restype = ci
        lib.dgs_adam_step_sum2.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_float, vp, vp, ci, vp, vp, vp,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        if lib.dgs_train_ops_abi_version() != 3:
            raise RuntimeError("libdgs_train_ops.so ABI version mismatch (want 3, library says %d): rebuild it" % lib.dgs_train_ops_abi_version())
        _lib = lib
    return _lib


def _check(lib, rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.dgs_train_ops_last_error().decode()))


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        lib = load()
        a, b = img1.contiguous(), img2.contiguous()
        C, H, W = a.shape[-3:]
        total = torch.zeros(1, dtype=torch.float32, device=a.device)
        need = img1.requires_grad
        maps = torch.empty((3,) + tuple(a.shape), dtype=torch.float32, device=a.device) if need else None
        with torch.cuda.device(a.device):
            rc = lib.dgs_ssim_forward(C, H, W, a.data_ptr(), b.data_ptr(), total.data_ptr(),
                                      maps[0].data_ptr() if need else None, maps[1].data_ptr() if need else None,
                                      maps[2].data_ptr() if need else None, _stream(a.device))
        _check(lib, rc, "dgs_ssim_forward")
        if need:
            ctx.save_for_backward(a, b, maps)
        return (total / float(C * H * W)).reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = load()
        a, b, maps = ctx.saved_tensors
        C, H, W = a.shape[-3:]
        gd = g.reshape(1).to(torch.float32).contiguous()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            rc = lib.dgs_ssim_backward(C, H, W, a.data_ptr(), b.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(),
                                       maps[2].data_ptr(), gd.data_ptr(), out.data_ptr(), _stream(a.device))
        _check(lib, rc, "dgs_ssim_backward")
        return out, None


def fused_ssim(img1, img2):
    """Mean SSIM of two [C,H,W] fp32 HIP tensors; differentiable w.r.t. img1."""
    if img1.dim() != 3 or img1.shape != img2.shape or img1.dtype != torch.float32:
        raise RuntimeError("fused_ssim expects two fp32 [C,H,W] tensors of equal shape")
    return _FusedSSIM.apply(img1, img2.detach())


def knn_indices(x, nodes, K):
    """idx[N,K] (int64) of the K nearest rows of `nodes` for every row of `x` (squared L2, ascending)."""
    lib = load()
    x, nodes = x.detach().contiguous().float(), nodes.detach().contiguous().float()
    N, D = x.shape
    idx = torch.empty((N, K), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dgs_knn_points(N, nodes.shape[0], D, K, x.data_ptr(), nodes.data_ptr(), idx.data_ptr(), None, _stream(x.device))
    _check(lib, rc, "dgs_knn_points")
    return idx


class _FusedLBS(torch.autograd.Function):
    """Control-node blend skinning (dgs_lbs_forward / dgs_lbs_backward)."""

    @staticmethod
    def forward(ctx, x, feature, idx, ntab, attrs, mask, H):
        lib = load()
        x, feature, idx = x.contiguous(), feature.contiguous(), idx.contiguous()
        ntab, attrs, mask = ntab.contiguous(), attrs.contiguous(), mask.contiguous().reshape(-1)
        N, M = x.shape[0], ntab.shape[0]
        d_xyz = torch.empty((N, 3), dtype=torch.float32, device=x.device)
        d_rot = torch.empty((N, 4), dtype=torch.float32, device=x.device)
        d_scale = torch.empty((N, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dgs_lbs_forward(N, M, H, x.data_ptr(), feature.data_ptr(), feature.shape[1], idx.data_ptr(), ntab.data_ptr(),
                                     attrs.data_ptr(), mask.data_ptr(), d_xyz.data_ptr(), d_rot.data_ptr(), d_scale.data_ptr(),
                                     _stream(x.device))
        _check(lib, rc, "dgs_lbs_forward")
        ctx.save_for_backward(x, feature, idx, ntab, attrs, mask)
        ctx.H = H
        return d_xyz, d_rot, d_scale

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scale):
        lib = load()
        x, feature, idx, ntab, attrs, mask = ctx.saved_tensors
        H, N, M = ctx.H, x.shape[0], ntab.shape[0]
        dev = x.device
        z = lambda g, c: torch.zeros((N, c), dtype=torch.float32, device=dev) if g is None else g.contiguous()
        g_xyz, g_rot, g_scale = z(g_xyz, 3), z(g_rot, 4), z(g_scale, 2)
        g_feat = torch.zeros_like(feature)
        g_feat_h = torch.empty((N, H), dtype=torch.float32, device=dev)
        g_ntab = torch.empty_like(ntab)
        g_attrs = torch.empty_like(attrs)
        scratch = torch.empty(int(lib.dgs_lbs_scratch_bytes(M, H)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.dgs_lbs_backward(N, M, H, x.data_ptr(), feature.data_ptr(), feature.shape[1], idx.data_ptr(), ntab.data_ptr(),
                                      attrs.data_ptr(), mask.data_ptr(), g_xyz.data_ptr(), g_rot.data_ptr(), g_scale.data_ptr(),
                                      g_feat_h.data_ptr(), g_ntab.data_ptr(), g_attrs.data_ptr(), scratch.data_ptr(), _stream(dev))
        _check(lib, rc, "dgs_lbs_backward")
        g_feat[:, :H] = g_feat_h
        return None, g_feat, None, g_ntab, g_attrs, None, None


def fused_lbs(x, feature, idx, ntab, attrs, mask, H):
    """(d_xyz, d_rotation, d_scaling) of ControlNodeWarp.forward (local frame, K = 3) from the per-node tables."""
    return _FusedLBS.apply(x, feature, idx, ntab, attrs, mask, H)


def lbs_supported(M, H):
    lib = load()
    lib.dgs_lbs_supported.restype = ctypes.c_int
    lib.dgs_lbs_supported.argtypes = [ctypes.c_int, ctypes.c_int]
    return bool(lib.dgs_lbs_supported(int(M), int(H)))


def select_row(table, counter, override, stride, offset, row_out):
    """dgs_select_row: the view row of this replay, chosen on the device (see include/dgs_train_ops.h)."""
    lib = load()
    dev = table.device
    assert table.dtype == torch.float32 and table.is_contiguous() and row_out.is_contiguous() and counter.dtype == torch.int32 and override.dtype == torch.int32
    with torch.cuda.device(dev):
        _check(lib, lib.dgs_select_row(table.data_ptr(), table.shape[0], table.shape[1], counter.data_ptr(), override.data_ptr(), int(stride), int(offset),
                                       row_out.data_ptr(), _stream(dev)), "dgs_select_row")


class FlatAdam:
    """torch.optim.Adam (betas, eps, no weight decay) over parameters whose .grad tensors are consecutive views of
    one flat fp32 buffer (dgs_amd.train.FlatGradBucket): ONE kernel launch per step instead of one multi-tensor
    launch per parameter group, with the step counter on the device (HIP-graph capturable)."""

    def __init__(self, params, lrs, flat_grad, betas=(0.9, 0.999), eps=1e-15, patterns=None, schedules=None, sched_t0=0.0):
        """patterns: optional {index: (period, split, lr2)}: element i of parameter `index` uses lr2 when
        (i % period) >= split (dgs_adam_step_pattern).
        schedules: optional {index: (lr_final, max_steps)}: exponential decay from lrs[index] to lr_final over max_steps
        steps, evaluated on the device from the step counter (dgs_adam_step_sched)."""
        lib = load()
        assert len(params) == len(lrs) <= 64
        self.params, self.lrs, self.betas, self.eps = list(params), [float(l) for l in lrs], betas, float(eps)
        patterns = patterns or {}
        n_ = len(self.params)
        self._lr2 = (ctypes.c_float * n_)(*[float(patterns[i][2]) if i in patterns else self.lrs[i] for i in range(n_)])
        self._period = (ctypes.c_int * n_)(*[int(patterns[i][0]) if i in patterns else 0 for i in range(n_)])
        self._split = (ctypes.c_int * n_)(*[int(patterns[i][1]) if i in patterns else 0 for i in range(n_)])
        schedules = schedules or {}
        self._lr_final = (ctypes.c_float * n_)(*[float(schedules[i][0]) if i in schedules else self.lrs[i] for i in range(n_)])
        self._sched_steps = (ctypes.c_float * n_)(*[float(schedules[i][1]) if i in schedules else 0.0 for i in range(n_)])
        self.sched_t0 = float(sched_t0)
        # step origin per parameter (dgs_adam_step_origin): the bias corrections of parameter i use t - origin[i]; set_origin()
        self._origin = (ctypes.c_float * n_)(*([0.0] * n_))
        self.grad_scale = 1.0   # gradients are read as grad * grad_scale (1 / world when the bucket holds the sum over ranks)
        self.grad2 = None       # optional second flat gradient buffer of the same layout, added on the fly (dgs_adam_step_sum2)
        # step guard (dgs_step_guard): `skip` = a device int32 that is non-zero when this step must not change anything (a
        # rank's rasterizer overflowed its list capacity); None = every step is applied
        self.skip = None
        self.zero_grads = False  # clear every gradient element behind its read (step + zero_grad in one pass; also on a skipped step)
        self.host_ring = None    # optional pinned float tensor [ring_len, 4] the guard kernel reports into
        self.loss = None         # optional device float: the step's loss, copied into the ring entry by the guard kernel
        dev = flat_grad.device
        n = sum(p.numel() for p in self.params)
        assert flat_grad.numel() >= n and flat_grad.is_contiguous()
        off = [0]
        for p in self.params:
            assert p.is_contiguous() and p.dtype == torch.float32 and p.grad.data_ptr() == flat_grad.data_ptr() + 4 * off[-1]
            off.append(off[-1] + p.numel())
        self.grad = flat_grad
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.status = torch.zeros(3, dtype=torch.float32, device=dev)   # skip flag of the last step, skipped steps, guarded steps
        self._n = len(self.params)
        self._ptrs = (ctypes.c_void_p * self._n)(*[p.data_ptr() for p in self.params])
        self._off = (ctypes.c_longlong * (self._n + 1))(*off)
        self._lr = (ctypes.c_float * self._n)(*self.lrs)
        self._offsets = off
        self._plans = {}
        self._range(0, self._n)

    def _range(self, first, last):
        """ctypes argument slices + block plan of the parameter range [first, last) (built once per range)."""
        key = (first, last)
        if key not in self._plans:
            lib = load()
            dev = self.grad.device
            k = last - first
            off = (ctypes.c_longlong * (k + 1))(*self._offsets[first:last + 1])
            plan = torch.empty(int(lib.dgs_adam_plan_bytes(self._offsets[last] - self._offsets[first])), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _check(lib, lib.dgs_adam_plan(k, off, plan.data_ptr(), _stream(dev)), "dgs_adam_plan")
            sl = lambda arr, typ: (typ * k)(*list(arr)[first:last])
            self._plans[key] = (k, sl(self._ptrs, ctypes.c_void_p), off, sl(self._lr, ctypes.c_float), sl(self._lr2, ctypes.c_float),
                                sl(self._period, ctypes.c_int), sl(self._split, ctypes.c_int), sl(self._lr_final, ctypes.c_float),
                                sl(self._sched_steps, ctypes.c_float), plan)
        return self._plans[key]

    def set_origin(self, first, last, t0):
        """Parameters [first, last) count their Adam steps from t0 (the run's step count when they join the optimisation): what
        torch.optim.Adam's per-parameter step does for a parameter whose .grad was None until then.  Kernel arguments: a captured
        step must be re-captured afterwards."""
        for i in range(first, self._n if last is None else last):
            self._origin[i] = float(t0)
        self.__dict__.pop("_origin_slices", None)

    def moments(self, p):
        """(exp_avg, exp_avg_sq) of parameter p as views shaped like p (state surgery of dgs_amd/densify.py)."""
        for i, q in enumerate(self.params):
            if q is p:
                a, b = self._offsets[i], self._offsets[i + 1]
                return self.exp_avg[a:b].view_as(p), self.exp_avg_sq[a:b].view_as(p)
        return None, None

    @torch.no_grad()
    def guard(self):
        """The step's one-thread guard kernel alone: t += 1 unless the step is to be skipped, bookkeeping for the host.  step()
        launches it when advance=True; a caller that spreads the update over streams launches it first and orders the
        update launches (advance=False) behind it."""
        lib = load()
        dev = self.grad.device
        skip = None if self.skip is None else self.skip.data_ptr()
        ring = self.host_ring
        with torch.cuda.device(dev):
            _check(lib, lib.dgs_step_guard(skip, self.t.data_ptr(), self.status.data_ptr(), None if ring is None else ring.data_ptr(),
                                           0 if ring is None else ring.shape[0], None if self.loss is None else self.loss.data_ptr(),
                                           _stream(dev)), "dgs_step_guard")

    def guard_pointers(self):
        """(skip, step count, status, ring, ring length) for a kernel that runs the guard itself (dgs_photo_backward_combine_guard)."""
        ring = self.host_ring
        return (None if self.skip is None else self.skip.data_ptr(), self.t.data_ptr(), self.status.data_ptr(),
                None if ring is None else ring.data_ptr(), 0 if ring is None else ring.shape[0])

    @torch.no_grad()
    def step(self, first=0, last=None, advance=True):
        """Adam update of parameters [first, last) (default: all).  advance=False reuses the step count of the previous
        call: a step split over several launches advances the counter on its first launch only."""
        lib = load()
        dev = self.grad.device
        last = self._n if last is None else last
        k, ptrs, off, lr, lr2, period, split, lr_final, sched_steps, plan = self._range(first, last)
        okey = (first, last)
        cache = self.__dict__.setdefault("_origin_slices", {})
        origin = cache.get(okey)
        if origin is None:   # (one ctypes array per (first, last), like _range: an eager step makes no Python list of all origins)
            origin = cache[okey] = (ctypes.c_float * k)(*[self._origin[i] for i in range(first, last)])
        skip = None if self.skip is None else self.skip.data_ptr()
        with torch.cuda.device(dev):
            if advance:
                self.guard()
            g2 = self.grad2
            assert g2 is None or (g2.numel() >= self.grad.numel() and g2.is_contiguous() and g2.dtype == torch.float32 and g2.device == dev)
            rc = lib.dgs_adam_step_sum2(k, ptrs, off, lr, lr2, period, split, lr_final, sched_steps, self.sched_t0, origin, float(self.grad_scale),
                                      self.grad.data_ptr(), None if g2 is None else g2.data_ptr(), 1 if self.zero_grads else 0,
                                      self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.t.data_ptr(), self.betas[0],
                                      self.betas[1], self.eps, plan.data_ptr(), skip, _stream(dev))
        _check(lib, rc, "dgs_adam_step")

    @torch.no_grad()
    def step_slice(self, index, lo, hi, advance=True):
        """Adam update of ELEMENTS [lo, hi) of parameter `index` only: the share of a rank that owns these rows of the parameter
        (Trainer.shard_optimizer: reduce-scatter of the gradient -> this -> all-gather of the parameter).  Gradient and moments are
        read at the same flat offsets as a full step; the moments outside [lo, hi) are not touched (and go stale on this rank).
        A periodic learning-rate pattern keeps its phase only if lo is a multiple of the period."""
        lib = load()
        dev = self.grad.device
        n = self._offsets[index + 1] - self._offsets[index]
        assert 0 <= lo <= hi <= n, (lo, hi, n)
        assert self._period[index] == 0 or lo % self._period[index] == 0, "slice start must keep the phase of the learning-rate pattern"
        key = ("slice", index, lo, hi)
        if key not in self._plans:
            off = (ctypes.c_longlong * 2)(self._offsets[index] + lo, self._offsets[index] + hi)
            plan = torch.empty(int(lib.dgs_adam_plan_bytes(hi - lo)), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _check(lib, lib.dgs_adam_plan(1, off, plan.data_ptr(), _stream(dev)), "dgs_adam_plan")
            one = lambda arr, typ: (typ * 1)(arr[index])
            self._plans[key] = ((ctypes.c_void_p * 1)(self.params[index].data_ptr() + 4 * lo), off, one(self._lr, ctypes.c_float),
                                one(self._lr2, ctypes.c_float), one(self._period, ctypes.c_int), one(self._split, ctypes.c_int),
                                one(self._lr_final, ctypes.c_float), one(self._sched_steps, ctypes.c_float), plan)
        ptrs, off, lr, lr2, period, split, lr_final, sched_steps, plan = self._plans[key]
        if hi == lo:
            if advance:
                self.guard()
            return
        origin = (ctypes.c_float * 1)(self._origin[index])
        skip = None if self.skip is None else self.skip.data_ptr()
        with torch.cuda.device(dev):
            if advance:
                self.guard()
            rc = lib.dgs_adam_step_origin(1, ptrs, off, lr, lr2, period, split, lr_final, sched_steps, self.sched_t0, origin, float(self.grad_scale),
                                        self.grad.data_ptr(), 1 if self.zero_grads else 0,
                                        self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.t.data_ptr(), self.betas[0],
                                        self.betas[1], self.eps, plan.data_ptr(), skip, _stream(dev))
        _check(lib, rc, "dgs_adam_step (slice)")


class _FusedRegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, rays_d, rays_o, wvt, lam_n, lam_d):
        lib = load()
        allmap = allmap.contiguous()
        H, W = allmap.shape[1:]
        loss = torch.zeros(1, dtype=torch.float32, device=allmap.device)
        with torch.cuda.device(allmap.device):
            rc = lib.dgs_regloss_forward(H, W, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), lam_n, lam_d,
                                         loss.data_ptr(), _stream(allmap.device))
        _check(lib, rc, "dgs_regloss_forward")
        ctx.save_for_backward(allmap, rays_d, rays_o, wvt)
        ctx.lam = (lam_n, lam_d)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = load()
        allmap, rays_d, rays_o, wvt = ctx.saved_tensors
        H, W = allmap.shape[1:]
        gd = g.reshape(1).to(torch.float32).contiguous()
        out = torch.zeros_like(allmap)
        with torch.cuda.device(allmap.device):
            rc = lib.dgs_regloss_backward(H, W, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), ctx.lam[0],
                                          ctx.lam[1], gd.data_ptr(), out.data_ptr(), _stream(allmap.device))
        _check(lib, rc, "dgs_regloss_backward")
        return out, None, None, None, None, None


def fused_reg_loss(allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist):
    """lambda_normal * mean(1 - <rend_normal, surf_normal>) + lambda_dist * mean(rend_dist) straight from the allmap."""
    return _FusedRegLoss.apply(allmap, rays_d.contiguous(), rays_o.contiguous(), wvt.contiguous(), float(lambda_normal), float(lambda_dist))


def node_mlp_params(net):
    """The 28 (weight, bias) tensors of a dgs_amd.deform.DeformMLP in the order of dgs_mlp_forward, or None when the
    module is not the configuration the kernels are written for."""
    try:
        ok = (net.D == 8 and net.W == 256 and net.multires == 10 and net.t_multires == 6 and net.local_frame
              and net.skips == [4] and net.timenet[2].out_features == 30)
    except AttributeError:
        return None
    if not ok:
        return None
    mods = [net.timenet[0], net.timenet[2]] + list(net.linear) + [net.local_rotation, net.gaussian_warp, net.gaussian_rotation,
                                                                 net.gaussian_scaling]
    out = []
    for m in mods:
        out += [m.weight, m.bias]
    if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in out):
        return None
    return out


def _mlp_forward_raw(x, t, rot_bias, params, select=None):
    """select: None, or the arguments of select_row() -- the weight-packing launch then also picks the view (dgs_mlp_forward_select)."""
    lib = load()
    dev = x.device
    M = x.shape[0]
    if x.stride(-1) != 1 or t.dim() != 2 or t.shape[0] != M:
        raise RuntimeError("fused node MLP: x must be row-major [M,>=3], t [M,1]")
    packed = torch.empty(int(lib.dgs_mlp_packed_floats()), dtype=torch.float32, device=dev)
    saved = torch.empty(int(lib.dgs_mlp_saved_floats(M)), dtype=torch.float32, device=dev)
    attrs = torch.empty((M, 13), dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * 28)(*[p.data_ptr() for p in params])
    rb = (ctypes.c_float * 4)(*rot_bias)
    with torch.cuda.device(dev):
        if select is None:
            rc = lib.dgs_mlp_forward(M, x.data_ptr(), x.stride(0), t.data_ptr(), t.stride(0), ptrs, rb, packed.data_ptr(),
                                     saved.data_ptr(), attrs.data_ptr(), _stream(dev))
        else:
            table, counter, override, stride, offset, row_out = select
            assert (table.dtype == torch.float32 and table.is_contiguous() and row_out.is_contiguous() and counter.dtype == torch.int32
                    and override.dtype == torch.int32 and table.device == dev)
            rc = lib.dgs_mlp_forward_select(M, x.data_ptr(), x.stride(0), t.data_ptr(), t.stride(0), ptrs, rb, packed.data_ptr(),
                                            saved.data_ptr(), attrs.data_ptr(), table.data_ptr(), table.shape[0], table.shape[1],
                                            counter.data_ptr(), override.data_ptr(), int(stride), int(offset), row_out.data_ptr(), _stream(dev))
    _check(lib, rc, "dgs_mlp_forward")
    return attrs, packed, saved


def _mlp_backward_raw(g_attrs, packed, saved, outs, accumulate, fold=None):
    """fold: None, or the arguments of a deferred node-table reduction (the closure's fold_args, see _FusedDeform.backward):
    (M, H, node_radius, node_weight, g_nodes, g_radius, g_weight, g_attrs, flags, table) -- dgs_mlp_backward_reduce then reduces the
    table inside the backward chain's first kernel and WRITES g_attrs."""
    lib = load()
    dev = packed.device
    M = g_attrs.shape[0]
    scratch = torch.empty(int(lib.dgs_mlp_scratch_floats(M)), dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * 28)(*[o.data_ptr() for o in outs])
    with torch.cuda.device(dev):
        if fold is None:
            rc = lib.dgs_mlp_backward(M, g_attrs.data_ptr(), packed.data_ptr(), saved.data_ptr(), scratch.data_ptr(), ptrs,
                                      1 if accumulate else 0, _stream(dev))
        else:
            fM, H, nr, nw, g_nodes, g_rad, g_w, fg, flags, table = fold
            if fM != M or fg.data_ptr() != g_attrs.data_ptr():
                raise RuntimeError("node MLP backward: the deferred reduction belongs to another attribute table")
            rc = lib.dgs_mlp_backward_reduce(M, g_attrs.data_ptr(), packed.data_ptr(), saved.data_ptr(), scratch.data_ptr(), ptrs,
                                             1 if accumulate else 0, H, nr.data_ptr(), nw.data_ptr(), g_nodes.data_ptr(), g_rad.data_ptr(),
                                             g_w.data_ptr(), int(flags), table.data_ptr(), _stream(dev))
    _check(lib, rc, "dgs_mlp_backward")


class _FusedNodeMLP(torch.autograd.Function):
    """attrs[M,13] = [local_rotation + rot_bias | d_xyz | d_rotation | d_scaling] of the control nodes
    (dgs_mlp_forward / dgs_mlp_backward).  `sink`: None, or the list of 28 gradient tensors (the parameters' .grad
    views of a FlatGradBucket) the backward adds into directly instead of returning gradients to autograd."""

    @staticmethod
    def forward(ctx, x, t, rot_bias, sink, *params):
        attrs, packed, saved = _mlp_forward_raw(x, t, rot_bias, params)
        ctx.save_for_backward(packed, saved)
        ctx.sink = sink
        ctx.shapes = [tuple(p.shape) for p in params]
        return attrs

    @staticmethod
    def backward(ctx, g_attrs):
        packed, saved = ctx.saved_tensors
        dev = packed.device
        g_attrs = g_attrs.contiguous()
        if ctx.sink is not None:
            outs, ret = ctx.sink, [None] * 28
        else:
            sizes = [int(torch.Size(s).numel()) for s in ctx.shapes]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            outs, off = [], 0
            for n, shp in zip(sizes, ctx.shapes):
                outs.append(flat[off:off + n].view(shp))
                off += n
            ret = outs
        _mlp_backward_raw(g_attrs, packed, saved, outs, ctx.sink is not None)
        return (None, None, None, None) + tuple(ret)


def fused_node_mlp(net, x, t, rot_bias=(1.0, 0.0, 0.0, 0.0), grad_sink=False):
    """Per-node attribute table [M,13] of DeformMLP `net` at positions x[M,>=3] (no gradient) and times t[M,1].
    grad_sink=True: parameter gradients are ADDED to the existing .grad tensors by the kernel (they must exist, be
    contiguous fp32) and autograd sees no gradient for them -- for trainers that own a flat gradient buffer."""
    params = node_mlp_params(net)
    if params is None or x.shape[0] % 64:
        raise RuntimeError("fused_node_mlp: unsupported DeformMLP configuration")
    sink = None
    if grad_sink and torch.is_grad_enabled():
        sink = [p.grad for p in params]
        if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in sink):
            raise RuntimeError("fused_node_mlp(grad_sink=True): every parameter needs a contiguous fp32 .grad")
    return _FusedNodeMLP.apply(x.detach(), t.detach(), tuple(float(v) for v in rot_bias), sink, *params)


def knn_indices2(x1, x2, nodes, K, seed=None, mode="box"):
    """knn_indices(cat([x1, x2], 1), nodes, K) without materialising the concatenation (x2 may be a column slice).
    seed: an int64 [N,K] tensor holding an earlier answer (e.g. last step's); it is refined IN PLACE to the exact
    current answer and returned (dgs_knn_refine: ~3x cheaper than the plain scan, exact for any seed content).  mode: "box" = 3-D
    culling, "mfma" = dense scores on the matrix cores (dgs_knn_refine_mode; ControlNodes.pick_knn_refine chooses)."""
    lib = load()
    if seed is not None:
        N = x1.shape[0]
        if not (seed.shape == (N, K) and seed.dtype == torch.int64 and seed.is_contiguous() and x1.shape[1] >= 3
                and nodes.shape[0] <= 2048):
            raise RuntimeError("knn_indices2: seed must be a contiguous int64 [N,K] tensor (and <= 2048 nodes)")
        x1d, x2d, nd = x1.detach(), x2.detach(), nodes.detach()
        with torch.cuda.device(x1.device):
            m = {"box": 0, "mfma": 1}[os.environ.get("DGS_KNN_REFINE", mode)]      # (env: development override)
            if m == 1 and (x1d.shape[1] + x2d.shape[1] > 11 or nd.shape[0] > 1024):
                m = 0
            rc = lib.dgs_knn_refine_mode(N, nd.shape[0], x1d.shape[1], x2d.shape[1], K, x1d.data_ptr(), x2d.data_ptr(), x2d.stride(0),
                                         nd.data_ptr(), seed.data_ptr(), m, _stream(x1.device))
        _check(lib, rc, "dgs_knn_refine")
        return seed
    x1, nodes = x1.detach(), nodes.detach()
    x2 = x2.detach()
    if not (x1.is_contiguous() and nodes.is_contiguous() and x2.stride(1) == 1 and x1.dtype == x2.dtype == nodes.dtype == torch.float32):
        raise RuntimeError("knn_indices2: fp32 row-major inputs expected")
    N = x1.shape[0]
    idx = torch.empty((N, K), dtype=torch.int64, device=x1.device)
    with torch.cuda.device(x1.device):
        rc = lib.dgs_knn_points2(N, nodes.shape[0], x1.shape[1], x2.shape[1], K, x1.data_ptr(), x2.data_ptr(), x2.stride(0),
                                 nodes.data_ptr(), idx.data_ptr(), None, _stream(x1.device))
    _check(lib, rc, "dgs_knn_points2")
    return idx


class DeferredNodeMLP:
    """The node MLP outside autograd, for trainers that own the gradient buffers: forward() returns the attribute table
    (no graph), backward(g_attrs) ADDS the 28 parameter gradients into the parameters' .grad tensors whenever the caller
    launches it (e.g. on a side stream, next to the Adam update of the parameters that do not depend on it)."""

    def __init__(self, net):
        self.params = node_mlp_params(net)
        if self.params is None:
            raise RuntimeError("DeferredNodeMLP: unsupported DeformMLP configuration")
        self.state = None

    @torch.no_grad()
    def forward(self, x, t, rot_bias=(1.0, 0.0, 0.0, 0.0), select=None):
        attrs, packed, saved = _mlp_forward_raw(x.detach(), t.detach(), tuple(float(v) for v in rot_bias), self.params, select=select)
        self.state = (packed, saved)
        return attrs

    @torch.no_grad()
    def backward(self, g_attrs, store=False, fold=None):
        """store=True: the parameter gradients are overwritten instead of added to (a buffer that is never cleared).
        fold: a deferred node-table reduction to run inside the first kernel (_mlp_backward_raw)."""
        packed, saved = self.state
        sink = [p.grad for p in self.params]
        if any(g is None or not g.is_contiguous() for g in sink):
            raise RuntimeError("DeferredNodeMLP.backward: every parameter needs a contiguous .grad")
        _mlp_backward_raw(g_attrs, packed, saved, sink, not store, fold=fold)
        self.state = None


class CoherentTables:
    """The persistent zeroed [M][13+H+2] node tables of the coherent skinning backward, owned by whoever runs it (ControlNodes keeps
    one per module: two trainers in one process never add into each other's table).  One table per (device, M, H, stream): two
    backward passes on different streams must not share one; created during the warm-up steps that precede a capture, never inside
    one.  The reduce kernel leaves a table zeroed again; a backward whose deferred reduction never ran (an exception between the
    two) leaves sums behind -- the table is marked dirty for that span and cleared before it is used again."""

    def __init__(self):
        self._tables = {}
        self._dirty = set()

    def get(self, lib, dev, M, H):
        key = (dev, M, H, int(torch.cuda.current_stream(dev).cuda_stream))
        t = self._tables.get(key)
        if t is None:
            t = self._tables[key] = torch.zeros(int(lib.dgs_lbs_scratch_bytes(M, H)), dtype=torch.uint8, device=dev)
        elif key in self._dirty:
            t.zero_()
            self._dirty.discard(key)
        return key, t

    def mark(self, key, dirty):
        (self._dirty.add if dirty else self._dirty.discard)(key)


_DEFAULT_TABLES = CoherentTables()   # for callers of fused_deform that do not bring their own


class _FusedDeform(torch.autograd.Function):
    """(means3D, scales, rotations, opacity) of the deformed surfels from the raw surfel parameters, the node tables
    and the node attribute table (dgs_deform_forward / dgs_deform_backward).  sink: None or the list of the eight
    gradient tensors [xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight] to add into."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight, attrs, idx, mask, H, sink,
                g_attrs_out=None, coherent=False, reduce_later=None, sink_store=False, tables=None, fixed=False):
        lib = load()
        dev = xyz.device
        N, M = xyz.shape[0], nodes.shape[0]
        tens = (xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight)
        if not all(t.is_contiguous() and t.dtype == torch.float32 for t in tens) or nodes.shape[1] != 3 + H:
            raise RuntimeError("fused deform: contiguous fp32 parameters expected, nodes [M, 3 + hyper_dim]")
        attrs, idx = attrs.contiguous(), idx.contiguous()
        mask = None if mask is None else mask.contiguous().reshape(-1)
        # four separate row-major outputs carved from one allocation
        r = lambda n: (n + 63) // 64 * 64  # segment starts stay 256-byte aligned
        o1, o2, o3 = r(3 * N), r(3 * N) + r(2 * N), r(3 * N) + r(2 * N) + r(4 * N)
        buf = torch.empty(o3 + N, dtype=torch.float32, device=dev)
        means3D, scales = buf[:3 * N].view(N, 3), buf[o1:o1 + 2 * N].view(N, 2)
        rots, opac = buf[o2:o2 + 4 * N].view(N, 4), buf[o3:o3 + N].view(N, 1)
        with torch.cuda.device(dev):
            rc = lib.dgs_deform_forward(N, M, H, xyz.data_ptr(), feature.data_ptr(), feature.shape[1], idx.data_ptr(), nodes.data_ptr(),
                                        node_radius.data_ptr(), node_weight.data_ptr(), attrs.data_ptr(),
                                        None if mask is None else mask.data_ptr(), scaling.data_ptr(), rotation.data_ptr(),
                                        opacity.data_ptr(), means3D.data_ptr(), scales.data_ptr(), rots.data_ptr(), opac.data_ptr(),
                                        _stream(dev))
        _check(lib, rc, "dgs_deform_forward")
        ctx.save_for_backward(xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight, attrs, idx)
        ctx.mask, ctx.H, ctx.sink, ctx.g_attrs_out, ctx.coherent = mask, H, sink, g_attrs_out, bool(coherent)
        ctx.reduce_later = reduce_later if (g_attrs_out is not None and sink is not None) else None
        ctx.sink_store = bool(sink_store) and sink is not None and feature.shape[1] == H
        ctx.tables = tables if tables is not None else _DEFAULT_TABLES
        ctx.fixed = bool(fixed) and bool(coherent)   # the coherent table as 64-bit fixed-point sums (order-free integer atomics)
        return means3D, scales, rots, opac

    @staticmethod
    def backward(ctx, g_means, g_scales, g_rots, g_opac):
        lib = load()
        xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight, attrs, idx = ctx.saved_tensors
        dev = xyz.device
        N, M, H = xyz.shape[0], nodes.shape[0], ctx.H
        z = lambda g, ref: torch.zeros_like(ref) if g is None else g.contiguous()
        g_means, g_scales, g_rots, g_opac = z(g_means, xyz), z(g_scales, scaling), z(g_rots, rotation), z(g_opac, opacity)
        g_attrs = torch.empty_like(attrs) if ctx.g_attrs_out is None else ctx.g_attrs_out  # deferred node MLP: caller's buffer
        persistent = 0
        if ctx.coherent:
            tkey, scratch = ctx.tables.get(lib, dev, M, H)   # the caller's persistent zeroed table (CoherentTables)
            persistent = 4
        else:
            scratch = torch.empty(int(lib.dgs_lbs_scratch_bytes(M, H)), dtype=torch.uint8, device=dev)
        tens = (xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight)
        if ctx.sink is not None:
            outs, ret, acc = ctx.sink, [None] * 8, (0 if ctx.sink_store else 1)
        else:
            outs = [torch.empty_like(t) for t in tens]
            if feature.shape[1] > H:
                outs[4].zero_()
            ret, acc = outs, 0
        mask = ctx.mask
        defer = ctx.reduce_later is not None and ctx.coherent and persistent == 4
        flags = acc | (2 if ctx.coherent else 0) | persistent | (8 if defer else 0) | (16 if ctx.fixed else 0)
        with torch.cuda.device(dev):
            rc = lib.dgs_deform_backward(
                N, M, H, xyz.data_ptr(), feature.data_ptr(), feature.shape[1], idx.data_ptr(), nodes.data_ptr(), node_radius.data_ptr(),
                node_weight.data_ptr(), attrs.data_ptr(), None if mask is None else mask.data_ptr(), scaling.data_ptr(),
                rotation.data_ptr(), opacity.data_ptr(), g_means.data_ptr(), g_scales.data_ptr(), g_rots.data_ptr(), g_opac.data_ptr(),
                outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), outs[5].data_ptr(),
                outs[6].data_ptr(), outs[7].data_ptr(), g_attrs.data_ptr(), flags, scratch.data_ptr(), _stream(dev))
        _check(lib, rc, "dgs_deform_backward")
        if defer:
            tables = ctx.tables
            tables.mark(tkey, True)   # sums stay in the table until the deferred reduction has run

            def reduce(M=M, H=H, nr=node_radius, nw=node_weight, outs=outs, g_attrs=g_attrs, flags=flags, scratch=scratch, dev=dev):
                with torch.cuda.device(dev):
                    _check(lib, lib.dgs_deform_reduce(M, H, nr.data_ptr(), nw.data_ptr(), outs[5].data_ptr(), outs[6].data_ptr(), outs[7].data_ptr(),
                                                      g_attrs.data_ptr(), flags & 21, scratch.data_ptr(), _stream(dev)), "dgs_deform_reduce")
                tables.mark(tkey, False)
            # a caller that launches the node MLP's backward next may fold the reduction into it (DeferredNodeMLP.backward(fold=...))
            # and call done() instead of the closure
            reduce.fold_args = (M, H, node_radius, node_weight, outs[5], outs[6], outs[7], g_attrs, flags & 21, scratch)
            reduce.done = lambda: tables.mark(tkey, False)
            ctx.reduce_later.append(reduce)
        return tuple(ret) + (g_attrs if ctx.g_attrs_out is None else None, None, None, None, None, None, None, None, None, None, None)


def fused_deform(xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight, attrs, idx, mask, H, grad_sink=False,
                 g_attrs_out=None, coherent=False, reduce_later=None, tables=None, fixed=False):
    # grad_sink: False, True (ADD into the .grad tensors) or "store" (OVERWRITE them: every element of the eight tensors is written by
    # every backward, so a gradient buffer that only ever receives stores needs no clearing; needs feature.shape[1] == H)
    """Raw surfel parameters + node tables + node attributes -> (means3D, scales, rotations, opacity) for the rasterizer.
    grad_sink=True: gradients of the eight parameters are ADDED to their existing .grad tensors by the kernels.
    coherent=True: the surfels are stored in the order of their nearest node (Trainer.sort_surfels) -- the backward sums per
    wave and issues global atomics instead of building 256 per-workgroup LDS tables (dgs_deform_backward, accumulate bit 1).
    reduce_later: a list (coherent + grad_sink + g_attrs_out only).  The backward then leaves its node table unreduced and appends
    ONE callable to the list; the caller must run it (on any stream ordered behind the backward) before the node gradients or
    g_attrs_out are read -- ControlNodes.finish_backward does, on the node-MLP backward's side stream.
    tables: the CoherentTables that own the persistent node table of the coherent backward (default: a process-wide one).
    fixed (coherent only): the node table as 64-bit fixed-point sums added with integer atomics -- order-free, bit-reproducible."""
    params = (xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight)
    sink = None
    if grad_sink and torch.is_grad_enabled():
        sink = [p.grad for p in params]
        if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in sink):
            raise RuntimeError("fused_deform(grad_sink=True): every parameter needs a contiguous fp32 .grad")
    return _FusedDeform.apply(xyz, scaling, rotation, opacity, feature, nodes, node_radius, node_weight, attrs, idx, mask, H, sink,
                              g_attrs_out, coherent, reduce_later, grad_sink == "store", tables, fixed)


_ONES = {}


def _one(dev):
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    return t


_MERGED_LOSS_FORWARD = os.environ.get("DGS_MERGED_LOSS_FORWARD", "1") != "0"   # 0: the two forward kernels launched one after the other (A/B)


class _FusedTrainLoss(torch.autograd.Function):
    """(1 - l) * L1(image, gt) + l * (1 - SSIM(image, gt)) + lambda_normal * normal consistency + lambda_dist * distortion
    from the rasterizer outputs: 3 launches forward, 2 backward -- or, with unit_grad, 3 launches in the forward and none in the
    backward (see fused_train_loss)."""

    @staticmethod
    def forward(ctx, image, allmap, gt, rays_d, rays_o, wvt, lam_dssim, lam_n, lam_d, slots=None, unit_grad=False, guard=None):
        """slots: None, or an int64 device tensor [2] holding the pointers of the target image and the ray table to use
        (read by the kernels at run time: a captured graph switches views by rewriting them)."""
        lib = load()
        gslot = None if slots is None else ctypes.c_void_p(slots.data_ptr())
        rslot = None if slots is None else ctypes.c_void_p(slots.data_ptr() + 8)
        dev = image.device
        image, allmap, gt = image.contiguous(), allmap.contiguous(), gt.contiguous()
        C, H, W = image.shape
        want_grad = torch.is_grad_enabled() or allmap.requires_grad
        unit = bool(unit_grad) and want_grad
        nb = int(lib.dgs_photo_blocks(C, H, W))
        nr = int(lib.dgs_regloss_fused_blocks(H, W) if unit else lib.dgs_regloss_blocks(H, W))
        part = torch.empty(2 * nb + nr, dtype=torch.float32, device=dev)  # per-workgroup partial sums, no zero fill needed
        maps = torch.empty((3, C, H, W), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        # the backward's gradient image of the allmap.  Two-kernel path: its kernel stores every plane but 5, which collects atomics
        # and is cleared HERE by the regulariser's forward kernel (one fill launch less per step); unit path: stored in full here
        g_allmap = torch.empty_like(allmap) if want_grad else None
        g_image = torch.empty_like(image) if unit else None
        with torch.cuda.device(dev):
            st = _stream(dev)
            if unit and _MERGED_LOSS_FORWARD:   # both forward halves in one launch (ssim 21 + regularisers 17.5 us -> see DESIGN.md section 7)
                _check(lib, lib.dgs_loss_forward_merged(C, H, W, image.data_ptr(), gt.data_ptr(), part.data_ptr(), maps[0].data_ptr(),
                                                        maps[1].data_ptr(), maps[2].data_ptr(), gslot, allmap.data_ptr(), rays_d.data_ptr(),
                                                        rays_o.data_ptr(), wvt.data_ptr(), lam_n, lam_d, part.data_ptr() + 8 * nb,
                                                        g_allmap.data_ptr(), rslot, st), "dgs_loss_forward_merged")
            else:
                _check(lib, lib.dgs_photo_forward(C, H, W, image.data_ptr(), gt.data_ptr(), part.data_ptr(), maps[0].data_ptr(),
                                                  maps[1].data_ptr(), maps[2].data_ptr(), gslot, st), "dgs_photo_forward")
            if unit:
                if not _MERGED_LOSS_FORWARD:
                    _check(lib, lib.dgs_regloss_fused(H, W, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), lam_n, lam_d,
                                                      part.data_ptr() + 8 * nb, g_allmap.data_ptr(), rslot, st), "dgs_regloss_fused")
                gp = (None, None, None, None, 0) if guard is None else guard.guard_pointers()
                _check(lib, lib.dgs_photo_backward_combine_guard(C, H, W, image.data_ptr(), gt.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(),
                                                                 maps[2].data_ptr(), lam_dssim, _one(dev).data_ptr(), g_image.data_ptr(), gslot,
                                                                 part.data_ptr(), nb, part.data_ptr() + 8 * nb, nr, loss.data_ptr(),
                                                                 gp[0], gp[1], gp[2], gp[3], gp[4], st),
                       "dgs_photo_backward_combine")     # its last workgroup sums the partials into the loss (and runs the step guard)
            else:
                _check(lib, lib.dgs_regloss_forward_partials_z(H, W, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(),
                                                               lam_n, lam_d, part.data_ptr() + 8 * nb, rslot,
                                                               None if g_allmap is None else g_allmap[5].data_ptr(), st), "dgs_regloss_forward_partials")
            if not unit:
                _check(lib, lib.dgs_loss_combine(part.data_ptr(), nb, part.data_ptr() + 8 * nb, nr, C * H * W, lam_dssim, loss.data_ptr(), st),
                       "dgs_loss_combine")
        if unit:
            ctx.grads = (g_image, g_allmap)
            return loss.reshape(())
        ctx.grads = None
        ctx.save_for_backward(image, allmap, gt, rays_d, rays_o, wvt, maps)
        ctx.lam = (lam_dssim, lam_n, lam_d)
        ctx.slots = slots
        ctx.g_allmap = g_allmap
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is not None:             # unit_grad: the caller promised g == 1; both gradient images exist already
            g_image, g_allmap = ctx.grads
            ctx.grads = None
            return g_image, g_allmap, None, None, None, None, None, None, None, None, None, None
        lib = load()
        image, allmap, gt, rays_d, rays_o, wvt, maps = ctx.saved_tensors
        gslot = None if ctx.slots is None else ctypes.c_void_p(ctx.slots.data_ptr())
        rslot = None if ctx.slots is None else ctypes.c_void_p(ctx.slots.data_ptr() + 8)
        dev = image.device
        C, H, W = image.shape
        gd = g.reshape(1).to(torch.float32).contiguous()
        g_image = torch.empty_like(image)
        g_allmap = ctx.g_allmap               # plane 5 was cleared in forward; the kernel stores every other plane
        ctx.g_allmap = None
        if g_allmap is None:                  # forward ran without grad mode knowledge (or backward is run twice)
            g_allmap = torch.empty_like(allmap)
            g_allmap[5].zero_()
        with torch.cuda.device(dev):
            st = _stream(dev)
            _check(lib, lib.dgs_photo_backward(C, H, W, image.data_ptr(), gt.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(),
                                               maps[2].data_ptr(), ctx.lam[0], gd.data_ptr(), g_image.data_ptr(), gslot, st), "dgs_photo_backward")
            _check(lib, lib.dgs_regloss_backward_slot(H, W, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(),
                                                      ctx.lam[1], ctx.lam[2], gd.data_ptr(), g_allmap.data_ptr(), rslot, 1, st),
                   "dgs_regloss_backward")
        return g_image, g_allmap, None, None, None, None, None, None, None, None, None, None


def fused_train_loss(image, allmap, gt, rays_d, rays_o, wvt, lambda_dssim, lambda_normal, lambda_dist, slots=None, unit_grad=False, guard=None):
    """unit_grad=True is the caller's PROMISE that backward() will be called with dL/dloss == 1 (Trainer: loss.backward(unit)): the
    gradient images then depend on the forward's inputs only and are produced in the forward -- the regularisers' value and
    gradient in one kernel (dgs_regloss_fused), nothing launched in the backward.  Any other upstream gradient would be ignored.
    guard (unit_grad only): a FlatAdam whose step guard (dgs_step_guard) is run by the thread that writes the loss, with that loss --
    the caller then updates with advance=False."""
    if guard is not None and not (unit_grad and torch.is_grad_enabled()):
        raise RuntimeError("fused_train_loss: the guard rides in the unit-gradient path only")
    return _FusedTrainLoss.apply(image, allmap, gt.detach(), rays_d.contiguous(), rays_o.contiguous(), wvt.contiguous(),
                                 float(lambda_dssim), float(lambda_normal), float(lambda_dist), slots, bool(unit_grad), guard)


def densify_view(radii, g_means2D, grad_norm, visible, radii_vis):
    """Per-view densification statistics into the given [P] output tensors (see dgs_densify_view)."""
    lib = load()
    P = radii.shape[0]
    g = g_means2D.contiguous()
    with torch.cuda.device(radii.device):
        rc = lib.dgs_densify_view(P, radii.data_ptr(), g.data_ptr(), grad_norm.data_ptr(), visible.data_ptr(), radii_vis.data_ptr(),
                                  _stream(radii.device))
    _check(lib, rc, "dgs_densify_view")


def densify_accumulate(grad_norm, visible, radii_vis, accum, denom, max_radii, skip=None):
    """skip: optional device int32; non-zero leaves the running statistics untouched (see dgs_step_guard)."""
    lib = load()
    P = radii_vis.shape[0]
    with torch.cuda.device(radii_vis.device):
        rc = lib.dgs_densify_accumulate_guarded(P, grad_norm.data_ptr(), visible.data_ptr(), radii_vis.data_ptr(), accum.data_ptr(),
                                                denom.data_ptr(), max_radii.data_ptr(), None if skip is None else skip.data_ptr(),
                                                _stream(radii_vis.device))
    _check(lib, rc, "dgs_densify_accumulate")
