"""ctypes binding of libdgs_train_ops.so (include/dgs_train_ops.h): fused SSIM and brute-force KNN kernels for
gfx950, used by dgs_amd.losses / dgs_amd.deform when their inputs live on a HIP device.  CPU tensors keep using
the PyTorch formulation (that is what the CPU tests and bench.py's cpu_baseline leg run)."""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.path.join(_CSRC, "libdgs_train_ops.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize"]
_lib = None
_EXPORTS = ("dgs_train_ops_abi_version", "dgs_train_ops_last_error", "dgs_ssim_forward", "dgs_ssim_backward", "dgs_knn_points")


def build(force=False, verbose=False):
    src = os.path.join(_CSRC, "train_ops.hip")
    hdr = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include", "dgs_train_ops.h")
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return LIB_PATH
    cmd = ["hipcc"] + HIPCC_FLAGS + [src, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    return LIB_PATH


def exported_symbols():
    return _EXPORTS


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found; build it with __graft_entry__.build()" % LIB_PATH)
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
        if lib.dgs_train_ops_abi_version() != 1:
            raise RuntimeError("libdgs_train_ops.so ABI version mismatch")
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
