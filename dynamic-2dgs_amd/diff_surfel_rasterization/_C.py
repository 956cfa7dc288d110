"""Binding of libdgs_surfel_rasterizer.so (include/dgs_surfel_rasterizer.h) for PyTorch-ROCm tensors.

Plays the role of the reference's pybind module ``diff_surfel_rasterization._C``
(submodules/diff-surfel-rasterization/ext.cpp:15-19) with the same three functions, argument order
and return tuples as rasterize_points.cu:39-141 / :143-240 / :242-261, but goes through the C ABI
with ctypes: torch only owns the memory and supplies the current HIP stream.

There is NO CPU path here: if the HIP library is missing, or a tensor is not on a HIP device, the
call fails loudly.
"""
import ctypes
import os
import torch

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_NAME = "libdgs_surfel_rasterizer.so"
LIB_PATH = os.environ.get("DGS_SURFEL_LIB", os.path.join(_CSRC, LIB_NAME))  # override: development A/B builds only
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent fp32 ops into v_pk_*_f32, which on gfx950 cost twice the cycles
# of the scalar forms (no throughput gain) plus v_mov shuffles to pair the operands; measured -17 % / -19 % on the
# forward / backward blend kernels without it.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
_SOURCES = ["surfel_rasterizer.hip", "kernels_blend.h", "kernels_preprocess.h", "surfel_math.h", "wave_reduce.h"]

_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_lib = None

_EXPORTS = ("dgs_abi_version", "dgs_last_error", "dgs_rasterizer_mark_visible", "dgs_rasterizer_forward",
            "dgs_rasterizer_backward", "dgs_debug_layout", "dgs_set_tight_rects", "dgs_set_option", "dgs_read_overflow", "dgs_profile_enable", "dgs_profile_reset",
            "dgs_profile_read", "dgs_set_overflow_flag", "dgs_context_create", "dgs_context_destroy", "dgs_context_set_option",
            "dgs_context_set_overflow_flag", "dgs_context_read_overflow", "dgs_context_profile_enable", "dgs_context_profile_reset",
            "dgs_context_profile_read", "dgs_context_forward", "dgs_context_backward", "dgs_get_option", "dgs_context_get_option")


def _deps():
    hdr = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include", "dgs_surfel_rasterizer.h")
    return [os.path.join(_CSRC, s) for s in _SOURCES] + [hdr]


def source_hash():
    """Hash of the sources + flags the library is built from (recorded next to the .so and with the PMC profiles)."""
    import _dgs_build
    return _dgs_build.source_hash(_deps(), HIPCC_FLAGS)


def build(force=False, verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).  Rebuilds whenever the hash of
    sources + flags differs from the one recorded with the existing binary."""
    import _dgs_build
    cmd = ["hipcc"] + HIPCC_FLAGS + [os.path.join(_CSRC, "surfel_rasterizer.hip"), "-o", LIB_PATH]
    return _dgs_build.build(LIB_PATH, cmd, _deps(), HIPCC_FLAGS, _CSRC, force=force, verbose=verbose)[0]


PRECISE_LIB_PATH = os.path.join(_CSRC, "libdgs_surfel_rasterizer_precise.so")


def build_precise(force=False, verbose=False):
    """Test-only twin of the library with -DDGS_PRECISE_MATH: IEEE division and expf() instead of v_rcp_f32 / v_exp_f32 in
    the per-pixel arithmetic (surfel_math.h fast_rcp / fast_exp).  tests/test_precise_math_gpu.py loads it in a subprocess
    (DGS_SURFEL_LIB) to show how much of the distance to the oracle the fast intrinsics account for; never the product."""
    import _dgs_build
    flags = HIPCC_FLAGS + ["-DDGS_PRECISE_MATH"]
    cmd = ["hipcc"] + flags + [os.path.join(_CSRC, "surfel_rasterizer.hip"), "-o", PRECISE_LIB_PATH]
    return _dgs_build.build(PRECISE_LIB_PATH, cmd, _deps(), flags, _CSRC, force=force, verbose=verbose)[0]


def _refuse_stale(lib_path, hash_fn, build_fn):
    """A binary built from other sources than the ones in the tree must never be what the tests or the bench measure: if the
    hash recorded next to it differs from the hash of the current sources + flags, rebuild; if that is impossible, fail."""
    import _dgs_build
    if os.path.basename(lib_path) not in (LIB_NAME, "libdgs_train_ops.so") or os.path.dirname(os.path.abspath(lib_path)) != os.path.abspath(_CSRC):
        return   # DGS_SURFEL_LIB override: an explicit A/B or twin build (its own flags), the caller's responsibility
    have, want = _dgs_build.recorded_hash(lib_path), hash_fn()
    if have == want:
        return
    try:
        build_fn()
    except Exception as ex:
        raise RuntimeError("%s was built from other sources (recorded inputs %s, tree %s) and cannot be rebuilt here: %r"
                           % (lib_path, have, want, ex))


def load():
    """dlopen the library and declare the prototypes of include/dgs_surfel_rasterizer.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s not found: the MI355X surfel rasterizer has no CPU fallback. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)." % LIB_PATH)
    _refuse_stale(LIB_PATH, source_hash, build)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    lib.dgs_abi_version.restype = ci
    lib.dgs_abi_version.argtypes = []
    lib.dgs_last_error.restype = ctypes.c_char_p
    lib.dgs_last_error.argtypes = []
    lib.dgs_rasterizer_mark_visible.restype = ci
    lib.dgs_rasterizer_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
    fwd_args = [_ALLOC_FN, vp, _ALLOC_FN, vp, _ALLOC_FN, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp,
                cf, cf, ci, vp, vp, vp, ci, vp]
    bwd_args = [ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                vp, vp, vp, vp, vp, ci, vp]
    lib.dgs_rasterizer_forward.restype = ci
    lib.dgs_rasterizer_forward.argtypes = fwd_args
    lib.dgs_rasterizer_backward.restype = ci
    lib.dgs_rasterizer_backward.argtypes = bwd_args
    lib.dgs_context_forward.restype = ci
    lib.dgs_context_forward.argtypes = [vp] + fwd_args
    lib.dgs_context_backward.restype = ci
    lib.dgs_context_backward.argtypes = [vp] + bwd_args
    lib.dgs_context_create.restype = vp
    lib.dgs_context_create.argtypes = []
    lib.dgs_context_destroy.restype = None
    lib.dgs_context_destroy.argtypes = [vp]
    lib.dgs_context_set_option.restype = ci
    lib.dgs_context_set_option.argtypes = [vp, ci, ci]
    lib.dgs_context_get_option.restype = ci
    lib.dgs_context_get_option.argtypes = [vp, ci]
    lib.dgs_get_option.restype = ci
    lib.dgs_get_option.argtypes = [ci]
    lib.dgs_context_set_overflow_flag.restype = ci
    lib.dgs_context_set_overflow_flag.argtypes = [vp, vp]
    lib.dgs_context_read_overflow.restype = ci
    lib.dgs_context_read_overflow.argtypes = [vp, ci]
    lib.dgs_context_profile_enable.restype = ci
    lib.dgs_context_profile_enable.argtypes = [vp, ci]
    lib.dgs_context_profile_reset.restype = None
    lib.dgs_context_profile_reset.argtypes = [vp]
    lib.dgs_context_profile_read.restype = ci
    lib.dgs_context_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ci]
    lib.dgs_set_overflow_flag.restype = ci
    lib.dgs_set_overflow_flag.argtypes = [vp]
    lib.dgs_debug_layout.restype = ci
    lib.dgs_debug_layout.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(sz), ci]
    lib.dgs_set_tight_rects.restype = None
    lib.dgs_set_tight_rects.argtypes = [ci]
    lib.dgs_set_option.restype = ci
    lib.dgs_set_option.argtypes = [ci, ci]
    lib.dgs_read_overflow.restype = ci
    lib.dgs_read_overflow.argtypes = [ci]
    lib.dgs_profile_enable.restype = None
    lib.dgs_profile_enable.argtypes = [ci]
    lib.dgs_profile_reset.restype = None
    lib.dgs_profile_reset.argtypes = []
    lib.dgs_profile_read.restype = ci
    lib.dgs_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ci]
    if lib.dgs_abi_version() != 3:
        raise RuntimeError("libdgs_surfel_rasterizer.so ABI version mismatch")
    _lib = lib
    return lib


def exported_symbols():
    return _EXPORTS


def _raise(lib, code, where):
    msg = lib.dgs_last_error().decode("utf-8", "replace")
    raise RuntimeError("%s failed (status %d): %s" % (where, code, msg))


def _need_device(**tensors):
    for name, t in tensors.items():
        if not t.is_cuda:  # mirrors CHECK_INPUT, rasterize_points.cu:27-28
            raise RuntimeError("%s must be a CUDA (HIP) tensor" % name)


def _ptr(t):
    """Device pointer or NULL for an empty tensor (the reference's nullptr convention)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _f32c(t):
    if t.dtype != torch.float32:
        raise RuntimeError("expected float32 tensor, got %s" % t.dtype)
    return t.contiguous()


class _Resizer:
    """resizeFunctional of rasterize_points.cu:31-37 as a C callback."""

    def __init__(self, device):
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _ALLOC_FN(self._alloc)

    def _alloc(self, _ctx, nbytes):
        try:
            self.tensor.resize_(int(nbytes))
            return self.tensor.data_ptr()
        except Exception:  # surfaces as DGS_ERR_ALLOC
            return 0


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context:
    """An explicit library context (dgs_context_create): its own options, overflow flag, staging word and timing state.
    The operator surface uses the per-device default context; a caller that drives one device from several threads, or wants
    different options side by side, passes `context=` to rasterize_gaussians / rasterize_gaussians_backward."""

    def __init__(self, device=None):
        lib = load()
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            self.handle = ctypes.c_void_p(lib.dgs_context_create())
        if not self.handle:
            raise RuntimeError("dgs_context_create failed")

    def set_option(self, key, value):
        lib = load()
        rc = lib.dgs_context_set_option(self.handle, int(key), int(value))
        if rc < 0:
            _raise(lib, rc, "context.set_option")

    def get_option(self, key):
        lib = load()
        v = lib.dgs_context_get_option(self.handle, int(key))
        if v < 0:
            _raise(lib, v, "context.get_option")
        return int(v)

    def set_overflow_flag(self, tensor):
        lib = load()
        rc = lib.dgs_context_set_overflow_flag(self.handle, None if tensor is None else tensor.data_ptr())
        if rc < 0:
            _raise(lib, rc, "context.set_overflow_flag")
        self._flag = tensor   # keep it alive

    def read_overflow(self, reset=True):
        return int(load().dgs_context_read_overflow(self.handle, 1 if reset else 0))   # 0, or the reason bits (dgs_set_overflow_flag)

    def close(self):
        if self.handle:
            load().dgs_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, context=None):
    """-> (num_rendered, out_color[3,H,W], out_others[8,H,W], radii[P] i32, geomBuffer, binningBuffer, imgBuffer)"""
    lib = load()
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if scales.dim() != 2 or scales.size(1) != 2:
        raise RuntimeError("scales must have dimensions (num_points, 2)")
    if rotations.dim() != 2 or rotations.size(1) != 4:
        raise RuntimeError("rotations must have dimensions (num_points, 4)")
    _need_device(background=background, means3D=means3D, colors=colors, opacity=opacity, scales=scales, rotations=rotations,
                 transMat_precomp=transMat_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, sh=sh, campos=campos)
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    # P > 0: the preprocess kernel writes every radius and the forward blend every pixel of all eleven planes (background
    # included), so the buffers need no fill; P == 0 returns zeros like the reference (rasterize_points.cu:60-106)
    alloc = torch.empty if P != 0 else torch.zeros
    planes = alloc((11, H, W), dtype=torch.float32, device=dev)
    out_color, out_others = planes[:3], planes[3:]
    radii = alloc((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Resizer(dev), _Resizer(dev), _Resizer(dev)
    rendered = 0
    if P != 0:
        M = sh.size(1) if sh.numel() != 0 else 0
        bg, m3, col, opa = _f32c(background), _f32c(means3D), _f32c(colors), _f32c(opacity)
        sc, rot, tm = _f32c(scales), _f32c(rotations), _f32c(transMat_precomp)
        vm, pm, shc, cp = _f32c(viewmatrix), _f32c(projmatrix), _f32c(sh), _f32c(campos)
        with torch.cuda.device(dev):
            entry = lib.dgs_rasterizer_forward if context is None else (lambda *a: lib.dgs_context_forward(context.handle, *a))
            rendered = entry(
                geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), int(M), _ptr(bg), W, H, _ptr(m3), _ptr(shc),
                _ptr(col), _ptr(opa), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm), _ptr(cp),
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), out_color.data_ptr(), out_others.data_ptr(),
                radii.data_ptr(), int(bool(debug)), _stream(dev))
        if rendered < 0:
            _raise(lib, rendered, "rasterize_gaussians")
    return rendered, out_color, out_others, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, transMat_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_others, sh, degree,
                                 campos, geomBuffer, R, binningBuffer, imageBuffer, debug, dL_dsh_out=None, context=None,
                                 want_colors=True, want_transmat=True):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dtransMat[P,9], dL_dsh[P,M,3],
    dL_dscales[P,2], dL_drotations[P,4])  -- rasterize_points.cu:239.
    want_colors / want_transmat = False: that array is not computed (None in its place): the autograd node passes what its inputs'
    requires_grad says (dL_dcolors is the gradient of colors_precomp, dL_dtransMat of cov3Ds_precomp)."""
    lib = load()
    _need_device(background=background, means3D=means3D, radii=radii, colors=colors, scales=scales, rotations=rotations,
                 transMat_precomp=transMat_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, sh=sh, campos=campos,
                 binningBuffer=binningBuffer, imageBuffer=imageBuffer, geomBuffer=geomBuffer)
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    opts = dict(dtype=torch.float32, device=dev)
    # the eight per-surfel arrays from ONE allocation, not filled: the library writes every row of them (zeros for culled
    # surfels; dgs_surfel_rasterizer.h).  dL_dsh follows the reference (visible rows only): zero-filled here unless the caller
    # provides the array it accumulates into
    cols = (3, 3, 3, 3, 1, 9, 2, 4)
    seg = [(P * c + 63) // 64 * 64 for c in cols]  # segment starts stay 256-byte aligned
    flat = torch.empty(sum(seg), **opts) if P != 0 else torch.zeros(sum(seg), **opts)
    views, off = [], 0
    for c, n in zip(cols, seg):
        views.append(flat[off:off + P * c].view(P, c))
        off += n
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dnormal, dL_dopacity, dL_dtransMat, dL_dscales, dL_drotations = views
    full = want_colors and want_transmat   # every array of the reference's call, its scratch (dL_dnormal) included
    dL_dsh = dL_dsh_out if dL_dsh_out is not None else torch.zeros((P, M, 3), **opts)  # caller-provided: written in place (extension)
    if P != 0:
        bg, m3, col = _f32c(background), _f32c(means3D), _f32c(colors)
        sc, rot, tm = _f32c(scales), _f32c(rotations), _f32c(transMat_precomp)
        vm, pm, shc, cp = _f32c(viewmatrix), _f32c(projmatrix), _f32c(sh), _f32c(campos)
        gc, go = _f32c(dL_dout_color), _f32c(dL_dout_others)
        rad = radii.contiguous()
        gb, bb, ib = geomBuffer.contiguous(), binningBuffer.contiguous(), imageBuffer.contiguous()
        with torch.cuda.device(dev):
            entry = lib.dgs_rasterizer_backward if context is None else (lambda *a: lib.dgs_context_backward(context.handle, *a))
            rc = entry(
                P, int(degree), int(M), int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(sc), float(scale_modifier),
                _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx), float(tan_fovy), _ptr(rad), _ptr(gb),
                _ptr(bb), _ptr(ib), _ptr(gc), _ptr(go), dL_dmeans2D.data_ptr(), dL_dnormal.data_ptr() if full else None, dL_dopacity.data_ptr(),
                dL_dcolors.data_ptr() if want_colors else None, dL_dmeans3D.data_ptr(), dL_dtransMat.data_ptr() if want_transmat else None,
                _ptr(dL_dsh), dL_dscales.data_ptr(),
                dL_drotations.data_ptr(), int(bool(debug)), _stream(dev))
        if rc < 0:
            _raise(lib, rc, "rasterize_gaussians_backward")
    return (dL_dmeans2D, dL_dcolors if want_colors else None, dL_dopacity, dL_dmeans3D, dL_dtransMat if want_transmat else None, dL_dsh,
            dL_dscales, dL_drotations)


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P], view-space z > 0.2 (rasterize_points.cu:242-261)."""
    lib = load()
    _need_device(means3D=means3D, viewmatrix=viewmatrix, projmatrix=projmatrix)
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m3, vm, pm = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        with torch.cuda.device(means3D.device):
            rc = lib.dgs_rasterizer_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), present.data_ptr(), _stream(means3D.device))
        if rc < 0:
            _raise(lib, rc, "mark_visible")
    return present


# ---- introspection for tests / bench (not part of the reference surface) -----------------------------

def debug_layout(which, P=0, width=1, height=1, R=0):
    lib = load()
    buf = (ctypes.c_size_t * 8)()
    n = lib.dgs_debug_layout(int(which), int(P), int(width), int(height), int(R), buf, 8)
    if n < 0:
        _raise(lib, n, "debug_layout")
    return [int(buf[i]) for i in range(n)]


import contextlib


@contextlib.contextmanager
def _on(device):
    """The reference-shaped C entry points act on the DEFAULT CONTEXT OF THE CALLING THREAD'S CURRENT DEVICE.  Every wrapper below
    takes `device` (a torch device, an index, or a tensor's .device) and makes it current around the call, so that a trainer on
    cuda:1 configures cuda:1's context whatever the caller's current device is; None keeps the current device."""
    if device is None:
        yield
    else:
        with torch.cuda.device(device):
            yield


def set_tight_rects(on=True, device=None):
    """Tile-list policy (see dgs_set_tight_rects): False reproduces the reference's lists entry for entry."""
    with _on(device):
        load().dgs_set_tight_rects(1 if on else 0)


def set_option(key, value, device=None):
    lib = load()
    with _on(device):
        rc = lib.dgs_set_option(int(key), int(value))
    if rc < 0:
        _raise(lib, rc, "set_option")


def get_option(key, device=None):
    """Current value of an option of the device's default context (dgs_get_option)."""
    lib = load()
    with _on(device):
        v = lib.dgs_get_option(int(key))
    if v < 0:
        _raise(lib, v, "get_option")
    return int(v)


def set_capacity(n_entries, device=None):
    """Capacity mode (dgs_set_option key 2): > 0 makes forward/backward free of host synchronisation, 0 restores it."""
    set_option(2, int(n_entries), device)


def read_overflow(reset=True, device=None):
    with _on(device):
        return int(load().dgs_read_overflow(1 if reset else 0))   # 0, or the reason bits (dgs_set_overflow_flag)


_OVERFLOW_FLAGS = {}   # device index -> the caller-owned flag tensor its default context points at (kept alive here)


def set_overflow_flag(tensor, device=None):
    """Hand the default context of the tensor's device a caller-owned int32[1] overflow flag (dgs_set_overflow_flag), so that
    the caller's own kernels can read it on the device; None (+ `device`) returns that device's context to the library-owned flag."""
    lib = load()
    if tensor is not None:
        assert tensor.dtype == torch.int32 and tensor.numel() == 1 and tensor.is_cuda
        device = tensor.device
        with torch.cuda.device(device):
            rc = lib.dgs_set_overflow_flag(tensor.data_ptr())
    else:
        with _on(device):
            rc = lib.dgs_set_overflow_flag(None)
    if rc < 0:
        _raise(lib, rc, "set_overflow_flag")
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if tensor is None:
        _OVERFLOW_FLAGS.pop(idx, None)
    else:
        _OVERFLOW_FLAGS[idx] = tensor   # keep it alive while the library points at it


def profile_enable(mode=True, device=None):
    """0 / False: off; 1 / True: HIP events around the timed kernels (eager launches); 2: device timestamps, legal inside a
    captured graph (see dgs_profile_enable)."""
    with _on(device):
        load().dgs_profile_enable(int(mode))


def profile_reset(device=None):
    with _on(device):
        load().dgs_profile_reset()


def profile_read(device=None):
    """Accumulated over the timed launches: blend kernels {'fwd_ms', 'fwd_n', 'bwd_ms', 'bwd_n', 'fwd_S', 'bwd_S'}, and
    {'pre_ms', 'pre_n'} preprocess_fwd, {'bin_ms', 'bin_n'} binning (count + scan + scatter + per-tile sort; kernel time only in capacity mode: the exact-size path has a host read inside the span), {'sbw_ms', 'sbw_n'}
    surfel_bwd, 'R' = sum of num_rendered, 'Pv' = sum of visible surfels over the timed forwards."""
    out = (ctypes.c_double * 14)()
    with _on(device):
        load().dgs_profile_read(out, 14)
    return {"fwd_ms": out[0], "fwd_n": int(out[1]), "bwd_ms": out[2], "bwd_n": int(out[3]), "fwd_S": out[4], "bwd_S": out[5],
            "pre_ms": out[6], "pre_n": int(out[7]), "bin_ms": out[8], "bin_n": int(out[9]), "sbw_ms": out[10], "sbw_n": int(out[11]),
            "R": out[12], "Pv": out[13]}
