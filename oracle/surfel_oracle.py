"""ctypes/numpy front-end of the CPU oracle (oracle/surfel_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Parity unpinned against the reference's own
tests (it has none); pinned by fp64 autograd, see surfel_oracle.c header.

The call shapes follow the reference's extension entry points
(submodules/diff-surfel-rasterization/rasterize_points.cu:39-141, :143-240).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_PTR_FIELDS = {
    "depths": 0, "radii": 1, "means2D": 2, "transMat": 3, "normal_opacity": 4, "rgb": 5, "clamped": 6,
    "tiles_touched": 7, "point_offsets": 8, "point_list": 9, "point_tile": 10, "ranges": 11,
    "final_T": 12, "n_contrib": 13,
}


def build(force=False, verbose=False):
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile).  Rebuilt when the hash of the C source +
    Makefile differs from the one recorded next to the binaries (file times are not trusted)."""
    import hashlib
    h = hashlib.sha256()
    for n in ("surfel_oracle.c", "Makefile"):
        with open(os.path.join(_HERE, n), "rb") as f:
            h.update(f.read())
    want = h.hexdigest()[:16]
    stamp = os.path.join(_HERE, "liboracle.srchash")
    have = None
    if os.path.exists(stamp):
        with open(stamp) as f:
            have = f.read().strip()
    libs_ok = all(os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle_f32.so", "liboracle_f64.so"))
    if force or not libs_ok or have != want:
        if verbose:
            print("[build] compiling oracle (inputs %s)" % want)
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    elif verbose:
        print("[build] reused   oracle (inputs %s)" % want)


def _lib(dtype):
    dtype = np.dtype(dtype)
    key = "f64" if dtype == np.float64 else "f32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "liboracle_%s.so" % key)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        real = ctypes.c_double if key == "f64" else ctypes.c_float
        vp = ctypes.c_void_p
        lib.oracle_forward.restype = vp
        lib.oracle_forward.argtypes = [ctypes.c_int] * 3 + [vp, ctypes.c_int, ctypes.c_int] + [vp] * 9 + [real, real, vp, vp, vp]
        lib.oracle_backward.restype = None
        lib.oracle_backward.argtypes = [vp] * 9 + [real, real] + [vp] * 11
        lib.oracle_free.restype = None
        lib.oracle_free.argtypes = [vp]
        lib.oracle_get_int.restype = ctypes.c_int
        lib.oracle_get_int.argtypes = [vp, ctypes.c_int]
        lib.oracle_get_ptr.restype = vp
        lib.oracle_get_ptr.argtypes = [vp, ctypes.c_int]
        lib.oracle_set_threads.restype = ctypes.c_int
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_pixel_trace.restype = ctypes.c_int
        lib.oracle_pixel_trace.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
        lib.oracle_mark_visible.restype = None
        lib.oracle_mark_visible.argtypes = [ctypes.c_int, vp, vp, vp]
        _LIBS[key] = lib
    return _LIBS[key]


def _arr(x, dtype):
    if x is None:
        return None
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)


class OracleRaster:
    """One forward pass and (optionally) its backward.  Keeps the oracle's state alive like the
    reference keeps geom/binning/img buffers between forward and backward
    (diff_surfel_rasterization/__init__.py:97)."""

    def __init__(self, *, means3D, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy,
                 image_height, image_width, shs=None, colors_precomp=None, sh_degree=0, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.lib = _lib(self.dtype)
        dt = self.dtype
        self.means3D = _arr(means3D, dt).reshape(-1, 3)
        self.P = self.means3D.shape[0]
        self.opacities = _arr(opacities, dt).reshape(-1)
        self.scales = _arr(scales, dt).reshape(-1, 2)
        self.rotations = _arr(rotations, dt).reshape(-1, 4)
        self.shs = _arr(shs, dt)
        self.colors_precomp = _arr(colors_precomp, dt)
        if (self.shs is None) == (self.colors_precomp is None):
            raise ValueError("provide exactly one of shs / colors_precomp")
        self.M = 0 if self.shs is None else int(self.shs.reshape(self.P, -1, 3).shape[1])
        self.D = int(sh_degree)
        self.viewmatrix = _arr(viewmatrix, dt).reshape(16)
        self.projmatrix = _arr(projmatrix, dt).reshape(16)
        self.campos = _arr(campos, dt).reshape(3)
        self.bg = _arr(bg, dt).reshape(3)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        self.H, self.W = int(image_height), int(image_width)
        self.state = None
        self._run_forward()

    def _run_forward(self):
        dt, H, W, P = self.dtype, self.H, self.W, self.P
        self.color = np.zeros((3, H, W), dt)
        self.allmap = np.zeros((8, H, W), dt)
        self.radii = np.zeros((P,), np.int32)
        self.state = self.lib.oracle_forward(
            P, self.D, self.M, _p(self.bg), W, H, _p(self.means3D), _p(self.shs), _p(self.colors_precomp),
            _p(self.opacities), _p(self.scales), _p(self.rotations), _p(self.viewmatrix), _p(self.projmatrix),
            _p(self.campos), self.tanfovx, self.tanfovy, _p(self.color), _p(self.allmap),
            self.radii.ctypes.data_as(ctypes.c_void_p))
        self.num_rendered = self.lib.oracle_get_int(self.state, 0)
        self.tiles_x = self.lib.oracle_get_int(self.state, 1)
        self.tiles_y = self.lib.oracle_get_int(self.state, 2)

    def field(self, name):
        """Copy of an internal state array (for stage-by-stage parity tests)."""
        P, R, N, Tn = self.P, self.num_rendered, self.H * self.W, self.tiles_x * self.tiles_y
        shapes = {
            "depths": ((P,), self.dtype), "radii": ((P,), np.int32), "means2D": ((P, 2), self.dtype),
            "transMat": ((P, 9), self.dtype), "normal_opacity": ((P, 4), self.dtype), "rgb": ((P, 3), self.dtype),
            "clamped": ((P, 3), np.uint8), "tiles_touched": ((P,), np.uint32), "point_offsets": ((P,), np.uint32),
            "point_list": ((R,), np.uint32), "point_tile": ((R,), np.uint32), "ranges": ((Tn, 2), np.uint32),
            "final_T": ((3, self.H, self.W), self.dtype), "n_contrib": ((2, self.H, self.W), np.uint32),
        }
        shape, dt = shapes[name]
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dt)
        ptr = self.lib.oracle_get_ptr(self.state, _PTR_FIELDS[name])
        buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(shape).copy()

    def pixel_trace(self, px, py, cap=4096):
        """Entries pixel (px, py) blends, in list order: (1-based contributor index [n], (depth, weight, T before) [n,3]).  Test aid
        for the discontinuous median channels (oracle_pixel_trace)."""
        contrib = np.zeros(cap, np.uint32)
        vals = np.zeros((cap, 3), self.dtype)
        n = self.lib.oracle_pixel_trace(self.state, int(px), int(py), cap, contrib.ctypes.data_as(ctypes.c_void_p), _p(vals))
        assert 0 <= n <= cap, n
        return contrib[:n], vals[:n]

    def backward(self, dL_dcolor, dL_dallmap):
        """Returns a dict with the 8 gradients of rasterize_points.cu:239 plus dL_dnormal."""
        dt, P = self.dtype, self.P
        gc = _arr(dL_dcolor, dt).reshape(3, self.H, self.W)
        go = _arr(dL_dallmap, dt).reshape(8, self.H, self.W)
        Pn = max(P, 1)
        out = {
            "dL_dmeans2D": np.zeros((Pn, 3), dt), "dL_dnormal": np.zeros((Pn, 3), dt), "dL_dopacity": np.zeros((Pn, 1), dt),
            "dL_dcolors": np.zeros((Pn, 3), dt), "dL_dmeans3D": np.zeros((Pn, 3), dt), "dL_dtransMat": np.zeros((Pn, 9), dt),
            "dL_dsh": np.zeros((Pn, max(self.M, 1), 3), dt), "dL_dscales": np.zeros((Pn, 2), dt), "dL_drotations": np.zeros((Pn, 4), dt),
        }
        self.lib.oracle_backward(
            self.state, _p(self.bg), _p(self.means3D), _p(self.shs), _p(self.colors_precomp), _p(self.scales),
            _p(self.rotations), _p(self.viewmatrix), _p(self.campos), self.tanfovx, self.tanfovy, _p(gc), _p(go),
            _p(out["dL_dmeans2D"]), _p(out["dL_dnormal"]), _p(out["dL_dopacity"]), _p(out["dL_dcolors"]),
            _p(out["dL_dmeans3D"]), _p(out["dL_dtransMat"]), _p(out["dL_dsh"]), _p(out["dL_dscales"]), _p(out["dL_drotations"]))
        out = {k: v[:P] for k, v in out.items()}
        out["dL_dsh"] = out["dL_dsh"][:, :self.M]
        return out

    def close(self):
        if self.state is not None:
            self.lib.oracle_free(self.state)
            self.state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mark_visible(means3D, viewmatrix, dtype=np.float32):
    lib = _lib(dtype)
    m = _arr(means3D, dtype).reshape(-1, 3)
    v = _arr(viewmatrix, dtype).reshape(16)
    out = np.zeros((m.shape[0],), np.uint8)
    if m.shape[0]:
        lib.oracle_mark_visible(m.shape[0], _p(m), _p(v), out.ctypes.data_as(ctypes.c_void_p))
    return out.astype(bool)


def set_threads(n, dtype=np.float32):
    """Sets the OpenMP thread count of the oracle's tile loops; returns the count in effect."""
    return int(_lib(dtype).oracle_set_threads(int(n)))
