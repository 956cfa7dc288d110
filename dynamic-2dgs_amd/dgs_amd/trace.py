"""Optional roctx ranges around the stages of the train step (the reference has no tracing; its stages are
what a profile of it would be read by).  Off unless DGS_ROCTX=1: `rocprofv3 --marker-trace -- python bench.py` (with DGS_NO_GRAPHS=1:
ranges are host-side markers, a replayed graph has none) then shows deform / rasterize+loss / backward / update per step next to
the kernel trace.  No-ops when the variable is unset or libroctx64.so is not there."""
import contextlib
import ctypes
import os

_lib = None
_on = os.environ.get("DGS_ROCTX", "0") == "1"


def _load():
    global _lib, _on
    if _lib is None and _on:
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"):   # rocprofv3 traces the SDK library
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _lib.roctxRangePushA.restype = ctypes.c_int
                _lib.roctxRangePop.restype = ctypes.c_int
                break
            except (OSError, AttributeError):
                _lib = None
        if _lib is None:
            _on = False
    return _lib


def enabled():
    return _on and _load() is not None


@contextlib.contextmanager
def stage(name):
    """with trace.stage("backward"): ...   -- a roctx range when tracing is on, nothing otherwise."""
    lib = _load() if _on else None
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
