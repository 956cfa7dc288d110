"""The two C-ABI libraries load and export every function their headers declare (no compute: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynamic-2dgs_amd", "csrc")
CASES = [("dgs_surfel_rasterizer.h", "libdgs_surfel_rasterizer.so"), ("dgs_train_ops.h", "libdgs_train_ops.so")]


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    # prototypes: "<type> name(args);" -- skip typedef'd function pointers "(*name)"
    return sorted(set(re.findall(r"\b(dgs_[a-z0-9_]+)\s*\(", text)) - set(re.findall(r"\(\s*\*\s*(dgs_[a-z0-9_]+)\s*\)", text)))


@pytest.mark.parametrize("header,lib", CASES)
def test_library_exports_every_declared_symbol(header, lib):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
    from dgs_amd import _ops
    from diff_surfel_rasterization import _C
    path = (_C if "rasterizer" in lib else _ops).build()  # rebuilds only when a source is newer than the library
    assert os.path.basename(path) == lib
    names = declared_functions(header)
    assert len(names) >= 5, names
    handle = ctypes.CDLL(path)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, "%s does not export %s" % (lib, missing)


def test_python_bindings_list_matches_headers():
    """The symbol lists __graft_entry__.build() checks are the headers' declarations."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
    from dgs_amd import _ops
    from diff_surfel_rasterization import _C
    assert sorted(_ops.exported_symbols()) == declared_functions("dgs_train_ops.h")
    assert set(declared_functions("dgs_surfel_rasterizer.h")) <= set(_C.exported_symbols()) | {"dgs_alloc_fn"}
