"""What the backward blend's visits are made of: needs the counting build of the library --
    tools/ab_variants.sh build count:-DDGS_COUNT_VISITS   (here)      DGS_SURFEL_LIB=.../csrc/ab_count.so python tools/diag/visit_counts.py   (GPU box)
Prints, for one backward of one view of the workload: staged entries, visits (wave x staged entry that passes the quadrant's footprint
test), visits in which NO lane passes the alpha test (they cost ~28 VALU instructions instead of ~120), lanes blending per full visit."""
import ctypes
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)
import torch
import bench
from diff_surfel_rasterization import _C

lib = _C.load()
lib.dgs_debug_visit_counts.restype = ctypes.c_int
lib.dgs_debug_visit_counts.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS[os.environ.get("WORKLOAD", "metric")]
tr = bench.build_trainer(P, H, W, dev)
tr.opt_surfels.step = lambda *a, **k: None     # the scene stays the initial one
for v in (0, 9, 23):
    tr.view_for = lambda it, j=0, v=v: v
    torch.cuda.synchronize()
    lib.dgs_debug_visit_counts(None, 1)
    tr.step()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 4)()
    lib.dgs_debug_visit_counts(out, 0)
    visits, empty, lanes, staged = (int(x) for x in out)
    print("view %2d: staged (wave, entry) pairs %d, visits %d (%.3f of staged), empty visits %d (%.3f of visits), lanes blending per full visit %.1f"
          % (v, staged, visits, visits / max(staged, 1), empty, empty / max(visits, 1), lanes / max(visits - empty, 1)))
