"""Dev tool (GPU box): which of two things makes test_stored_gradients_equal_cleared_and_added_ones flaky -- the store mode or the
position of a run in the process?  Runs trainers in the given order of modes (1 = store, 0 = cleared) and prints the pairwise maximum
parameter differences after three steps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

import bench

dev = torch.device("cuda:0")
order = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1000")]
runs = []
for store in order:
    tr = bench.build_trainer(20000, 256, 256, dev, n_views=4, n_targets=2)
    tr.store_grads = bool(store)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    runs.append([p.detach().clone() for p in tr.bucket.params[:7]])
    names = [tuple(p.shape) for p in tr.bucket.params[:7]]
    del tr
for k, nm in enumerate(names):
    print("param %d %s" % (k, nm))
    for i in range(len(runs)):
        print("  run %d (%s):" % (i, "store" if order[i] else "clear"), " ".join("%.2e" % float((runs[i][k] - runs[j][k]).abs().max()) for j in range(len(runs))))
