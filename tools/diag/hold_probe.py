"""Development probe: fit() eager vs captured, both deterministic, across densifying iterations (the surfels are held there):
the two must agree bit for bit.  Usage (GPU box): python tools/diag/hold_probe.py"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
from dgs_amd.fit import fit  # noqa: E402
from dgs_amd.synthetic import write_dynamic_dnerf  # noqa: E402

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "scene")
write_dynamic_dnerf(data, n_train=24, n_test=2, H=128, W=128, device=dev)
kw = dict(iterations=int(os.environ.get("ITERS", "460")), device=dev, num_pts=6000, node_num=128, seed=0, warm_up=150, regularize_from=300,
          densify_from=200, opacity_reset_interval=400, deterministic=True)
out = {}
for tag, graph, order in (("eager", False, True), ("graph", True, True), ("eager_free", False, False), ("graph_free", True, False)):
    tr, losses = fit(data, os.path.join(tmp, tag), graph=graph, reference_update_order=order, **kw)
    tr.set_deterministic(False)
    out[tag] = np.asarray(losses)
    print(tag, "mean loss last 50: %.5f" % out[tag][-50:].mean(), "surfels", tr.surfels.num_surfels, flush=True)
for a, b in (("eager", "graph"), ("eager_free", "graph_free")):
    d = np.abs(out[a] - out[b])
    first = int(np.argmax(d > 0)) if (d > 0).any() else -1
    print(a, "vs", b, "first differing iteration", first + 1, "max |d|", float(d.max()), "around:", out[a][max(first - 1, 0):first + 3], out[b][max(first - 1, 0):first + 3])
