"""Development probe: held-out PSNR of fit() on the hidden dynamic scene of tests/test_learning_gpu.py -- as before, with the node
pre-training stage in front, and with the reference's update order."""
import math
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
from dgs_amd import io as dio  # noqa: E402
from dgs_amd.fit import fit  # noqa: E402
from dgs_amd.render import render  # noqa: E402
from dgs_amd.synthetic import write_dynamic_dnerf  # noqa: E402

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "scene")
write_dynamic_dnerf(data, n_train=60, n_test=12, H=200, W=200, device=dev)
test = dio.load_dnerf(data, num_pts=20_000)["test"]
bg = torch.zeros(3, device=dev)


def psnr(tr):
    vals = []
    with torch.no_grad():
        for f in test:
            cam = f.camera.to(dev)
            dv = tr.deform(tr.surfels.get_xyz.detach(), tr.deform.expand_time(cam.fid), tr.surfels.feature, tr.surfels.motion_mask)
            img = render(cam, tr.surfels, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])["render"]
            vals.append(-10.0 * math.log10(max(float(((img.clamp(0, 1).cpu() - f.image) ** 2).mean()), 1e-12)))
    return float(np.mean(vals))


IT = int(os.environ.get("ITERS", "6000"))
stage = dict(iterations=IT // 2, node_warm_up=IT // 10, sampling_at=3 * IT // 8, densify_interval=100, opacity_reset_interval=IT // 6)
for tag, kw in (("as before", {}), ("node pre-training stage first", dict(node_pretrain=stage)), ("reference update order", dict(reference_update_order=True)),
                ("both", dict(node_pretrain=stage, reference_update_order=True))):
    t0 = time.time()
    probes = {}
    tr, losses = fit(data, os.path.join(tmp, tag.replace(" ", "_")), iterations=IT, device=dev, num_pts=20_000, node_num=256, seed=0, deterministic=True,
                     warm_up=IT // 3 + 1, regularize_from=8 * IT // 9, on_iteration=lambda it, t: probes.__setitem__(it, psnr(t)) if it in (IT // 3, 2 * IT // 3, IT) else None, **kw)
    tr.set_deterministic(False)
    print("%-32s held-out PSNR at %s: %s | live surfels %d | mean loss of the last 300 iterations %.4f | %.0f s" % (
        tag, sorted(probes), [round(probes[k], 2) for k in sorted(probes)], tr.surfels.num_surfels, float(np.mean(losses[-300:])), time.time() - t0), flush=True)
