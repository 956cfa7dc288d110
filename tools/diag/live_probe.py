import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_update_order_gpu as t
from dgs_amd.fit import fit
from dgs_amd.synthetic import write_dynamic_dnerf
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(); data = os.path.join(tmp, "scene")
write_dynamic_dnerf(data, n_train=24, n_test=2, H=128, W=128, device=dev)
stage = dict(iterations=100, node_warm_up=25, sampling_at=75, densify_interval=20, opacity_reset_interval=50)
for tag, extra in (("plain", {}), ("node stage", {"node_pretrain": stage}), ("node stage, default order", {"node_pretrain": stage, "reference_update_order": False})):
    logs = []
    tr, losses = fit(data, os.path.join(tmp, tag.replace(" ", "_").replace(",", "")), device=dev, log=logs.append, **dict(t.KW, **extra))
    print(tag, "live", tr.surfels.num_surfels, "loss first/last 20: %.4f %.4f" % (np.mean(losses[:20]), np.mean(losses[-20:])), [l for l in logs if "cloned" in l])
