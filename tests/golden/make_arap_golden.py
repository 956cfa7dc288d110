"""Generates tests/golden/arap_golden.npz by IMPORTING the reference's utils/time_utils.py / utils/deform_utils.py and running
ControlNodeWarp.arap_loss (time_utils.py:1080-1089) on the CPU for 64 nodes (no subsampling) and 600 nodes (512-node
subsample), with its gradient with respect to the deformation network's last layer, plus landmark_interpolate values.
The reference's random draws (torch.rand for the time samples, np.random.choice for the subsample) are recorded.
Run from the repo root:  python tests/golden/make_arap_golden.py
"""
import os

import numpy as np
import torch

from make_deform_golden import fill_params, import_reference, knn_stub

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tu = import_reference()
    import sys
    sys.modules["pytorch3d.ops"].knn_points = lambda p1, p2, l1=None, l2=None, K=1, **kw: __import__("types").SimpleNamespace(
        **dict(zip(("dists", "idx", "knn"), knn_stub(p1, p2, K=K))))
    import utils.deform_utils as du
    out = {}
    for tag, M in (("small", 64), ("large", 600)):
        torch.manual_seed(0)
        ref = tu.ControlNodeWarp(is_blender=True, node_num=M, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=True,
                                 with_arap_loss=True, with_node_weight=True)
        fill_params(ref)
        with torch.no_grad():   # a deformation large enough for non-trivial rotations
            ref.network.gaussian_warp.weight.mul_(50.0)
        g = torch.Generator().manual_seed(3)
        ref.nodes.data = torch.cat([(torch.rand(M, 3, generator=g) * 2 - 1) * 0.6, 0.01 * torch.ones(M, 8)], -1)
        drawn = []
        orig_rand, orig_choice = torch.rand, np.random.choice

        def rand(*a, **k):
            v = orig_rand(*a, generator=g)
            drawn.append(v.clone())
            return v

        def choice(n, k):
            v = torch.randint(0, n, (k,), generator=g).numpy()
            out[tag + "_sample_idx"] = v.copy()
            return v
        torch.rand, np.random.choice = rand, choice
        to_saved = torch.Tensor.to     # produce_edge_matrix_nfmt moves its result to "cuda" (deform_utils.py:39)
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else to_saved(self, *a, **k)
        try:
            loss = ref.arap_loss(t=torch.tensor([0.4]))
        finally:
            torch.rand, np.random.choice = orig_rand, orig_choice
            torch.Tensor.to = to_saved
        loss.backward()
        out[tag + "_nodes"] = ref.nodes.data.numpy().copy()
        out[tag + "_t_jitter"] = drawn[0].numpy()
        out[tag + "_t_samp_raw"] = drawn[1].numpy()
        out[tag + "_t_samp"] = (drawn[1] * 0.05 + (0.4 + 0.05 * (drawn[0] - 0.5)) - 0.5 * 0.05).numpy()
        out[tag + "_loss"] = np.float64(loss.item())
        out[tag + "_grad_warp"] = ref.network.gaussian_warp.weight.grad.numpy().copy()
    steps = np.array([0, 1, 2500, 5000, 7500, 10000, 15000, 20000, 20001, 30000])
    ref_l = tu.ControlNodeWarp(is_blender=True, node_num=16, K=3, hyper_dim=8, local_frame=True, with_arap_loss=True)
    out["lambda_steps"] = steps
    out["lambda_arap"] = np.array([tu.landmark_interpolate(ref_l.lambda_arap_landmarks, ref_l.lambda_arap_steps, int(s)) for s in steps], np.float64)
    np.savez_compressed(os.path.join(HERE, "arap_golden.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "loss" in k or "lambda_arap" == k}, out["small_loss"], out["large_loss"])


if __name__ == "__main__":
    main()
