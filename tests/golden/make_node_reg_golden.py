"""Generates tests/golden/node_reg_golden.npz by IMPORTING the reference's utils/time_utils.py and running the two node regularisers of
its node pre-training stage (train_gui.py:502-504) on the CPU: ControlNodeWarp.elastic_loss (time_utils.py:1091-1108) and acc_loss
(:1110-1120), with their gradients with respect to the deformation network's translation head.  The reference's random draws
(torch.rand: the jitter of t, the time samples) are recorded so that the comparison does not depend on the RNG stream.
Run from the repo root:  python tests/golden/make_node_reg_golden.py
"""
import os

import numpy as np
import torch

from make_deform_golden import fill_params, import_reference

HERE = os.path.dirname(os.path.abspath(__file__))
M = 96


def main():
    tu = import_reference()
    torch.manual_seed(0)
    ref = tu.ControlNodeWarp(is_blender=True, node_num=M, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=True, with_arap_loss=False,
                             with_node_weight=True)
    fill_params(ref)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        ref.network.gaussian_warp.weight.mul_(50.0)
        ref.nodes.data = torch.cat([(torch.rand(M, 3, generator=g) * 2 - 1) * 0.6, 0.01 + 0.02 * torch.rand(M, 8, generator=g)], -1)
        ref._node_radius.data = torch.log(torch.tensor(0.26)) + 0.1 * torch.randn(M, generator=g)
        ref._node_weight.data = 0.3 * torch.randn(M, 1, generator=g)
    out = {"nodes": ref.nodes.data.numpy().copy(), "node_radius_raw": ref._node_radius.data.numpy().copy(),
           "node_weight_raw": ref._node_weight.data.numpy().copy()}
    orig_rand = torch.rand
    for name, call in (("elastic", lambda: ref.elastic_loss(t=torch.tensor([0.4]), delta_t=0.02)),
                       ("acc", lambda: ref.acc_loss(t=torch.tensor([0.4]), delta_t=0.06))):
        drawn = []

        def rand(*a, **k):
            v = orig_rand(*a, generator=g)
            drawn.append(v.clone())
            return v
        torch.rand = rand
        try:
            ref.zero_grad()
            loss = call()
        finally:
            torch.rand = orig_rand
        loss.backward()
        out[name + "_loss"] = np.float64(loss.item())
        out[name + "_grad_warp"] = ref.network.gaussian_warp.weight.grad.numpy().copy()
        out[name + "_jitter"] = drawn[0].numpy()
        if name == "elastic":
            out["elastic_t_samp"] = (drawn[1] * 0.02 + (0.4 + 0.02 * (drawn[0] - 0.5)) - 0.5 * 0.02).numpy()
        else:
            out["acc_t0"] = (0.4 + 0.06 * (drawn[0] - 0.5)).numpy()
    np.savez_compressed(os.path.join(HERE, "node_reg_golden.npz"), **out)
    print({k: float(v) for k, v in out.items() if k.endswith("_loss")}, {k: float(np.abs(v).max()) for k, v in out.items() if "grad" in k})


if __name__ == "__main__":
    main()
