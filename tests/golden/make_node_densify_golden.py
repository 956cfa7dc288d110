"""Generates tests/golden/node_densify_golden.npz by IMPORTING the reference's utils/time_utils.py and running
ControlNodeWarp.densify (time_utils.py:1269-1385, with cal_node_importance) on the CPU: 48 nodes (four of them far from every
Gaussian -> pruned), 500 Gaussians with a gradient statistic that is large around a few nodes (-> new nodes at the
importance-weighted mean of the affected Gaussians), an Adam optimiser holding non-trivial moments.
The node cloud's visualisation twin (`self.gs`, a GaussianModel that mirrors the nodes) is replaced by a no-op object: it does
not feed back into the node parameters.  Run from the repo root:  python tests/golden/make_node_densify_golden.py
"""
import os
import types

import numpy as np
import torch

from make_deform_golden import fill_params, import_reference

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tu = import_reference()
    torch.manual_seed(0)
    M, N = 48, 500
    ref = tu.ControlNodeWarp(is_blender=True, node_num=M, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=True,
                             with_arap_loss=False, with_node_weight=True)
    fill_params(ref)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(N, 3, generator=g) * 2 - 1) * 1.3
    feature = 0.05 * torch.randn(N, 8, generator=g)
    nodes = torch.cat([x[:M].clone() + 0.01, 0.01 + 0.02 * torch.rand(M, 8, generator=g)], -1)
    nodes[5:9, :3] += 40.0                                   # four nodes no Gaussian is near
    ref.nodes.data = nodes
    ref._node_radius.data = torch.log(torch.tensor(0.26)) + 0.1 * torch.randn(M, generator=g)
    ref._node_weight.data = 0.3 * torch.randn(M, 1, generator=g)
    ref.inited.data = torch.ones_like(ref.inited)
    ref.gs = types.SimpleNamespace(densify_and_split=lambda **k: None, prune_points=lambda m: None, _xyz=types.SimpleNamespace(data=None))
    groups = [{"params": gr["params"], "lr": 1e-3, "name": gr["name"]} for gr in ref.trainable_parameters()]
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for it in range(2):
        for p in (ref.nodes, ref._node_radius, ref._node_weight):
            p.grad = 1e-2 * torch.randn(p.shape, generator=g)
        for p in ref.network.parameters():
            p.grad = torch.zeros_like(p)
        opt.step()
    x_grad = 1e-4 * torch.rand(N, 1, generator=g)
    hot = (x - x[11]).norm(dim=1) < 0.6
    x_grad[hot] += 3e-3
    out = dict(x=x.numpy(), feature=feature.numpy(), x_grad=x_grad.numpy().copy())
    for n in ("nodes", "_node_radius", "_node_weight"):
        p = getattr(ref, n)
        out["pre_" + n] = p.detach().numpy().copy()
        out["pre_m_" + n] = opt.state[p]["exp_avg"].numpy().copy()
        out["pre_v_" + n] = opt.state[p]["exp_avg_sq"].numpy().copy()
    imp, avg_x, edges = ref.cal_node_importance(x=x, weights=x_grad.norm(dim=-1), feature=feature)
    out.update(importance=imp.detach().numpy(), avg_x=avg_x.detach().numpy(), edge_count=edges.detach().numpy())
    ref.densify(max_grad=0.0002, optimizer=opt, x=x, x_grad=x_grad.clone(), feature=feature, force_dp=True)
    for n in ("nodes", "_node_radius", "_node_weight"):
        p = getattr(ref, n)
        out["post_" + n] = p.detach().numpy().copy()
        out["post_m_" + n] = opt.state[p]["exp_avg"].numpy().copy()
        out["post_v_" + n] = opt.state[p]["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "node_densify_golden.npz"), **out)
    print("nodes %d -> %d" % (M, out["post_nodes"].shape[0]), "selected", int((imp > 0.0002).sum()), "pruned", int((edges == 0).sum()))


if __name__ == "__main__":
    main()
