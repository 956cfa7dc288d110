"""Generates tests/golden/floyd_golden.npz by IMPORTING the reference's utils/time_utils.py / utils/deform_utils.py and running, on the
CPU, the pieces of ControlNodeWarp that sit off the default training path:

* geodesic_distance_floyd (time_utils.py:1122-1131, deform_utils.py:47-56): K-nearest-neighbour graph of the nodes, all-pairs shortest
  paths;
* cal_nn_weight_floyd (time_utils.py:969-984): skinning weights along the graph (surfels against nodes, and nodes against nodes);
* cal_connectivity_from_points with a trajectory and in its 'floyd' mode (deform_utils.py:58-110);
* arap_deformation_loss (deform_utils.py:246-289) and ControlNodeWarp.arap_loss_with_rot (time_utils.py:1035-1042), with the gradient
  with respect to the deformation network's translation and rotation heads.

The reference's random draws (torch.rand for the time samples, torch.randint for the target frame) are recorded.
Run from the repo root:  python tests/golden/make_floyd_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

from make_deform_golden import fill_params, import_reference, knn_stub

HERE = os.path.dirname(os.path.abspath(__file__))
M = 80


def main():
    tu = import_reference()
    sys.modules["pytorch3d.ops"].knn_points = lambda p1, p2, l1=None, l2=None, K=1, **kw: _KnnResult(knn_stub(p1, p2, K=K))
    import utils.deform_utils as du
    tu.pytorch3d.ops.knn_points = sys.modules["pytorch3d.ops"].knn_points
    out = {}
    g = torch.Generator().manual_seed(11)
    # ---- graph distances: a bent sheet, so that geodesic and Euclidean neighbours differ
    u = torch.rand(M, 2, generator=g)
    cur = torch.stack([torch.cos(3.0 * u[:, 0]) * (0.4 + 0.02 * u[:, 1]), torch.sin(3.0 * u[:, 0]) * 0.4, u[:, 1] * 0.5], -1)
    x = cur[torch.randint(0, M, (150,), generator=g)] + 0.03 * torch.randn(150, 3, generator=g)
    out["cur_node"], out["x"] = cur.numpy().copy(), x.numpy().copy()
    torch.manual_seed(0)
    ref = tu.ControlNodeWarp(is_blender=True, node_num=M, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=False, with_arap_loss=False,
                             with_node_weight=True)
    for K in (2, 4):
        out["geo_K%d" % K] = ref.geodesic_distance_floyd(cur_node=cur, K=K).numpy().copy()
        assert np.array_equal(out["geo_K%d" % K], du.geodesic_distance_floyd(cur, K=K).numpy())
    w, d, idx = ref.cal_nn_weight_floyd(x=x, t0=torch.tensor([0.3]), cur_node=cur, K=8, GraphK=3, temperature=1e-3, XisNode=False)
    out["floyd_w"], out["floyd_d"], out["floyd_idx"] = w.numpy().copy(), d.numpy().copy(), idx.numpy().copy()
    w, d, idx = ref.cal_nn_weight_floyd(x=cur, t0=torch.tensor([0.3]), cur_node=cur, K=9, GraphK=4, temperature=1e-1, cache_name="p2dR",
                                        XisNode=True)
    out["floyd_self_w"], out["floyd_self_d"], out["floyd_self_idx"] = w.numpy().copy(), d.numpy().copy(), idx.numpy().copy()
    # ---- connectivity from a trajectory, both modes
    traj = cur[:, None, :] + 0.05 * torch.randn(M, 5, 3, generator=g).cumsum(1)
    out["traj"] = traj.numpy().copy()
    for mode in ("nn", "floyd"):
        ii, jj, nn, weight = du.cal_connectivity_from_points(cur, radius=0.15, K=6, trajectory=traj, mode=mode, GraphK=3)
        out["conn_%s_ii" % mode], out["conn_%s_jj" % mode], out["conn_%s_nn" % mode] = ii.numpy(), jj.numpy(), nn.numpy()
        out["conn_%s_w" % mode] = weight.numpy().copy()
    ii, jj, nn, weight = du.cal_connectivity_from_points(cur, radius=0.15, K=6, mode="floyd", GraphK=3)
    out["conn_pts_floyd_ii"], out["conn_pts_floyd_jj"], out["conn_pts_floyd_nn"] = ii.numpy(), jj.numpy(), nn.numpy()
    out["conn_pts_floyd_w"] = weight.numpy().copy()
    rad = 0.1 + 0.1 * torch.rand(M, generator=g)
    ii, jj, nn, weight = du.cal_connectivity_from_points(cur, radius=0.2, K=6, node_radius=rad, adaptive_weighting=False)
    out["node_radius_conn"] = rad.numpy().copy()
    out["conn_rad_ii"], out["conn_rad_jj"], out["conn_rad_nn"], out["conn_rad_w"] = ii.numpy(), jj.numpy(), nn.numpy(), weight.numpy().copy()
    # ---- arap_loss_with_rot on a module with absolute node rotations (d_rot_as_res = False) and on one without
    to_saved = torch.Tensor.to     # produce_edge_matrix_nfmt moves its result to "cuda" (deform_utils.py:39)
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else to_saved(self, *a, **k)
    orig_rand, orig_randint = torch.rand, torch.randint
    # arap_deformation_loss weights its K = 50 neighbours by exp(-d / mean d) AFTER setting the distances of neighbours beyond the radius
    # (bounding-box diagonal / 8) to inf: one dropped neighbour makes every weight NaN (deform_utils.py:91-94).  The finite case therefore
    # needs every node to have 50 others within the radius of the time-averaged trajectory distance: nodes in a small ball, 16 times.
    v = torch.randn(M, 3, generator=g)
    ball = 0.3 * v / v.norm(dim=-1, keepdim=True) * torch.rand(M, 1, generator=g) ** (1 / 3)
    out["ball"] = ball.numpy().copy()
    try:
        for tag, as_res in (("rot", False), ("norot", True)):
            torch.manual_seed(0)
            ref = tu.ControlNodeWarp(is_blender=True, node_num=M, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=as_res,
                                     with_arap_loss=False, with_node_weight=True)
            fill_params(ref)
            with torch.no_grad():
                ref.network.gaussian_warp.weight.mul_(50.0)
                ref.nodes.data = torch.cat([ball, 0.01 * torch.ones(M, 8)], -1)
                ref._node_radius.data = torch.log(torch.tensor(0.26)) + 0.1 * torch.randn(M, generator=g)
            drawn = {}

            def rand(*a, **k):
                v = orig_rand(*a, generator=g)
                drawn["t_samp"] = v.clone()
                return v

            def randint(*a, **k):
                v = orig_randint(*a, generator=g)
                drawn["fid"] = v.clone()
                return v
            torch.rand, torch.randint = rand, randint
            loss = ref.arap_loss_with_rot(t_samp_num=16)
            torch.rand, torch.randint = orig_rand, orig_randint
            ref.zero_grad()
            loss.backward()
            out[tag + "_node_radius_raw"] = ref._node_radius.data.numpy().copy()
            out[tag + "_t_samp"], out[tag + "_fid"] = drawn["t_samp"].numpy(), np.int64(drawn["fid"].item())
            out[tag + "_loss"] = np.float64(loss.item())
            out[tag + "_grad_warp"] = ref.network.gaussian_warp.weight.grad.numpy().copy()
            if not as_res:
                out[tag + "_grad_rot"] = ref.network.gaussian_rotation.weight.grad.numpy().copy()
    finally:
        torch.rand, torch.randint = orig_rand, orig_randint
        torch.Tensor.to = to_saved
    np.savez_compressed(os.path.join(HERE, "floyd_golden.npz"), **out)
    print({k: float(v) for k, v in out.items() if k.endswith("_loss")}, {k: v.shape for k, v in out.items() if k.startswith("conn") and k.endswith("ii")})


class _KnnResult(tuple):
    """knn_points returns a namedtuple (dists, idx, knn): the reference both unpacks it and reads .dists / .idx."""
    dists = property(lambda s: s[0])
    idx = property(lambda s: s[1])
    knn = property(lambda s: s[2])


if __name__ == "__main__":
    main()
