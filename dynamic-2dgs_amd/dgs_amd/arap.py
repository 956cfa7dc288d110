"""As-rigid-as-possible regulariser of the control nodes: ControlNodeWarp.arap_loss
(utils/time_utils.py:1080-1089) with cal_connectivity_from_points / estimate_rotation / cal_arap_error
(utils/deform_utils.py:58-110,130-166,177-205) and the landmark schedule of its weight (time_utils.py:485-503,791-792).

It acts on the M control nodes only (M x 2 time samples x 10 neighbours, one batched 3x3 SVD): PyTorch operations on whatever
device the nodes live on.  Active for iterations < 20000 in the reference (weight 1e-4 -> 1e-5 -> 0); the captured train step
of this package is the regime after that, so the regulariser runs eagerly (Trainer(arap=True)).
The random draws of the reference (the time samples, the 512-node subsample) can be passed in, so that results are
reproducible and comparable with the reference's.

Also here (round 5): the two node regularisers of the reference's node PRE-TRAINING stage (train_gui.py:502-504; the stage itself is
dgs_amd/node_pretrain.py since round 6) -- elastic_loss and acc_loss (utils/time_utils.py:1091-1120), pinned by the imported
reference like arap_loss.

Round 6, off the default training path (no caller in train_gui.py's two stages; the graph distances serve the GUI's drag editing,
utils/time_utils.py:1169-1204): geodesic_distance_floyd / nn_weight_floyd (time_utils.py:1122-1131, 969-984), the trajectory and 'floyd'
modes of cal_connectivity_from_points (deform_utils.py:58-110), arap_deformation_loss (deform_utils.py:246-289) and
ControlNodeWarp.arap_loss_with_rot (time_utils.py:1035-1042) -- pinned by the imported reference (tests/golden/make_floyd_golden.py).
"""
import math

import torch

LAMBDA_ARAP_LANDMARKS = (1e-4, 1e-4, 1e-5, 1e-5, 0)
LAMBDA_ARAP_STEPS = (0, 5000, 10000, 20000, 20001)


def landmark_interpolate(landmarks, steps, step):
    """utils/time_utils.py:485-503, interpolation='log'."""
    stage = sum(1 for s in steps if step >= s)
    if stage == len(steps):
        return max(0, landmarks[-1])
    if stage == 0:
        return 0
    l1, l2 = landmarks[stage - 1], landmarks[stage]
    if l2 <= 0:
        return 0
    ratio = (step - steps[stage - 1]) / (steps[stage] - steps[stage - 1])
    return math.exp(math.log(l1) * (1 - ratio) + math.log(l2) * ratio)


def lambda_arap(iteration):
    return landmark_interpolate(LAMBDA_ARAP_LANDMARKS, LAMBDA_ARAP_STEPS, iteration)


def _sq_dists(a, b):
    return (a[:, None, :] - b[None, :, :]).pow(2).sum(-1)


def geodesic_distance_floyd(cur_node, K=8):
    """geodesic_distance_floyd (utils/time_utils.py:1122-1131 = utils/deform_utils.py:47-56): every node is joined to its K nearest
    others (Euclidean edge lengths, the graph made symmetric), then all-pairs shortest paths by Floyd-Warshall -- one [M, M] minimum
    per pivot.  Unreachable pairs stay inf; the diagonal is 0 (a node is its own nearest neighbour)."""
    M = cur_node.shape[0]
    nn_dist, nn_idx = _sq_dists(cur_node, cur_node).topk(min(K + 1, M), dim=1, largest=False)
    dist = torch.full((M, M), float("inf"), dtype=torch.float32, device=cur_node.device)
    dist.scatter_(1, nn_idx, nn_dist ** 0.5)
    dist = torch.minimum(dist, dist.T)
    for i in range(M):
        dist = torch.minimum(dist[:, i, None] + dist[None, i, :], dist)
    return dist


def nn_weight_floyd(x, cur_node, K, GraphK=2, temperature=1.0, XisNode=False, t0=None, cache=None):
    """ControlNodeWarp.cal_nn_weight_floyd (utils/time_utils.py:969-984): the K graph-nearest nodes of every point's NEAREST node, at
    distance (squared distance to that node) + (graph distance from it) -- the reference adds the two as they are --, softmax(-d /
    temperature) weights.  XisNode: the points are the nodes themselves, the node itself (graph distance 0) is skipped.
    `cache`: a dict that keeps the sorted graph distances between calls, recomputed when `t0` moves by more than 1e-2 (the
    reference's per-name attributes).  Returns (weights [N, K], distances [N, K], node indices [N, K])."""
    cache = {} if cache is None else cache
    if "nn_dist" not in cache or (t0 is not None and (cache["t"] - t0).abs().max() > 1e-2):
        g_dist, g_idx = geodesic_distance_floyd(cur_node, K=GraphK).sort(dim=1)
        off = 1 if XisNode else 0
        cache["nn_dist"], cache["nn_idx"] = g_dist[:, off:K + off], g_idx[:, off:K + off]
        if t0 is not None:
            cache["t"] = t0.clone()
    d1, i1 = _sq_dists(x, cur_node).min(dim=1)
    knn_dist, knn_idx = cache["nn_dist"][i1] + d1[:, None], cache["nn_idx"][i1]
    return torch.softmax(-knn_dist / temperature, dim=-1), knn_dist, knn_idx


def connectivity_from_points(points=None, K=10, radius=0.1, least_edge_num=3, trajectory=None, mode="nn", GraphK=4,
                             adaptive_weighting=True, node_radius=None):
    """cal_connectivity_from_points (utils/deform_utils.py:58-110): the K nearest other points of every point; beyond the first
    `least_edge_num`, neighbours farther than `radius` are dropped.  Distances are squared Euclidean between the points, or between the
    flattened trajectories [Nv, T, 3] divided by T; mode 'floyd' takes squared graph distances (geodesic_distance_floyd over `GraphK`
    neighbours) instead.  Weights [Nv, K]: exp(-d / mean d) (the default; the dropped distances are inf when the mean is taken, so
    with ANY neighbour dropped the rows with a dropped one are NaN and the others uniform -- the reference's arithmetic, kept), exp(-d) or exp(-d / (2 r_j^2)) with `node_radius`;
    normalised per point.  Returns the edge lists (ii, jj, nn) and the weights."""
    Nv = points.shape[0] if points is not None else trajectory.shape[0]
    q = points if trajectory is None else trajectory.reshape(Nv, -1) / trajectory.shape[1]
    if mode == "floyd":
        d = geodesic_distance_floyd(q, K=GraphK) ** 2
        d[torch.eye(Nv, dtype=torch.bool, device=d.device)] = float("inf")
        nn_dist, nn_idx = d.sort(dim=1)
        nn_dist, nn_idx = nn_dist[:, :K].clone(), nn_idx[:, :K].clone()
    else:
        nn_dist, nn_idx = _sq_dists(q, q).topk(min(K + 1, Nv), dim=1, largest=False)
        nn_dist, nn_idx = nn_dist[:, 1:].clone(), nn_idx[:, 1:].clone()         # without themselves
    far = nn_dist[:, least_edge_num:] >= radius ** 2
    nn_idx[:, least_edge_num:] = torch.where(far, torch.full_like(nn_idx[:, least_edge_num:], -1), nn_idx[:, least_edge_num:])
    nn_dist[:, least_edge_num:] = torch.where(far, torch.full_like(nn_dist[:, least_edge_num:], float("inf")), nn_dist[:, least_edge_num:])
    if adaptive_weighting:
        weight = torch.exp(-nn_dist / nn_dist.mean())
    elif node_radius is None:
        weight = torch.exp(-nn_dist)
    else:
        weight = torch.exp(-nn_dist / (2 * node_radius[nn_idx] ** 2))      # index -1 reads the last node, like the reference; its d is inf
    weight = weight / weight.sum(dim=-1, keepdim=True)
    Kn = nn_idx.shape[1]
    ii = torch.arange(Nv, device=q.device)[:, None].expand(Nv, Kn).reshape(-1)
    jj = nn_idx.reshape(-1)
    nn = torch.arange(Kn, device=q.device)[None].expand(Nv, Kn).reshape(-1)
    mask = jj != -1
    return ii[mask], jj[mask], nn[mask], weight


def edge_matrix(verts, shape, ii, jj, nn):
    """produce_edge_matrix_nfmt: E[i, n] = p_i - p_(J[n])."""
    E = torch.zeros(shape, dtype=verts.dtype, device=verts.device)
    E[ii, nn] = verts[ii] - verts[jj]
    return E


@torch.no_grad()
def estimate_rotation(source, target, ii, jj, nn, K, weight, sample_idx):
    """Per-node best-fit rotation source edges -> target edges (weighted Procrustes through a batched SVD)."""
    Nv = source.shape[0]
    src = edge_matrix(source, (Nv, K, 3), ii, jj, nn)[sample_idx]
    tgt = edge_matrix(target, (Nv, K, 3), ii, jj, nn)[sample_idx]
    S = torch.bmm(src.permute(0, 2, 1), weight[..., None] * tgt)
    unchanged = torch.unique(torch.where((src == tgt).all(dim=1))[0])
    S[unchanged] = 0
    U, sig, W = torch.svd(S)
    R = torch.bmm(W, U.permute(0, 2, 1))
    flip = torch.nonzero(torch.det(R) <= 0, as_tuple=False).flatten()
    if flip.numel() > 0:
        Umod = U.clone()
        cols = torch.argmin(sig[flip], dim=1)
        Umod[flip, :, cols] *= -1
        R[flip] = torch.bmm(W[flip], Umod[flip].permute(0, 2, 1))
    return R


def arap_error(nodes_sequence, ii, jj, nn, K=10, sample_num=512, sample_idx=None, generator=None):
    """cal_arap_error with unit edge weights: sum over the later frames of | target edges - R source edges |^2 for (a
    subsample of) the nodes.  nodes_sequence [Nt, Nv, 3]."""
    Nt, Nv, _ = nodes_sequence.shape
    dev = nodes_sequence.device
    weight = torch.zeros(Nv, K, dtype=nodes_sequence.dtype, device=dev)
    weight[ii, nn] = 1
    if sample_idx is None:
        if Nv > sample_num:   # with replacement, like np.random.choice
            sample_idx = torch.randint(0, Nv, (sample_num,), generator=generator, device=dev)
        else:
            sample_idx = torch.arange(Nv, device=dev)
    src = edge_matrix(nodes_sequence[0], (Nv, K, 3), ii, jj, nn)[sample_idx]
    weight = weight[sample_idx]
    err = nodes_sequence.new_zeros(())
    for t in range(1, Nt):
        R = estimate_rotation(nodes_sequence[0], nodes_sequence[t], ii, jj, nn, K, weight, sample_idx)
        tgt = edge_matrix(nodes_sequence[t], (Nv, K, 3), ii, jj, nn)[sample_idx]
        rigid = torch.bmm(R, src.permute(0, 2, 1)).permute(0, 2, 1)
        err = err + (weight * (tgt - rigid).norm(dim=2) ** 2).sum()
    return err


def arap_loss(deform, t=None, delta_t=0.05, t_samp_num=2, t_samp=None, sample_idx=None, generator=None):
    """ControlNodeWarp.arap_loss: node positions at `t_samp_num` random times within delta_t of t (or of a random time),
    connectivity from the first sample, ARAP error of the later samples against it."""
    nodes = deform.nodes
    live = getattr(deform, "live_nodes", None)
    if live is not None and not bool(live.all()):
        nodes = nodes[live]     # padding nodes (ControlNodes.live_nodes) would become each other's neighbours
    dev = nodes.device
    M = nodes.shape[0]
    if t_samp is None:
        rnd = lambda *s: torch.rand(*s, generator=generator, device=dev)
        t0 = rnd([]) if t is None else t.reshape(-1)[0] + delta_t * (rnd([]) - 0.5)
        t_samp = rnd(t_samp_num) * delta_t + t0 - 0.5 * delta_t
    T = t_samp.shape[0]
    x = nodes[:, None, :3].detach().expand(M, T, 3).reshape(-1, 3)
    d_xyz = deform.network(x, t_samp[None, :, None].expand(M, T, 1).reshape(-1, 1))["d_xyz"].view(M, T, 3)
    nodes_t = nodes[:, None, :3].detach() + d_xyz
    ii, jj, nn, _ = connectivity_from_points(nodes_t[:, 0], K=10)
    return arap_error(nodes_t.permute(1, 0, 2), ii, jj, nn, sample_idx=sample_idx, generator=generator)


# ---- the two node regularisers of the node pre-training stage (train_gui.py:502-504) --------------------------------------------
def _node_positions_at(deform, t_samp):
    """nodes[:, None, :3].detach() + node_deform(t)['d_xyz'] for the time samples t_samp [T] -> [M, T, 3] (live nodes only)."""
    nodes = deform.nodes
    live = getattr(deform, "live_nodes", None)
    if live is not None and not bool(live.all()):
        nodes = nodes[live]
    M, T = nodes.shape[0], t_samp.shape[0]
    x = nodes[:, None, :3].detach().expand(M, T, 3).reshape(-1, 3)
    d_xyz = deform.network(x, t_samp[None, :, None].expand(M, T, 1).reshape(-1, 1))["d_xyz"].view(M, T, 3)
    return nodes, nodes[:, None, :3].detach() + d_xyz


def _sample_times(t, delta_t, n, generator, dev, t_samp):
    if t_samp is not None:
        return t_samp
    rnd = lambda *s: torch.rand(*s, generator=generator, device=dev)
    t0 = rnd([]) if t is None else t.reshape(-1)[0] + delta_t * (rnd([]) - 0.5)
    return rnd(n) * delta_t + t0 - 0.5 * delta_t


def elastic_loss(deform, t=None, delta_t=0.005, K=2, t_samp_num=8, t_samp=None, generator=None):
    """ControlNodeWarp.elastic_loss (utils/time_utils.py:1091-1108): the lengths of every node's edges to its K nearest other nodes
    (11-D neighbour search with the skinning weights of cal_nn_weight) should not vary over `t_samp_num` times within delta_t of t;
    each edge's variance is normalised by its own detached value (so the LOSS is ~ the weighted edge count and only its gradient
    matters), weighted by the skinning weight of the neighbour, summed per node, averaged."""
    dev = deform.nodes.device
    t_samp = _sample_times(t, delta_t, t_samp_num, generator, dev, t_samp)
    nodes, nodes_t = _node_positions_at(deform, t_samp)
    # cal_nn_weight(x = node positions, feature = node hyper coordinates, K + 1): the node itself comes first and is dropped
    live = getattr(deform, "live_nodes", None)
    if live is not None and not bool(live.all()):
        raise NotImplementedError("elastic_loss with padding nodes: call it before ControlNodes.pad_nodes / on the unpadded module")
    K0 = deform.K
    try:
        deform.K = K + 1
        w, _, idx = deform.nn_weights(nodes[:, :3].detach(), nodes[:, 3:])
    finally:
        deform.K = K0
    # nn_weights leaves the neighbours in approximately ascending order: put the node itself first like pytorch3d does
    self_col = (idx == torch.arange(nodes.shape[0], device=dev)[:, None]).float().argmax(1)
    order = torch.arange(K + 1, device=dev)[None].expand(nodes.shape[0], K + 1).clone()
    order[torch.arange(nodes.shape[0]), self_col] = -1
    order = order.sort(dim=1).indices            # the self column first, the others in their order
    w, idx = torch.gather(w, 1, order)[:, 1:], torch.gather(idx, 1, order)[:, 1:]
    edge_t = (nodes_t[idx] - nodes_t[:, None]).norm(dim=-1)        # [M, K, T]
    var = edge_t.var(dim=2)
    var = var / (var.detach() + 1e-5)
    return (var * w).sum(dim=1).mean()


def acc_loss(deform, t=None, delta_t=0.005, t0=None, generator=None):
    """ControlNodeWarp.acc_loss (utils/time_utils.py:1110-1120): |x(t - dt) + x(t + dt) - 2 x(t)| of every node, normalised by its
    own detached value, averaged.  t0: the (already jittered) centre time, for reproducible comparisons."""
    dev = deform.nodes.device
    if t0 is None:
        rnd = torch.rand([], generator=generator, device=dev)
        t0 = rnd if t is None else t.reshape(-1)[0] + delta_t * (rnd - 0.5)
    t3 = torch.stack([t0 - delta_t, t0, t0 + delta_t]).to(dev)
    _, nodes_t = _node_positions_at(deform, t3)
    acc = (nodes_t[:, 0] + nodes_t[:, 2] - 2 * nodes_t[:, 1]).norm(dim=-1)
    acc = acc / (acc.detach() + 1e-5)
    return acc.mean()


# ---- off the training path: the rotation-aware ARAP term (no caller in the reference's train_gui.py) ----------------------------------
def arap_deformation_loss(trajectory, node_radius=None, trajectory_rot=None, K=50, with_rot=True, fid=None, generator=None):
    """arap_deformation_loss (utils/deform_utils.py:246-289).  trajectory [N, T, 3]: frame 0 against ONE random later frame `fid`;
    connectivity from the whole trajectory (K neighbours, radius = bounding-box diagonal of frame 0 / 8, adaptive weights -- finite only
    while no node loses a neighbour, see connectivity_from_points); per-node rotation by weighted Procrustes (no gradient through it);
    ARAP error = sum over (k, xyz) of the node-mean of (w (e' - R e))^2.  with_rot: `trajectory_rot` [N, T, 4] are the nodes' absolute
    rotations (quaternions), and R q_0 should equal q_fid: 100 x the same kind of mean.  Returns (arap_error, rot_error)."""
    init_pcl = trajectory[:, 0]
    N, T = trajectory.shape[0], trajectory.shape[1]
    if fid is None:
        fid = int(torch.randint(1, T, [], generator=generator))
    tar_pcl = trajectory[:, fid]
    with torch.no_grad():
        radius = torch.linalg.norm(init_pcl.max(dim=0).values - init_pcl.min(dim=0).values) / 8
        ii, jj, nn, weight = connectivity_from_points(init_pcl, K=K, radius=radius, trajectory=trajectory.detach(), node_radius=node_radius)
    Kn = weight.shape[1]
    P = edge_matrix(init_pcl, (N, Kn, 3), ii, jj, nn)
    P_prime = edge_matrix(tar_pcl, (N, Kn, 3), ii, jj, nn)
    with torch.no_grad():
        S = torch.bmm(P.permute(0, 2, 1), weight[..., None] * P_prime)
        U, sig, W = torch.svd(S)
        R = torch.bmm(W, U.permute(0, 2, 1))
        flip = torch.nonzero(torch.det(R) <= 0, as_tuple=False).flatten()
        if flip.numel() > 0:
            Umod = U.clone()
            cols = torch.argmin(sig[flip], dim=1)
            Umod[flip, :, cols] *= -1
            R[flip] = torch.bmm(W[flip], Umod[flip].permute(0, 2, 1))
    err = (weight[..., None] * (P_prime - torch.einsum("bxy,bky->bkx", R, P))).square().mean(dim=0).sum()
    if not with_rot:
        return err, 0.0
    from .deform import quaternion_to_matrix
    R_rot = torch.bmm(R, quaternion_to_matrix(trajectory_rot[:, 0]))
    rot_err = (R_rot - quaternion_to_matrix(trajectory_rot[:, fid])).square().mean(dim=0).sum()
    return err, rot_err * 1e2


def arap_loss_with_rot(deform, t_samp_num=8, d_rot_as_res=True, t_samp=None, fid=None, generator=None):
    """ControlNodeWarp.arap_loss_with_rot (utils/time_utils.py:1035-1042): node positions (and, when the network's rotation head is an
    ABSOLUTE node rotation, d_rot_as_res=False, the rotations) at `t_samp_num` uniform random times -> arap_deformation_loss, the two
    terms added.  This package's modules use the rotation head as a residual (the reference's default), hence the default."""
    nodes = deform.nodes
    live = getattr(deform, "live_nodes", None)
    if live is not None and not bool(live.all()):
        nodes = nodes[live]
    dev = nodes.device
    if t_samp is None:
        t_samp = torch.rand(t_samp_num, generator=generator, device=dev)
    M, T = nodes.shape[0], t_samp.shape[0]
    x = nodes[:, None, :3].detach().expand(M, T, 3).reshape(-1, 3)
    out = deform.network(x, t_samp[None, :, None].expand(M, T, 1).reshape(-1, 1))
    trajectory = nodes[:, None, :3].detach() + out["d_xyz"].view(M, T, 3)
    rot = None if d_rot_as_res else out["d_rotation"].view(M, T, 4)
    radius = deform.node_radius
    radius = radius[live] if radius.shape[0] != M else radius
    a, r = arap_deformation_loss(trajectory, node_radius=radius.detach(), trajectory_rot=rot, with_rot=not d_rot_as_res, fid=fid,
                                 generator=generator)
    return a + r
