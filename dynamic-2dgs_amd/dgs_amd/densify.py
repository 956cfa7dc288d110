"""Adaptive density control of the surfels: clone / split / prune / opacity reset with the Adam-state surgery
of scene/gaussian_model.py:309-486, re-designed for a FIXED number of slots.

The reference re-allocates every parameter, both Adam moments and the statistics whenever the point count changes
(cat_tensors_to_optimizer / _prune_optimizer).  Here the model owns `capacity` slots and a boolean `alive` mask: a pruned
surfel becomes a dead slot, a new one is written into a dead slot, a split parent is overwritten by its first child.
Addresses never change, so the captured HIP graphs of the train step, the flat gradient bucket and the flat Adam state stay
valid across densification, and nothing is re-captured or re-allocated until the slots run out (`Trainer.grow`).

A dead slot needs no special casing in the kernels: its opacity logit is DEAD_LOGIT (sigmoid == 0 exactly, so it produces no
tile-list entries and alpha < 1/255 everywhere), every gradient that reaches it is exactly zero, and Adam with zero gradient
and zero moments leaves a parameter unchanged (0 / (0 + eps)).  It costs the rasterizer's lists, sort and blend nothing, but it
still passes through the per-surfel kernels (neighbour search, skinning, preprocess, Adam): measured 0.37 ms per 100 k dead
slots next to 200 k live ones (1.34 -> 1.71 ms per step), which is why Trainer.grow adds a quarter, not multiples.

Data parallelism: the statistics are summed over the ranks inside the step's bucket all-reduce, parameters are replicas, and
the only random draw comes from a generator seeded identically on every rank -- each rank performs the same surgery and the
replicas stay bit-identical without any further exchange (tests/test_densify.py, world_size 2).

The result is the reference's set of surfels; only their ORDER differs (slots instead of survivors ++ clones ++ children),
which the rasterizer does not observe except for the tie-break of equal depths.
"""
import torch

DEAD_LOGIT = -1.0e4   # sigmoid(-1e4) == 0.0 in fp32


def _rotation_matrices(q):
    """general_utils.build_rotation (utils/general_utils.py:137-158): rows (r, x, y, z), normalised here."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), dim=1).view(-1, 3, 3)


def surfel_rows(model):
    """The per-surfel parameters (first dimension = slots), keyed by the reference's optimiser group names."""
    rows = {"xyz": model._xyz}
    if getattr(model, "packed_sh", False):
        rows["f_all"] = model._features
    else:
        rows["f_dc"], rows["f_rest"] = model._features_dc, model._features_rest
    rows.update(opacity=model._opacity, scaling=model._scaling, rotation=model._rotation, feature=model.feature)
    return rows


class TorchAdamMoments:
    """(exp_avg, exp_avg_sq) of a torch.optim.Adam parameter; (None, None) before its first step."""

    def __init__(self, optimizer):
        self.optimizer = optimizer

    def __call__(self, p):
        st = self.optimizer.state.get(p, None)
        return (st["exp_avg"], st["exp_avg_sq"]) if st else (None, None)


@torch.no_grad()
def densify_and_prune(model, moments, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01, N=2, noise=None,
                      generator=None):
    """GaussianModel.densify_and_prune (gaussian_model.py:466-482) on slots.  `moments(p)` returns the two Adam moment
    tensors of parameter p (shaped like p) or (None, None).  `noise`: optional standard-normal draws [N * n_split, 3] in the
    reference's order (row k * n_split + j = child k of the j-th split surfel); otherwise drawn from `generator`.
    Returns (n_cloned, n_split, n_pruned), or the int number of MISSING slots if the free ones do not suffice (nothing has
    been modified then: call Trainer.grow and retry)."""
    alive = model.alive
    rows = surfel_rows(model)
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    g = grads.squeeze(-1)
    scale_max = torch.exp(model._scaling).max(dim=1).values
    hot = alive & (g >= max_grad)
    clone = hot & (scale_max <= percent_dense * extent)          # densify_and_clone :449-464
    split = hot & (scale_max > percent_dense * extent)           # densify_and_split :418-447 (clones have zero statistics)
    src_clone, src_split = clone.nonzero().squeeze(1), split.nonzero().squeeze(1)
    n_clone, n_split = src_clone.numel(), src_split.numel()
    free = (~alive).nonzero().squeeze(1)
    need = n_clone + (N - 1) * n_split          # a split parent's slot takes its first child
    if need > free.numel():
        return need - free.numel()

    # children of the split surfels: N samples of the surfel's own (planar) Gaussian, scales shrunk by 0.8 N
    std = torch.exp(model._scaling[src_split]).repeat(N, 1)
    std = torch.cat((std, torch.zeros_like(std[:, :1])), dim=1)
    if callable(noise):      # (tests: draws keyed by the parents' positions, whatever order the slots hold them in)
        noise = noise(model._xyz[src_split])
    if noise is None:
        noise = torch.randn(std.shape, generator=generator, device=std.device, dtype=std.dtype)
    samples = noise.to(std) * std
    R = _rotation_matrices(model._rotation[src_split]).repeat(N, 1, 1)
    child_xyz = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + model._xyz[src_split].repeat(N, 1)
    child_scaling = torch.log(torch.exp(model._scaling[src_split]).repeat(N, 1) / (0.8 * N))

    dst_clone = free[:n_clone]
    dst_child = torch.cat((src_split, free[n_clone:need]))   # child 0 in place, children 1.. in free slots (same order as `noise`)
    src_child = src_split.repeat(N)
    for name, p in rows.items():
        m, v = moments(p)
        if name == "xyz":
            new_clone, new_child = p[src_clone], child_xyz
        elif name == "scaling":
            new_clone, new_child = p[src_clone], child_scaling
        else:
            new_clone, new_child = p[src_clone], p[src_child]
        p[dst_clone] = new_clone
        p[dst_child] = new_child
        for t in (m, v):                     # new rows start with fresh moments (cat_tensors_to_optimizer :373-397)
            if t is not None:
                t[dst_clone] = 0
                t[dst_child] = 0
    alive[dst_clone] = True
    alive[dst_child] = True

    # densification_postfix :414-416 clears the statistics of EVERY surfel -- before the prune mask reads max_radii2D, so
    # the reference's screen-size criterion (`big_points_vs`, :476) never fires; kept as is.
    model.xyz_gradient_accum.zero_()
    model.denom.zero_()
    model.max_radii2D.zero_()
    prune = torch.sigmoid(model._opacity).squeeze(-1) < min_opacity
    if max_screen_size:
        big_vs = model.max_radii2D > max_screen_size
        big_ws = torch.exp(model._scaling).max(dim=1).values > 0.1 * extent
        prune = prune | big_vs | big_ws
    prune &= alive
    n_pruned = int(prune.sum())
    kill(model, moments, prune)
    return n_clone, n_split, n_pruned


@torch.no_grad()
def kill(model, moments, mask):
    """prune_points (gaussian_model.py:346-364): the slots in `mask` become dead."""
    idx = mask.nonzero().squeeze(1)
    if idx.numel() == 0:
        return
    for name, p in surfel_rows(model).items():
        m, v = moments(p)
        if name == "opacity":
            p[idx] = DEAD_LOGIT
        for t in (m, v):
            if t is not None:
                t[idx] = 0
    model.alive[idx] = False
    model.xyz_gradient_accum[idx] = 0
    model.denom[idx] = 0
    model.max_radii2D[idx] = 0


@torch.no_grad()
def reset_opacity(model, moments):
    """GaussianModel.reset_opacity (gaussian_model.py:258-261): opacity <- min(opacity, 0.01), fresh moments."""
    o = torch.sigmoid(model._opacity)
    o = torch.min(o, torch.full_like(o, 0.01))
    new = torch.log(o / (1 - o))
    model._opacity.copy_(torch.where(model.alive[:, None], new, torch.full_like(new, DEAD_LOGIT)))
    for t in moments(model._opacity):
        if t is not None:
            t.zero_()
