"""The node pre-training stage of the reference trainer (GUI.train_node_rendering_step, train_gui.py:441-599): before the joint
stage starts, the control nodes themselves are rendered as small isotropic surfels through the SAME rasterizer and fitted to the
training views, so that the nodes end up where the scene moves and the deformation network already knows the motion.

What the stage does, per iteration `it` (1-based, `iterations` - 1 of them; defaults arguments/__init__.py:128-131):
  * a view is drawn without replacement (train_gui.py:466); the node surfels move by the NETWORK evaluated at their own positions
    (`query_network`, no skinning), detached while it < node_warm_up; rotation / scale offsets are zero;
  * loss = (1 - l) L1 + l (1 - SSIM); after the warm-up + 1e-3 elastic + 1e-5 acceleration + 1e-2 ARAP on the control nodes
    (dgs_amd/arap.py);
  * it < sampling_at: densification statistics, clone / split / prune (minimum opacity 0.005) every densify_interval iterations
    and at node_warm_up - 1, opacity reset every opacity_reset_interval;
  * it == sampling_at ('samp_hyper', train_gui.py:552-575): the node surfels are sampled down to `node_num` control nodes by
    farthest-point sampling of their TRAJECTORIES (positions at 16 times, a 48-D point each); the control nodes are re-initialised
    there and the node surfels restart from them with the survivors' appearance and a fresh optimiser; no update in this iteration;
  * it == iterations - 1: the node surfels' positions become the control nodes' positions; no update either;
  * otherwise both optimisers step (node surfels first; their position rate follows the reference's exponential schedule).

Two properties of the reference that are easy to miss and are kept:
  * its node surfels are created from `nodes[..., :3].detach()` without a copy (utils/time_utils.py:1236-1257 ->
    scene/gaussian_model.py:166), so their position parameter SHARES STORAGE with the control nodes until the first densification
    replaces it: every update of the node surfels moves the control nodes as well.  Here: `NodePretrainer.tied` + an explicit copy;
  * `StandardGaussianModel(all_the_same=True)` (scene/gaussian_model.py:489-497): ONE scale for all node surfels, the mean of the
    scale parameter -- which changes when rows are added, so clone, split and prune each see a different value.

Everything random comes from a `Draws` object (view choice, the regularisers' times, the split noise, the first index of a
farthest-point sampling) in the reference's order, so that a run can be replayed against the imported reference
(tests/golden/make_node_pretrain_golden.py).  The rasterizer is the package's operator (HIP on a device; tests inject the oracle).
"""
import math
import random

import torch
import torch.nn as nn

from . import arap as reg
from .deform import farthest_point_sample
from .densify import _rotation_matrices
from .io import mean_nn_dist2
from .losses import l1_loss, ssim
from .render import render
from .train import expon_lr


class Draws:
    """The stage's random numbers.  CPU generators (the values are moved to the device), so a seed gives the same run everywhere."""

    def __init__(self, seed=0):
        self.py = random.Random(seed)
        self.g = torch.Generator().manual_seed(seed)

    def pick(self, n):                       # random.randint(0, n - 1): the view (train_gui.py:466)
        return self.py.randint(0, n - 1)

    def rand(self, *shape):                  # torch.rand
        return torch.rand(*shape, generator=self.g)

    def randn(self, *shape):                 # the standard-normal draws behind torch.normal(mean, std)
        return torch.randn(*shape, generator=self.g)

    def start(self, n):                      # torch.randint(0, n, (1,)): where a farthest-point sampling starts
        return int(torch.randint(0, n, (1,), generator=self.g))

    def choice(self, n, k):                  # np.random.choice(n, k): the ARAP error's node subsample (utils/deform_utils.py:190)
        return torch.randint(0, n, (k,), generator=self.g)


class NodeSurfels(nn.Module):
    """The control nodes as surfels: StandardGaussianModel(sh_degree=0, all_the_same=True) after create_from_pcd with black
    colours (utils/time_utils.py:1248-1257, scene/gaussian_model.py:145-179,489-497).  Rows are re-allocated by the density control
    (at most a few ten thousand small rows, eager: none of the slot machinery of dgs_amd/densify.py is needed)."""
    ROWS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    max_sh_degree = 0
    active_sh_degree = 0
    packed_sh = False

    def __init__(self, points):
        super().__init__()
        pts = points.detach().float().clone()
        P = pts.shape[0]
        dist2 = torch.clamp_min(mean_nn_dist2(pts), 0.0000001)
        rot = torch.zeros(P, 4, device=pts.device)
        rot[:, 0] = 1
        o = 0.1 * torch.ones(P, 1, device=pts.device)
        self._xyz = nn.Parameter(pts)
        self._features_dc = nn.Parameter(torch.zeros(P, 1, 3, device=pts.device))
        self._features_rest = nn.Parameter(torch.zeros(P, 0, 3, device=pts.device))
        self._opacity = nn.Parameter(torch.log(o / (1 - o)))
        self._scaling = nn.Parameter(torch.log(torch.sqrt(dist2))[:, None].repeat(1, 2))
        self._rotation = nn.Parameter(rot)
        self.optimizer = None
        self._reset_statistics()

    def _reset_statistics(self):
        P, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)

    def row(self, name):
        return getattr(self, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                              "scaling": "_scaling", "rotation": "_rotation"}[name])

    def _set_row(self, name, value):
        setattr(self, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                       "scaling": "_scaling", "rotation": "_rotation"}[name], nn.Parameter(value.contiguous()))

    get_xyz = property(lambda self: self._xyz)
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))

    @property
    def get_scaling(self):
        return torch.exp(self._scaling.mean()[None, None].expand_as(self._scaling))

    def get_rotation_bias(self, rotation_bias=0.0):
        return torch.nn.functional.normalize(self._rotation + rotation_bias)

    @property
    def motion_mask(self):
        return torch.ones_like(self._xyz[..., :1])

    # ---- optimiser (GaussianModel.training_setup, scene/gaussian_model.py:181-203; no 'feature' group: fea_dim = 0) ----------------
    def training_setup(self, position_lr=0.00016, feature_lr=0.004, opacity_lr=0.05, scaling_lr=0.002, rotation_lr=0.002,
                       spatial_lr_scale=5.0):
        self._reset_statistics()
        rates = {"xyz": position_lr * spatial_lr_scale, "f_dc": feature_lr, "f_rest": feature_lr / 20.0, "opacity": opacity_lr,
                 "scaling": scaling_lr * spatial_lr_scale, "rotation": rotation_lr}
        self.optimizer = torch.optim.Adam([{"params": [self.row(n)], "lr": rates[n], "name": n} for n in self.ROWS], lr=0.0, eps=1e-15)

    def set_position_lr(self, lr):
        for grp in self.optimizer.param_groups:
            if grp["name"] == "xyz":
                grp["lr"] = lr

    def _moments(self, p):
        st = self.optimizer.state.get(p)
        return (st["exp_avg"], st["exp_avg_sq"], st["step"]) if st else (None, None, None)

    @torch.no_grad()
    def _reallocate(self, keep=None, extra=None):
        """Rows `keep` (bool mask, None = all) followed by the rows of `extra` (name -> tensor).  The Adam moments follow their rows,
        new rows start with zero moments, the step count stays (cat_tensors_to_optimizer / _prune_optimizer,
        scene/gaussian_model.py:327-387)."""
        for grp in self.optimizer.param_groups:
            name, old = grp["name"], grp["params"][0]
            m, v, step = self._moments(old)
            new = old.detach() if keep is None else old.detach()[keep]
            if extra is not None:
                new = torch.cat((new, extra[name]), dim=0)
            self._set_row(name, new)
            p = self.row(name)
            self.optimizer.state.pop(old, None)
            grp["params"] = [p]
            if m is not None:
                if keep is not None:
                    m, v = m[keep], v[keep]
                if extra is not None:
                    m, v = torch.cat((m, torch.zeros_like(extra[name])), dim=0), torch.cat((v, torch.zeros_like(extra[name])), dim=0)
                self.optimizer.state[p] = {"step": step, "exp_avg": m.contiguous(), "exp_avg_sq": v.contiguous()}

    @torch.no_grad()
    def add_densification_stats(self, viewspace_grad, visible):
        self.xyz_gradient_accum[visible] += torch.norm(viewspace_grad[visible, :2], dim=-1, keepdim=True)
        self.denom[visible] += 1

    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, draws, percent_dense=0.01, N=2):
        """GaussianModel.densify_and_prune (scene/gaussian_model.py:416-482) in the reference's order: clone, then split with the
        scale as it is AFTER the clones were added, then prune with the scale after the split.  Returns (cloned, split, pruned)."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        hot = torch.norm(grads, dim=-1) >= max_grad
        # clone
        sel = hot & (self.get_scaling.max(dim=1).values <= percent_dense * extent)
        n_clone = int(sel.sum())
        self._reallocate(extra={n: self.row(n).detach()[sel] for n in self.ROWS})
        self._reset_statistics()
        # split: the clones carry no gradient (padded with zeros)
        P = self._xyz.shape[0]
        padded = torch.zeros(P, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze(-1)
        sel = (padded >= max_grad) & (self.get_scaling.max(dim=1).values > percent_dense * extent)
        n_split = int(sel.sum())
        scale = self.get_scaling[sel].repeat(N, 1)
        std = torch.cat((scale, torch.zeros_like(scale[:, :1])), dim=-1)
        samples = draws.randn(*std.shape).to(std) * std
        R = _rotation_matrices(self._rotation[sel]).repeat(N, 1, 1)
        children = {n: self.row(n).detach()[sel].repeat(*((N,) + (1,) * (self.row(n).dim() - 1))) for n in self.ROWS}
        children["xyz"] = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + self._xyz[sel].repeat(N, 1)
        children["scaling"] = torch.log(scale / (0.8 * N))
        self._reallocate(extra=children)
        self._reset_statistics()
        parents = torch.cat((sel, torch.zeros(N * n_split, dtype=torch.bool, device=sel.device)))
        self._reallocate(keep=~parents)
        self._reset_statistics()
        # prune (max_radii2D was cleared above, so the screen-size criterion never fires: the reference's behaviour)
        prune = (self.get_opacity < min_opacity).squeeze(-1)
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        n_prune = int(prune.sum())
        self._reallocate(keep=~prune)
        self._reset_statistics()
        return n_clone, n_split, n_prune

    @torch.no_grad()
    def reset_opacity(self):
        """scene/gaussian_model.py:251-254: opacity <- min(opacity, 0.01), fresh moments."""
        o = torch.min(self.get_opacity, torch.ones_like(self._opacity) * 0.01)
        old = self._opacity
        m, v, step = self._moments(old)
        self._set_row("opacity", torch.log(o / (1 - o)))
        for grp in self.optimizer.param_groups:
            if grp["name"] == "opacity":
                grp["params"] = [self._opacity]
        self.optimizer.state.pop(old, None)
        if m is not None:
            self.optimizer.state[self._opacity] = {"step": step, "exp_avg": torch.zeros_like(m), "exp_avg_sq": torch.zeros_like(v)}


class NodePretrainer:
    # the reference's defaults (arguments/__init__.py:101-131)
    POSITION = (0.00016, 0.0000016, 30_000)      # x spatial_lr_scale 5
    DEFORM = (0.00016 * 5, 0.0000016, 40_000)

    def __init__(self, deform, cameras, targets, bg_color, points, extent, iterations=10_000, node_warm_up=2_000, sampling_at=7_500,
                 densify_interval=100, opacity_reset_interval=3_000, densify_grad_threshold=0.0002, densify_from=500,
                 white_background=False, lambda_dssim=0.2, arap=True, is_blender=True, node_max_num_ratio=16, draws=None,
                 rasterizer_cls=None, log=None, surfel_lrs=None, alpha_masks=None, mask_as_scene=False, mask_as_dynamic=False):
        """deform: dgs_amd.deform.ControlNodes (its node count is the number of control nodes the stage ends with).  points [N,3]:
        the scene's initial point cloud (the surfels' positions); the control nodes start as a farthest-point sample of it
        (GUI.__init__, train_gui.py:156-170 -> ControlNodeWarp.init, utils/time_utils.py:886-927).  extent: the cameras' extent.
        surfel_lrs: keyword arguments of NodeSurfels.training_setup (the reference's --feature_lr, --rotation_lr ... options).
        alpha_masks + mask_as_scene / mask_as_dynamic (train_gui.py:487,493-495): the node surfels are rendered over a fresh random
        background per step and the target is composited over the same one, so that only what lies inside the view's mask is fitted
        (the node surfels carry no motion mask of their own -- as_gs_force_with_motion_mask is off -- so no motion-mask term here)."""
        self.deform, self.cameras, self.targets, self.bg = deform, cameras, targets, bg_color
        self.extent = float(extent)
        self.iterations, self.node_warm_up, self.sampling_at = int(iterations), int(node_warm_up), int(sampling_at)
        self.densify_interval, self.opacity_reset_interval = int(densify_interval), int(opacity_reset_interval)
        self.densify_grad_threshold, self.densify_from = densify_grad_threshold, int(densify_from)
        self.white_background, self.lambda_dssim, self.arap, self.is_blender = bool(white_background), lambda_dssim, bool(arap), bool(is_blender)
        self.node_max_num_ratio = node_max_num_ratio
        self.draws = draws if draws is not None else Draws(0)
        self.rasterizer_cls, self.log = rasterizer_cls, log
        self.surfel_lrs = dict(surfel_lrs or {})
        self.alpha_masks = None if alpha_masks is None else list(alpha_masks)
        self.random_bg = bool(alpha_masks is not None and (mask_as_scene or mask_as_dynamic))
        self.iteration = 1
        self.stack = []
        self.losses = []
        self.history = []      # (iteration, cloned, split, pruned, node surfels) of every density-control call
        self.sampled = None    # indices (into the node surfels at that time) the control nodes were sampled from
        self._init_nodes(points.detach())
        groups = [{"params": list(deform.network.parameters()), "lr": self.DEFORM[0], "name": "deform"},
                  {"params": [deform.nodes, deform._node_radius, deform._node_weight], "lr": self.DEFORM[0], "name": "nodes"}]
        self.opt_deform = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    # ---- ControlNodeWarp.init (utils/time_utils.py:886-927) --------------------------------------------------------------------
    @torch.no_grad()
    def _init_nodes(self, pcl, hyper_pcl=None):
        """Control nodes <- farthest-point sample of pcl (of hyper_pcl if given: rows of the same points in another space); radius /
        weight parameters reset; the node surfels restart at the nodes.  In-place (`.data`), so the deformation optimiser keeps its
        state for these parameters, as in the reference.  Returns the sampled indices."""
        d = self.deform
        M = d.node_num
        if M > pcl.shape[0]:
            raise ValueError("fewer points (%d) than control nodes (%d): the reference replaces the node parameters without "
                             "telling their optimiser in this case; give fewer nodes or more points" % (pcl.shape[0], M))
        space = pcl if hyper_pcl is None else hyper_pcl
        idx = farthest_point_sample(space, M, start=self.draws.start(space.shape[0]))
        hyper = 1e-2 * torch.ones(M, d.hyper_dim, device=pcl.device, dtype=torch.float32)
        d.nodes.data = torch.cat([pcl[idx].float(), hyper], dim=-1)
        scene_range = pcl.max() - pcl.min()
        d._node_radius.data = torch.log(0.1 * scene_range + 1e-7) * torch.ones(M, device=pcl.device)
        d._node_weight.data = torch.zeros(M, 1, device=pcl.device)
        self.gs = NodeSurfels(d.nodes.data[:, :3])
        self.gs.training_setup(**self.surfel_lrs)
        self.tied = True      # the node surfels' positions ARE the control nodes' positions until the first re-allocation
        return idx

    def _next_view(self):
        if not self.stack:
            self.stack = list(range(len(self.cameras)))
        return self.stack.pop(self.draws.pick(len(self.stack)))

    def _regularisers(self, fid, time_interval):
        """train_gui.py:501-507 with the draws in the reference's order: elastic (jitter, 8 times), acceleration (jitter), ARAP
        (centre, 2 times)."""
        d, dev = self.deform, self.deform.nodes.device
        f = fid.reshape(-1)[0]
        dt = time_interval
        t0 = f + dt * (self.draws.rand([]).to(dev) - 0.5)
        t_samp = self.draws.rand(8).to(dev) * dt + t0 - 0.5 * dt
        loss = 1e-3 * reg.elastic_loss(d, t_samp=t_samp)
        dt = 3 * time_interval
        loss = loss + 1e-5 * reg.acc_loss(d, delta_t=dt, t0=f + dt * (self.draws.rand([]).to(dev) - 0.5))
        if self.arap:
            t0 = self.draws.rand([]).to(dev)
            t_samp = self.draws.rand(2).to(dev) * 0.05 + t0 - 0.5 * 0.05
            M = d.node_num
            sample_idx = self.draws.choice(M, 512).to(dev) if M > 512 else None
            loss = loss + 1e-2 * reg.arap_loss(d, t_samp=t_samp, sample_idx=sample_idx)
        return loss

    @torch.no_grad()
    def _sample_nodes(self):
        """'samp_hyper' (train_gui.py:552-575)."""
        old, d = self.gs, self.deform
        x = old.get_xyz.detach()
        t_samp = torch.linspace(0, 1, 16, device=x.device)
        traj = torch.stack([d.network(x, t_samp[i:i + 1, None].expand_as(x[..., :1]))["d_xyz"] * old.motion_mask for i in range(16)], dim=1)
        hyper_pcl = (traj + x[:, None]).reshape(x.shape[0], -1)
        idx = self._init_nodes(x, hyper_pcl)
        gs = self.gs
        for name in ("f_dc", "f_rest", "scaling", "opacity", "rotation"):
            gs._set_row(name, old.row(name).detach()[idx])
        gs.training_setup(**self.surfel_lrs)
        self.sampled = idx
        return idx

    def step(self):
        it, gs, d = self.iteration, self.gs, self.deform
        v = self._next_view()
        cam, gt = self.cameras[v], self.targets[v]
        time_interval = 1.0 / len(self.cameras)
        x = gs.get_xyz.detach()
        t = cam.fid.reshape(1, 1).expand(x.shape[0], 1)
        # (non-blender data adds annealed noise to the time input, train_gui.py:477 -- D-NeRF sets is_blender)
        d_xyz = d.network(x, t)["d_xyz"] * gs.motion_mask
        if it < self.node_warm_up:
            d_xyz = d_xyz.detach()
        random_bg = False
        if self.random_bg:
            random_bg = self.draws.rand(3).to(x.device)
        pkg = render(cam, gs, self.bg, d_xyz, 0.0, 0.0, rasterizer_cls=self.rasterizer_cls, random_bg_color=random_bg)
        image = pkg["render"]
        if self.random_bg:
            m = self.alpha_masks[v]
            gt = gt * m + pkg["bg_color"][:, None, None] * (1 - m)
        loss = (1.0 - self.lambda_dssim) * l1_loss(image, gt) + self.lambda_dssim * (1.0 - ssim(image, gt))
        if it > self.node_warm_up:
            loss = loss + self._regularisers(cam.fid, time_interval)
        loss.backward()
        self.losses.append(float(loss.detach()))
        sampled_now = it == self.sampling_at
        last = it == self.iterations - 1 and it > self.sampling_at
        with torch.no_grad():
            if it < self.sampling_at:
                gs.add_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"])
                if it % self.densify_interval == 0 or it == self.node_warm_up - 1:
                    size_threshold = 20 if it > self.opacity_reset_interval else None
                    grad_max = self.densify_grad_threshold
                    if not self.is_blender and gs.get_xyz.shape[0] > d.node_num * self.node_max_num_ratio:
                        grad_max = math.inf
                    counts = gs.densify_and_prune(grad_max, 0.005, self.extent, size_threshold, self.draws)
                    self.tied = False
                    self.history.append((it,) + counts + (gs.get_xyz.shape[0],))
                    if self.log:
                        self.log("[nodes %d] cloned %d, split %d, pruned %d -> %d node surfels" % self.history[-1])
                if it % self.opacity_reset_interval == 0 or (self.white_background and it == self.densify_from):
                    gs.reset_opacity()
            elif sampled_now:
                self._sample_nodes()
                gs = self.gs
                self.opt_deform.zero_grad()
                if self.log:
                    self.log("[nodes %d] %d control nodes sampled from %d node surfels" % (it, d.node_num, x.shape[0]))
            if last:
                d.nodes.data[:, :3] = gs._xyz.data
            if not sampled_now and not it == self.iterations - 1:
                gs.optimizer.step()
                gs.set_position_lr(expon_lr(it, self.POSITION[0] * 5.0, self.POSITION[1] * 5.0, self.POSITION[2]))
                gs.optimizer.zero_grad(set_to_none=True)
                # DeformModel.update_learning_rate returns after the FIRST group: the 'nodes' group keeps its initial rate
                self.opt_deform.param_groups[0]["lr"] = expon_lr(it, *self.DEFORM)
                self.opt_deform.step()
                self.opt_deform.zero_grad()
                if self.tied:
                    d.nodes.data[:, :3] = gs._xyz.data
        self.iteration += 1
        return self.losses[-1]

    def run(self, on_iteration=None):
        """All iterations of the stage (GUI.train, train_gui.py:207-213: while iteration_node_rendering < iterations_node_rendering)."""
        while self.iteration < self.iterations:
            self.step()
            if on_iteration is not None:
                on_iteration(self.iteration - 1, self)
        return self.losses
