"""MI355X-native ``diff_surfel_rasterization``: the operator surface of the reference's 2D-Gaussian
surfel rasterizer, backed by hand-written gfx950 kernels.

Drop-in for submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py of
hustvl/Dynamic-2DGS: ``gaussian_renderer.render()`` (gaussian_renderer/__init__.py:14,61-76,141-150)
imports ``GaussianRasterizationSettings`` and ``GaussianRasterizer`` from this package name and calls
them unchanged.  Contract kept from the reference (file:line in the reference package):

* ``GaussianRasterizationSettings``: 12-field NamedTuple, same names and order (:158-170).
* ``GaussianRasterizer(settings)(means3D, means2D, opacities, shs|colors_precomp, scales, rotations)``
  -> ``(color[3,H,W], radii[P] int32, allmap[8,H,W])`` (:188-222); ``markVisible(positions)`` (:177-186).
* gradient order ``(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
  None)`` (:144-156); ``means2D`` is a carrier whose ``.grad[:, :2]`` holds the densification signal.
* same exception text for the exactly-one-of argument checks (:192-196); ``debug=True`` writes
  ``snapshot_fw.dump`` / ``snapshot_bw.dump`` with the CPU copy of the arguments before re-raising
  (:83-90,133-140).

Documented deviation: ``cov3D_precomp`` (a [P,9] transMat in the reference, whose backward dereferences
null scales/rotations) is rejected with an explicit error; see DESIGN.md.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "Lane"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU copy of an argument tuple (tensors cloned) for the debug dumps."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _guarded(fn, args, debug, dump_name, what):
    if not debug:
        return fn(*args)
    saved = _snapshot(args)  # before anything can be corrupted
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print("\nAn error occured in %s. Writing %s for debugging.\n" % (what, dump_name))
        raise


import threading

_SINKS = {}                   # (device index, data_ptr of the shs tensor or None) -> (sink tensor, all_rows)
_SINKS_LOCK = threading.Lock()
_BWD_LOCKS = {}               # device index -> lock around (option 8, backward launch): the option lives in the device's default context


def set_sh_grad_sink(tensor, device=None, all_rows=False, shs=None):
    """Extension (not in the reference): while set to a contiguous fp32 [P,M,3] tensor, backward passes write dL/dSH of the visible
    surfels directly into it (the kernel stores, it does not accumulate; rows of culled surfels are left as they are) and return no
    gradient for `shs`.  For trainers that keep gradients in one flat buffer.

    shs=<tensor>: the sink belongs to THAT `shs` input -- only backward passes of forwards that were given this tensor use it.  This is
    the form an owner of parameters uses (dgs_amd.train.Trainer does): any number of trainers can share a device, each with its own
    sink, in any interleaving.  Without `shs` the sink applies to every backward on the tensor's device that has no sink of its own
    (one per device; the round-2 form).  None removes: the sink of `shs`, or the device-wide one of `device`, or -- neither given --
    every sink.  The state cannot be per thread: autograd runs the backward of a HIP device on that device's own worker thread, not
    on the thread that called backward().

    all_rows (dgs_set_option key 8, set around each backward that uses this sink): the backward also stores ZEROS in the rows of
    culled surfels and the unused SH bands, i.e. every element of the sink is written by every backward and the buffer never needs
    clearing."""
    with _SINKS_LOCK:
        if tensor is not None:
            _SINKS[(tensor.device.index, None if shs is None else shs.data_ptr())] = (tensor, bool(all_rows))
        elif shs is not None:
            _SINKS.pop((shs.device.index, shs.data_ptr()), None)
        elif device is not None:
            _SINKS.pop((torch.device(device).index, None), None)
        else:
            _SINKS.clear()


def _sink_for(dev, sh):
    with _SINKS_LOCK:
        hit = _SINKS.get((dev.index, sh.data_ptr())) if sh is not None and sh.numel() else None
        return hit if hit is not None else _SINKS.get((dev.index, None), (None, False))


def _bwd_lock(dev):
    with _SINKS_LOCK:
        lk = _BWD_LOCKS.get(dev.index)
        if lk is None:
            lk = _BWD_LOCKS[dev.index] = threading.Lock()
        return lk


class Lane:
    """Extension (not in the reference): everything that lets SEVERAL renders of one process be in flight on one device at the same
    time, each on its own stream -- an explicit library context (its own options, overflow flag, staging words: `_C.Context`) and
    the lane's own dL/dSH sink.  `GaussianRasterizer(settings, lane=lane)` runs forward and backward in that context; without a lane
    the operator uses the device's default context and the sinks registered with set_sh_grad_sink, as before.
    dgs_amd.train.Trainer(concurrent_views=True) gives every view of a multi-view step its lane."""

    def __init__(self, device=None):
        self.context = _C.Context(device)
        self.sink = None          # contiguous fp32 [P,M,3] tensor the backward stores dL/dSH into (see set_sh_grad_sink), or None
        self.all_rows = False     # the backward also stores zeros in the rows of culled surfels (option 8 of the lane's context)


class _SurfelRasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, cfg, lane=None):
        call = (cfg.bg, means3D, colors_precomp, opacities, scales, rotations, cfg.scale_modifier, cov3Ds_precomp,
                cfg.viewmatrix, cfg.projmatrix, cfg.tanfovx, cfg.tanfovy, cfg.image_height, cfg.image_width, sh,
                cfg.sh_degree, cfg.campos, cfg.prefiltered, cfg.debug)
        fwd = _C.rasterize_gaussians if lane is None else (lambda *a: _C.rasterize_gaussians(*a, context=lane.context))
        n_rendered, color, allmap, radii, geom, binning, img = _guarded(fwd, call, cfg.debug, "snapshot_fw.dump", "forward")
        ctx.cfg = cfg
        ctx.lane = lane
        ctx.n_rendered = n_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # otherwise autograd zero-fills a [P] int32 "gradient" of radii on every backward
        return color, radii, allmap

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_allmap):
        cfg = ctx.cfg
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if g_color is None:
            g_color = torch.zeros((3, cfg.image_height, cfg.image_width), dtype=torch.float32, device=means3D.device)
        if g_allmap is None:
            g_allmap = torch.zeros((8, cfg.image_height, cfg.image_width), dtype=torch.float32, device=means3D.device)
        call = (cfg.bg, means3D, radii, colors_precomp, scales, rotations, cfg.scale_modifier, cov3Ds_precomp,
                cfg.viewmatrix, cfg.projmatrix, cfg.tanfovx, cfg.tanfovy, g_color, g_allmap, sh, cfg.sh_degree, cfg.campos,
                geom, ctx.n_rendered, binning, img, cfg.debug)
        lane = ctx.lane
        sink, all_rows = _sink_for(means3D.device, sh) if lane is None else (lane.sink, lane.all_rows)
        if sink is not None and not (sink.shape == sh.shape and sink.dtype == torch.float32 and sink.is_contiguous()
                                     and sink.device == sh.device):
            raise RuntimeError("set_sh_grad_sink: the sink must be a contiguous fp32 tensor of the shape of shs")
        if lane is not None:
            # the lane's context is its own: option 8 is set on it, no device-wide lock (other lanes' backwards run next to this one)
            lane.context.set_option(8, 1 if (sink is not None and all_rows) else 0)
            want = dict(want_colors=bool(ctx.needs_input_grad[3]), want_transmat=bool(ctx.needs_input_grad[7]), context=lane.context)
            if sink is not None:
                want["dL_dsh_out"] = sink
            (g_means2D, g_colors, g_opac, g_means3D, g_transMat, g_sh, g_scales, g_rot) = _guarded(
                lambda *a: _C.rasterize_gaussians_backward(*a, **want), call, cfg.debug, "snapshot_bw.dump", "backward")
            return (g_means3D, g_means2D, None if sink is not None else g_sh, g_colors, g_opac, g_scales, g_rot, g_transMat, None, None)
        # option 8 (zeros for culled rows) belongs to the sink of THIS call; it is a switch of the device's default context, so it is
        # set and the launches are issued under one lock per device (the kernels read it at launch time, on the host)
        with _bwd_lock(means3D.device):
            _C.set_option(8, 1 if (sink is not None and all_rows) else 0, device=means3D.device.index)
            # arrays nobody differentiates are not computed: dL_dcolors is the gradient of colors_precomp, dL_dtransMat of cov3Ds_precomp
            want = dict(want_colors=bool(ctx.needs_input_grad[3]), want_transmat=bool(ctx.needs_input_grad[7]))
            if sink is not None:
                (g_means2D, g_colors, g_opac, g_means3D, g_transMat, g_sh, g_scales, g_rot) = _guarded(
                    lambda *a: _C.rasterize_gaussians_backward(*a, dL_dsh_out=sink, **want), call, cfg.debug, "snapshot_bw.dump", "backward")
                g_sh = None
            else:
                (g_means2D, g_colors, g_opac, g_means3D, g_transMat, g_sh, g_scales, g_rot) = _guarded(
                    lambda *a: _C.rasterize_gaussians_backward(*a, **want), call, cfg.debug, "snapshot_bw.dump", "backward")
        return (g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rot, g_transMat, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, lane=None):
    return _SurfelRasterFn.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                 raster_settings, lane)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings, lane=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.lane = lane   # extension: a `Lane` (own library context + dL/dSH sink) for renders that run concurrently on one device

    def markVisible(self, positions):
        """bool[P]: surfel centres in front of the near plane of this camera."""
        cfg = self.raster_settings
        with torch.no_grad():
            return _C.mark_visible(positions, cfg.viewmatrix, cfg.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        cfg = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if cov3D_precomp is not None:
            raise NotImplementedError(
                "cov3D_precomp (precomputed transMat) is not supported by the MI355X rasterizer; pass scales and rotations")

        def empty():
            return torch.empty(0, dtype=torch.float32, device=means3D.device)

        shs = empty() if shs is None else shs
        colors_precomp = empty() if colors_precomp is None else colors_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, empty(), cfg, self.lane)
