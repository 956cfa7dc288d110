"""TEST-ONLY stand-in for the HIP operator on machines without a GPU: the same GaussianRasterizer call
shape, executed by the CPU oracle.  Lets the CPU suite exercise dgs_amd.render / losses / Trainer / the
gloo data-parallel path.  Never imported by the product package."""
import numpy as np
import torch
import torch.nn as nn

from oracle.surfel_oracle import OracleRaster


class _OracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, cfg, colors=None):
        ctx.precomp = colors is not None
        orc = OracleRaster(
            means3D=means3D.detach().numpy(), opacities=opacities.detach().numpy(), scales=scales.detach().numpy(),
            rotations=rotations.detach().numpy(), viewmatrix=cfg.viewmatrix.numpy(), projmatrix=cfg.projmatrix.numpy(),
            campos=cfg.campos.numpy(), bg=cfg.bg.numpy(), tanfovx=cfg.tanfovx, tanfovy=cfg.tanfovy,
            image_height=cfg.image_height, image_width=cfg.image_width, shs=None if ctx.precomp else sh.detach().numpy(),
            colors_precomp=colors.detach().numpy() if ctx.precomp else None, sh_degree=cfg.sh_degree)
        ctx.orc = orc
        radii = torch.from_numpy(orc.radii.copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(orc.color.copy()), radii, torch.from_numpy(orc.allmap.copy())

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_allmap):
        H, W = ctx.orc.H, ctx.orc.W
        gc = np.zeros((3, H, W), np.float32) if g_color is None else g_color.contiguous().numpy()
        go = np.zeros((8, H, W), np.float32) if g_allmap is None else g_allmap.contiguous().numpy()
        g = ctx.orc.backward(gc, go)
        t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))
        return (t("dL_dmeans3D"), t("dL_dmeans2D"), None if ctx.precomp else t("dL_dsh"), t("dL_dopacity"), t("dL_dscales"), t("dL_drotations"), None,
                t("dL_dcolors") if ctx.precomp else None)


class OracleRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        assert cov3D_precomp is None and (shs is None) != (colors_precomp is None)
        return _OracleFn.apply(means3D, means2D, shs, opacities, scales, rotations, self.raster_settings, colors_precomp)
