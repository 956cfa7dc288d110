"""Image losses of the reference train step (utils/loss_utils.py:19-76, train_gui.py:292-313), PyTorch.

``ssim`` evaluates the same 11x11 sigma=1.5 Gaussian-window SSIM; the window is applied as two 1-D
passes (the 2-D window of loss_utils.py:38-42 is exactly the outer product of the 1-D one), which is the
same arithmetic up to fp32 rounding and 5.5x fewer taps.
"""
import math

import torch
import torch.nn.functional as F

_WINDOWS = {}
DEBUG_TERMS = None


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def _window_1d(size, sigma, device, dtype):
    key = (size, sigma, str(device), dtype)
    if key not in _WINDOWS:
        g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
        _WINDOWS[key] = (g / g.sum()).to(device=device, dtype=dtype)
    return _WINDOWS[key]


def _blur(x, w1d, channel):
    """Depthwise separable Gaussian blur with zero padding, x: [B,C,H,W]."""
    k = w1d.numel()
    wh = w1d.view(1, 1, 1, k).expand(channel, 1, 1, k)
    wv = w1d.view(1, 1, k, 1).expand(channel, 1, k, 1)
    x = F.conv2d(x, wh, padding=(0, k // 2), groups=channel)
    return F.conv2d(x, wv, padding=(k // 2, 0), groups=channel)


def ssim(img1, img2, window_size=11):
    """Mean SSIM of two [C,H,W] (or [B,C,H,W]) images.  HIP tensors go through the fused gfx950 kernel
    (csrc/train_ops.hip), CPU tensors through the PyTorch formulation below (same arithmetic)."""
    if img1.is_cuda and img1.dim() == 3 and window_size == 11 and img1.dtype == torch.float32:
        from . import _ops
        return _ops.fused_ssim(img1, img2)
    return ssim_torch(img1, img2, window_size)


def ssim_torch(img1, img2, window_size=11):
    if img1.dim() == 3:
        img1, img2 = img1[None], img2[None]
    C = img1.size(-3)
    w = _window_1d(window_size, 1.5, img1.device, img1.dtype)
    # one batched blur over the five maps instead of five grouped convs
    stack = torch.cat([img1, img2, img1 * img1, img2 * img2, img1 * img2], dim=1)
    b = _blur(stack, w, 5 * C)
    mu1, mu2, s11, s22, s12 = torch.split(b, C, dim=1)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq, sigma2_sq, sigma12 = s11 - mu1_sq, s22 - mu2_sq, s12 - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def training_loss(pkg, gt_image, lambda_dssim=0.2, lambda_normal=0.02, lambda_dist=1000.0):
    """train_gui.py:292-313 in the iteration > 8000 regime (all eight allmap channels receive gradient)."""
    image = pkg["render"]
    normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))[None]
    normal_loss = lambda_normal * normal_error.mean()
    dist_loss = lambda_dist * pkg["rend_dist"].mean()
    ll1 = l1_loss(image, gt_image)
    loss_img = (1.0 - lambda_dssim) * ll1 + lambda_dssim * (1.0 - ssim(image, gt_image))
    if DEBUG_TERMS is not None:  # development aid: static buffers that receive the individual terms
        for k, v in (("l1", ll1), ("ssim", loss_img * 0 + (1.0 - (loss_img - (1.0 - lambda_dssim) * ll1) / lambda_dssim)),
                     ("normal", normal_loss), ("dist", dist_loss), ("img_mean", image.mean()), ("gt_mean", gt_image.mean())):
            DEBUG_TERMS.setdefault(k, torch.zeros((), device=image.device)).copy_(v.detach())
    return loss_img + normal_loss + dist_loss


FUSE_PHOTOMETRIC = True  # L1 + D-SSIM + regularisers as one autograd node (csrc/train_ops.hip); False: separate ops


def training_loss_from_allmap(image, allmap, cam, gt_image, lambda_dssim=0.2, lambda_normal=0.02, lambda_dist=1000.0, unit_grad=False, guard=None):
    """Same value and gradients as training_loss(render(...), gt) but computed from the rasterizer outputs directly:
    on a HIP device the allmap post-processing (normal rotation, depth -> points -> normals) and the two regularisers
    are one fused kernel per direction (csrc/train_ops.hip) instead of ~100 elementwise launches."""
    from . import _ops
    from .render import camera_rays
    rays_d, rays_o = camera_rays(cam, image.device)
    if FUSE_PHOTOMETRIC:
        return _ops.fused_train_loss(image, allmap, gt_image, rays_d, rays_o, cam.world_view_transform, lambda_dssim, lambda_normal,
                                     lambda_dist, slots=getattr(cam, "slots", None), unit_grad=unit_grad, guard=guard)
    reg = _ops.fused_reg_loss(allmap, rays_d, rays_o, cam.world_view_transform, lambda_normal, lambda_dist)
    ll1 = l1_loss(image, gt_image)
    return (1.0 - lambda_dssim) * ll1 + lambda_dssim * (1.0 - ssim(image, gt_image)) + reg
