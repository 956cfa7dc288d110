"""render(): the caller of the rasterizer, restating gaussian_renderer/__init__.py:41-219 of the reference
for the path its trainer uses (SH colours, scales + rotations, depth_ratio = 1).  Returns the same dict keys.
The reference's render() itself also runs unchanged against ``diff_surfel_rasterization``;
this module exists because the reference's Python cannot travel to the GPU box."""
import math

import torch

from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_RAYS = {}


def _camera_rays(cam, device):
    """Per-camera constants of utils/point_utils.py:9-25 (cached: the reference rebuilds them per call)."""
    if getattr(cam, "rays_d", None) is not None:  # static-buffer camera of the graph-captured train step
        return cam.rays_d, cam.rays_o
    key = (id(cam), str(device))
    hit = _RAYS.get(key)
    if hit is not None and hit[0] is cam:
        return hit[1], hit[2]
    W, H = cam.image_width, cam.image_height
    c2w = (cam.world_view_transform.T).inverse()
    fx = W / (2 * math.tan(cam.FoVx / 2.))
    fy = H / (2 * math.tan(cam.FoVy / 2.))
    intrins = torch.tensor([[fx, 0., W / 2.], [0., fy, H / 2.], [0., 0., 1.0]], dtype=torch.float32, device=device)
    gx, gy = torch.meshgrid(torch.arange(W, device=device), torch.arange(H, device=device), indexing='xy')
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3).float()
    rays_d = pts @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    if len(_RAYS) > 256:
        _RAYS.clear()
    _RAYS[key] = (cam, rays_d, rays_o)
    return rays_d, rays_o


def camera_rays(cam, device):
    """(rays_d [H*W,3], rays_o [3]) of a camera: the per-view constants of depth_to_normal."""
    return _camera_rays(cam, device)


def depth_to_normal(cam, depth):
    """utils/point_utils.py:27-38: finite-difference normals of the back-projected depth map."""
    rays_d, rays_o = _camera_rays(cam, depth.device)
    points = (depth.reshape(-1, 1) * rays_d + rays_o).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return output, points


def render(cam, pc, bg_color, d_xyz=0.0, d_rotation=0.0, d_scaling=0.0, debug=False, rasterizer_cls=None, postprocess=True,
           assembled=None, d_opacity=None, d_color=None, random_bg_color=False, render_motion=False, detach_xyz=False,
           detach_scale=False, detach_rot=False, detach_opacity=False):
    """pc: dgs_amd.model.SurfelModel.  d_*: outputs of the deformation (or 0.0).  assembled: (means3D, scales, rotations,
    opacity) already computed by ControlNodes.forward_assembled (then d_xyz / d_rotation / d_scaling / d_opacity are ignored).
    d_opacity [P,1], d_color [P,3] (None in the reference's default configuration: pred_opacity / pred_color are off) and
    random_bg_color follow gaussian_renderer/__init__.py:41,58,82-85,114: opacity + d_opacity, d_color added to the DC coefficient,
    a fresh uniform background that the returned dict reports as 'bg_color' (the trainer composites its target over it).
    render_motion (:103-107): instead of the SH colours, the precomputed colour (motion_mask, 0, 1 - motion_mask) -- channel 0 of the
    image is then the rendered motion mask (train_gui.py:365-369); detach_* (:127-137) cut the geometry out of that render's graph."""
    if torch.is_tensor(random_bg_color):     # (a caller that draws the background itself: replayed runs)
        bg_color = random_bg_color
    elif random_bg_color:
        bg_color = torch.rand_like(bg_color)
    xyz = pc.get_xyz
    # leaf that only receives dL/dmeans2D (its values are never read): one persistent tensor per model instead of a
    # zero-filled [P,3] allocation per render
    screenspace_points = None if render_motion else getattr(pc, "_screenspace_leaf", None)   # (a motion render next to the colour render of a step keeps its screen-space gradients to itself)
    if render_motion:
        screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    elif screenspace_points is None or screenspace_points.shape != xyz.shape or screenspace_points.device != xyz.device:
        screenspace_points = torch.zeros_like(xyz, requires_grad=True)
        try:
            object.__setattr__(pc, "_screenspace_leaf", screenspace_points)
        except Exception:
            pass
    screenspace_points.grad = None
    cfg = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=cam.camera_center, prefiltered=False, debug=debug)
    rasterizer = (rasterizer_cls or GaussianRasterizer)(raster_settings=cfg)
    if assembled is not None:
        means3D, scales, rotations, opacity = assembled
    else:
        means3D = xyz + d_xyz
        scales = pc.get_scaling + d_scaling
        rotations = pc.get_rotation_bias(d_rotation)
        opacity = pc.get_opacity if d_opacity is None else pc.get_opacity + d_opacity
    shs = colors_precomp = None
    if render_motion:
        m = pc.motion_mask
        colors_precomp = torch.cat((m, torch.zeros_like(m), 1 - m), dim=-1)
    else:
        shs = pc.get_features
        if d_color is not None and not isinstance(d_color, float):
            shs = torch.cat([shs[:, :1] + d_color[:, None], shs[:, 1:]], dim=1)
    if detach_xyz:
        means3D = means3D.detach()
    if detach_rot:
        rotations = rotations.detach()
    if detach_scale:
        scales = scales.detach()
    if detach_opacity:
        opacity = opacity.detach()
    rendered_image, radii, allmap = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=None)
    # 'bg_color' on both paths: with random_bg_color the caller composites its target over the SAME background (gaussian_renderer/__init__.py:41,114)
    rets = {"render": rendered_image, "viewspace_points": screenspace_points, "radii": radii, "allmap": allmap, "bg_color": bg_color}
    if not postprocess:  # the fused loss / statistics kernels work on the rasterizer outputs directly (radii > 0 is the filter)
        return rets
    rets["visibility_filter"] = radii > 0
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal = (render_normal.permute(1, 2, 0) @ (cam.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    depth_ratio = 1  # hard-wired in the reference (gaussian_renderer/__init__.py:196)
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    surf_normal, surf_point = depth_to_normal(cam, surf_depth)
    surf_normal = surf_normal.permute(2, 0, 1) * render_alpha.detach()
    rets.update({'alpha': render_alpha, 'rend_normal': render_normal, 'rend_dist': render_dist, 'depth': surf_depth,
                 'surf_normal': surf_normal, 'surf_point': surf_point.permute(2, 0, 1), 'bg_color': bg_color})
    return rets
