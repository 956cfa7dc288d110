"""Dense, differentiable fp64 PyTorch re-derivation of the surfel rasterizer forward.

Purpose: pin the *mathematics* of the CPU oracle (oracle/surfel_oracle.c), in particular its
hand-derived backward, with torch.autograd -- the reference ships no tests or golden vectors
(SURVEY.md section 4).  This is written independently of the oracle's control flow: no tiles lists,
no sorting per tile; every surfel is evaluated against every pixel in global depth order and masked
by the tile rectangle test.  Only usable for tiny scenes.

Conventions deliberately matched to the reference's backward (cuda_rasterizer/backward.cu):
  * min(0.99, alpha) is a straight-through clamp (backward.cu:322,400 ignore it),
  * the 1/255 skip, the T<1e-4 stop, near-plane skip, culling and the dual-visible sign flip are
    treated as constants,
  * the quaternion is normalised without differentiating the norm (auxiliary.h:213-257).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_color(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def _rotmat(q):
    qn = q * (1.0 / q.norm(dim=-1, keepdim=True)).detach()
    w, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    # columns c0, c1, c2
    c0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], -1)
    c1 = torch.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], -1)
    c2 = torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], -1)
    return c0, c1, c2


def dense_render(means3D, scales, rotations, opacities, viewmatrix, campos, bg, tanfovx, tanfovy, H, W,
                 shs=None, sh_degree=0, colors_precomp=None):
    """Returns (color[3,H,W], allmap[8,H,W], radii[P], transMat[P,9] (retain_grad-able))."""
    dt = means3D.dtype
    P = means3D.shape[0]
    Wm = viewmatrix[:3, :3].T  # viewmatrix is W2C^T (row-vector convention) -> W2C rotation
    t = viewmatrix[3, :3]
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    cx, cy = W / 2.0, H / 2.0
    pv = means3D @ Wm.T + t
    c0, c1, c2 = _rotmat(rotations)
    M0 = (c0 * scales[:, 0:1]) @ Wm.T
    M1 = (c1 * scales[:, 1:2]) @ Wm.T
    tn = c2 @ Wm.T
    cosv = -(tn * pv).sum(-1)
    sign = torch.where(cosv > 0, torch.ones_like(cosv), -torch.ones_like(cosv)).detach()
    normal = tn * sign[:, None]
    Tu = torch.stack([fx * M0[:, 0] + cx * M0[:, 2], fx * M1[:, 0] + cx * M1[:, 2], fx * pv[:, 0] + cx * pv[:, 2]], -1)
    Tv = torch.stack([fy * M0[:, 1] + cy * M0[:, 2], fy * M1[:, 1] + cy * M1[:, 2], fy * pv[:, 1] + cy * pv[:, 2]], -1)
    Tw = torch.stack([M0[:, 2], M1[:, 2], pv[:, 2]], -1)
    transMat = torch.cat([Tu, Tv, Tw], -1)
    transMat.retain_grad()
    Tu, Tv, Tw = transMat[:, 0:3], transMat[:, 3:6], transMat[:, 6:9]
    sgn = torch.tensor([1.0, 1.0, -1.0], dtype=dt)
    d = (sgn * Tw * Tw).sum(-1)
    f = sgn[None, :] / d[:, None]
    center = torch.stack([(f * Tu * Tw).sum(-1), (f * Tv * Tw).sum(-1)], -1)
    h0 = center * center - torch.stack([(f * Tu * Tu).sum(-1), (f * Tv * Tv).sum(-1)], -1)
    ext = torch.sqrt(torch.clamp_min(h0, 0.0)).detach()
    radius = torch.ceil(3.0 * torch.clamp_min(ext.max(dim=-1).values, 0.7071067811865476))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    cd = center.detach()
    ri = radius.to(torch.int64).to(dt)
    rminx = torch.clamp(torch.trunc((cd[:, 0] - ri) / 16), 0, gx)
    rminy = torch.clamp(torch.trunc((cd[:, 1] - ri) / 16), 0, gy)
    rmaxx = torch.clamp(torch.trunc((cd[:, 0] + ri + 15) / 16), 0, gx)
    rmaxy = torch.clamp(torch.trunc((cd[:, 1] + ri + 15) / 16), 0, gy)
    visible = (pv[:, 2] > 0.2) & (cosv != 0) & (d != 0) & ((rmaxx - rminx) * (rmaxy - rminy) != 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    if colors_precomp is None:
        dirs = means3D - campos[None, :]
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        colors = _sh_color(sh_degree, shs, dirs)
    else:
        colors = colors_precomp

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pfx = (xs.to(dt) + 0.5).reshape(-1)
    pfy = (ys.to(dt) + 0.5).reshape(-1)
    tix = (xs // 16).reshape(-1).to(dt)
    tiy = (ys // 16).reshape(-1).to(dt)
    N = H * W
    T = torch.ones(N, dtype=dt)
    done = torch.zeros(N, dtype=torch.bool)
    C = torch.zeros(N, 3, dtype=dt)
    Nrm = torch.zeros(N, 3, dtype=dt)
    D = torch.zeros(N, dtype=dt)
    dist1 = torch.zeros(N, dtype=dt)
    dist2 = torch.zeros(N, dtype=dt)
    distortion = torch.zeros(N, dtype=dt)
    med_d = torch.zeros(N, dtype=dt)
    med_w = torch.zeros(N, dtype=dt)

    depth_key = pv[:, 2].detach()
    order = sorted([i for i in range(P) if bool(visible[i])], key=lambda i: (float(depth_key[i]), i))
    for i in order:
        in_rect = (tix >= rminx[i]) & (tix < rmaxx[i]) & (tiy >= rminy[i]) & (tiy < rmaxy[i])
        k = -Tu[i][None, :] + pfx[:, None] * Tw[i][None, :]
        l = -Tv[i][None, :] + pfy[:, None] * Tw[i][None, :]
        p = torch.cross(k, l, dim=-1)
        okz = p[:, 2] != 0
        pz = torch.where(okz, p[:, 2], torch.ones_like(p[:, 2]))
        sx, sy = p[:, 0] / pz, p[:, 1] / pz
        rho3d = sx * sx + sy * sy
        dx, dy = center[i, 0] - pfx, center[i, 1] - pfy
        rho2d = 2.0 * (dx * dx + dy * dy)
        use3d = rho3d <= rho2d
        rho = torch.where(use3d, rho3d, rho2d)
        depth = torch.where(use3d, sx * Tw[i, 0] + sy * Tw[i, 1] + Tw[i, 2], Tw[i, 2].expand(N))
        G = torch.exp(-0.5 * rho)
        a_raw = opacities[i] * G
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()
        cand = in_rect & okz & (~done) & (depth >= 0.2) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        stop = cand & (test_T < 1e-4)
        done = done | stop
        con = cand & (~stop)
        cf = con.to(dt)
        w = alpha * T
        m = (100.0 * depth - 20.0) / (99.8 * depth)
        A = 1 - T
        err = m * m * A + dist2 - 2 * m * dist1
        distortion = distortion + cf * err * w
        is_med = con & (T > 0.5)
        med_d = torch.where(is_med, depth, med_d)
        med_w = torch.where(is_med, w, med_w)
        Nrm = Nrm + (cf * w)[:, None] * normal[i][None, :]
        D = D + cf * depth * w
        dist1 = dist1 + cf * m * w
        dist2 = dist2 + cf * m * m * w
        C = C + (cf * w)[:, None] * colors[i][None, :]
        T = torch.where(con, test_T, T)

    color = (C + T[:, None] * bg[None, :]).T.reshape(3, H, W)
    allmap = torch.stack([D, 1 - T, Nrm[:, 0], Nrm[:, 1], Nrm[:, 2], med_d, distortion, med_w], 0).reshape(8, H, W)
    return color, allmap, radii, transMat
