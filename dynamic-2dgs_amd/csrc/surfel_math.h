// surfel_math.h -- per-surfel and per-(pixel,surfel) arithmetic of the MI355X surfel rasterizer.
//
// Written from scratch for gfx950; the behaviour it reproduces is cited per function against
// the reference (paths relative to submodules/diff-surfel-rasterization/ of hustvl/Dynamic-2DGS).
// Every function is __host__ __device__ so that tests/ can compile this header with g++ and check
// the arithmetic against the CPU oracle without a GPU (tests/hostmath/); the shipped library only
// ever calls it from HIP kernels.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DGS_HD __host__ __device__ __forceinline__
#else
#define DGS_HD inline
#endif

namespace dgs {

// ---- compile-time constants of the reference (cuda_rasterizer/config.h:15-17, auxiliary.h:18-37)
constexpr int kTileX = 16;
constexpr int kTileY = 16;
constexpr int kTilePix = 256;
constexpr float kFilterSize = 0.70710678118654752f;
constexpr float kNear = 0.2f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTmin = 0.0001f;

// ---- packed per-surfel record written by the preprocess kernel and gathered by the blend kernels.
// 24 floats = 96 B = six 16-B chunks, so one surfel is 6 dwordx4 transactions:
//   q0 = (Tu.x Tu.y Tu.z Tv.x)  q1 = (Tv.y Tv.z Tw.x Tw.y)  q2 = (Tw.z xy.x xy.y opacity)   <- alpha part
//   q3 = (n.x n.y n.z r)        q4 = (g b depth flags)                                      <- shading part
//   q5 = (x_lo x_hi y_lo y_hi)  exact pixel bounding box of {alpha >= 1/255} (see tight_tile_rect) <- culling part
constexpr int kRecFloats = 24;
constexpr int kRecQuads = 6;

struct Quad { float x, y, z, w; };

struct SurfelRec {
    float Tu[3], Tv[3], Tw[3];
    float xy[2];
    float opacity;
    float normal[3];
    float rgb[3];
    float depth;
    uint32_t flags;  // bit c set: colour channel c was clamped at 0 (forward.cu:66-69)
    float bbox[4];   // x_lo, x_hi, y_lo, y_hi in pixel-centre coordinates; (-inf, +inf) when not applicable
};
static_assert(sizeof(SurfelRec) == kRecFloats * 4, "record must be 96 bytes");

// ---- per-surfel gradient accumulator slots written by the backward blend (one 80-B row per surfel)
constexpr int kAccFloats = 20;
enum AccSlot { kAccColor = 0, kAccNormal = 3, kAccT = 6, kAccOpacity = 15, kAccMean2D = 16 };

DGS_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DGS_PRECISE_MATH)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

DGS_HD float fast_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DGS_PRECISE_MATH)
    return __expf(x);
#else
    return expf(x);
#endif
}

struct Camera {
    const float* view;    // [16] world_view_transform flattened: W2C column-major, translation at [12..14]
    const float* campos;  // [3]   (both are read where the kernel runs: device pointers in the library)
    float focal_x, focal_y;
    float tan_fovx, tan_fovy;
    int width, height;
    int tiles_x, tiles_y;
};

// ---- per-surfel forward: FMA contraction off for this whole section, so that culling / ceil / (int)
// decisions and the stored records are bit-identical to the CPU oracle (gcc -ffp-contract=off).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// auxiliary.h:64-74 getRect: tile rectangle [min,max) touched by a disc of integer radius.
DGS_HD void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    int v;
    v = (int)((px - radius) / kTileX); v = v < 0 ? 0 : v; x0 = v < gx ? v : gx;
    v = (int)((py - radius) / kTileY); v = v < 0 ? 0 : v; y0 = v < gy ? v : gy;
    v = (int)((px + radius + kTileX - 1) / kTileX); v = v < 0 ? 0 : v; x1 = v < gx ? v : gx;
    v = (int)((py + radius + kTileY - 1) / kTileY); v = v < 0 ? 0 : v; y1 = v < gy ? v : gy;
}

// auxiliary.h:188-210: rotation matrix columns from a (r,x,y,z) quaternion, normalised on the fly.
DGS_HD void quat_columns(const float* q, float c0[3], float c1[3], float c2[3])
{
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    c0[0] = 1.f - 2.f * (y * y + z * z); c0[1] = 2.f * (x * y + w * z);       c0[2] = 2.f * (x * z - w * y);
    c1[0] = 2.f * (x * y - w * z);       c1[1] = 1.f - 2.f * (x * x + z * z); c1[2] = 2.f * (y * z + w * x);
    c2[0] = 2.f * (x * z + w * y);       c2[1] = 2.f * (y * z - w * x);       c2[2] = 1.f - 2.f * (x * x + y * y);
}

DGS_HD void view_rotate(const float* vm, const float* v, float* r)  // W * v, forward.cu:79-83
{
    r[0] = vm[0] * v[0] + vm[4] * v[1] + vm[8] * v[2];
    r[1] = vm[1] * v[0] + vm[5] * v[1] + vm[9] * v[2];
    r[2] = vm[2] * v[0] + vm[6] * v[1] + vm[10] * v[2];
}

DGS_HD void view_rotate_T(const float* vm, const float* v, float* r)  // W^T * v, auxiliary.h:108-116
{
    r[0] = vm[0] * v[0] + vm[1] * v[1] + vm[2] * v[2];
    r[1] = vm[4] * v[0] + vm[5] * v[1] + vm[6] * v[2];
    r[2] = vm[8] * v[0] + vm[9] * v[1] + vm[10] * v[2];
}

// forward.cu:20-71 computeColorFromSH.  `sh` points at this surfel's [M,3] coefficients.
DGS_HD uint32_t sh_to_rgb(int deg, const float* sh, const float* pos, const float* campos, float rgb[3])
{
    float dx = pos[0] - campos[0], dy = pos[1] - campos[1], dz = pos[2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    uint32_t flags = 0;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    for (int c = 0; c < 3; c++) {
        float r = C0 * sh[c];
        if (deg > 0) {
            r = r - C1 * y * sh[3 + c] + C1 * z * sh[6 + c] - C1 * x * sh[9 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + C2[0] * xy * sh[12 + c] + C2[1] * yz * sh[15 + c] + C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
                    C2[3] * xz * sh[21 + c] + C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + C3[1] * xy * z * sh[30 + c] +
                        C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] + C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                        C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + C3[5] * z * (xx - yy) * sh[42 + c] +
                        C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        r += 0.5f;
        if (r < 0.f) flags |= (1u << c);
        rgb[c] = r > 0.f ? r : 0.f;
    }
    return flags;
}

// Exact, opacity-aware tile rectangle (NOT in the reference): a (surfel, tile) pair can only produce
// alpha >= 1/255 (forward.cu:397-399) if  rho = min(rho3d, rho2d) <= tau = 2 ln(255 o).  The region
// {rho3d <= tau} is the projection of the splat-space disc of radius sqrt(tau): its exact screen bounding box
// follows from the same bounding-box algebra as computeAABB (forward.cu:133-163) applied to T diag(r,r,1);
// {rho2d <= tau} is the disc of radius sqrt(tau/2) pixels around the projected centre.  Intersecting the
// union of both boxes with the reference's rectangle drops pairs that cannot touch any pixel -- rendered
// results are unchanged, only the private lists get shorter.  Evaluated in double (the box is a difference
// of O(width^2) terms) with a 0.01 px + 1e-4 relative margin; when the disc is not entirely in front of
// the camera plane (projection not an ellipse) the reference rectangle is kept.
DGS_HD void tight_tile_rect(const float* Tu, const float* Tv, const float* Tw, float cxy_x, float cxy_y, float opacity,
                            int& x0, int& y0, int& x1, int& y1, float bbox[4])
{
    bbox[0] = bbox[2] = -3.0e38f; bbox[1] = bbox[3] = 3.0e38f;
    if (!(opacity >= kAlphaMin)) { x1 = x0; y1 = y0; return; }  // alpha <= o * G <= o < 1/255 everywhere
    const double tau = 2.0 * log(255.0 * (double)opacity) * (1.0 + 1e-6) + 1e-5;
    const double r2 = tau;
    const double twx = Tw[0], twy = Tw[1], twz = Tw[2];
    const double d = r2 * (twx * twx + twy * twy) - twz * twz;
    if (!(d < -1e-9 * twz * twz) || !(twz > 0.0)) return;   // not a regular ellipse: keep the reference rectangle
    const double inv = 1.0 / d;
    const double r2d = sqrt(0.5 * tau);
    double lo[2], hi[2];
    const float* Trow[2] = {Tu, Tv};
    const double cref[2] = {(double)cxy_x, (double)cxy_y};
    for (int a = 0; a < 2; a++) {
        const double tx = Trow[a][0], ty = Trow[a][1], tz = Trow[a][2];
        const double c = (r2 * (tx * twx + ty * twy) - tz * twz) * inv;
        double h0 = c * c - (r2 * (tx * tx + ty * ty) - tz * tz) * inv;
        const double e = sqrt(h0 > 0.0 ? h0 : 0.0);
        double l = c - e, h = c + e;
        if (cref[a] - r2d < l) l = cref[a] - r2d;
        if (cref[a] + r2d > h) h = cref[a] + r2d;
        const double m = 0.01 + 1e-4 * (h - l);
        lo[a] = l - m; hi[a] = h + m;
    }
    // float box for the per-quadrant culling of the blend kernels (rounded outwards; the margin above dwarfs one ulp)
    bbox[0] = (float)lo[0] - 1e-3f; bbox[1] = (float)hi[0] + 1e-3f;
    bbox[2] = (float)lo[1] - 1e-3f; bbox[3] = (float)hi[1] + 1e-3f;
    // tile t holds pixel centres 16 t + 0.5 ... 16 t + 15.5
    const double fx0 = ceil((lo[0] - 15.5) / kTileX), fx1 = floor((hi[0] - 0.5) / kTileX) + 1.0;
    const double fy0 = ceil((lo[1] - 15.5) / kTileY), fy1 = floor((hi[1] - 0.5) / kTileY) + 1.0;
    if (fx0 > (double)x0) x0 = fx0 < (double)x1 ? (int)fx0 : x1;
    if (fy0 > (double)y0) y0 = fy0 < (double)y1 ? (int)fy0 : y1;
    if (fx1 < (double)x1) x1 = fx1 > (double)x0 ? (int)fx1 : x0;
    if (fy1 < (double)y1) y1 = fy1 > (double)y0 ? (int)fy1 : y0;
}

// Pixel <-> lane mapping of the blend kernels: wave w of a tile's workgroup owns the 8x8 quadrant (w & 1, w >> 1); its 16-lane
// DPP row r = lane >> 4 owns the 4x4 block (r & 1, r >> 1) of the quadrant, lane j = lane & 15 of the row the pixel (j & 3, j >> 2)
// of the block.  (Square wave footprints are visited by ~10 % fewer (wave, entry) pairs than 16x4 strips for the same splats,
// tools/blend_stats.py; rows as blocks since round 4, when every row got its own list.)
DGS_HD void lane_pixel(int tid, int& lx, int& ly)
{
    const int w = tid >> 6, l = tid & 63, r = l >> 4, j = l & 15;
    lx = 8 * (w & 1) + 4 * (r & 1) + (j & 3);
    ly = 8 * (w >> 1) + 4 * (r >> 1) + (j >> 2);
}

// Bit w set: the box reaches a pixel centre of quadrant w of the tile whose first pixel is (px0, py0).  Used by the blend
// kernels while staging: a wave only visits entries whose bit for its quadrant is set.
DGS_HD uint32_t quad_mask(float x_lo, float x_hi, float y_lo, float y_hi, float px0, float py0)
{
    const bool x0 = x_hi >= px0 + 0.5f && x_lo <= px0 + 7.5f, x1 = x_hi >= px0 + 8.5f && x_lo <= px0 + 15.5f;
    const bool y0 = y_hi >= py0 + 0.5f && y_lo <= py0 + 7.5f, y1 = y_hi >= py0 + 8.5f && y_lo <= py0 + 15.5f;
    return ((x0 && y0) ? 1u : 0u) | ((x1 && y0) ? 2u : 0u) | ((x0 && y1) ? 4u : 0u) | ((x1 && y1) ? 8u : 0u);
}

// Refinement of quad_mask (NOT in the reference): the footprint {alpha >= 1/255} is the union of the conic
//   Q(x, y) = |P.xy|^2 - tau P.z^2 <= 0,   P(x, y) = Tu x Tv + x (Tv x Tw) + y (Tw x Tu)     (rho3d <= tau, forward.cu:359-384)
// and the low-pass disc |xy - centre|^2 <= tau / 2 (rho2d <= tau), tau = 2 ln(255 o).  A thin or tilted splat fills a small
// part of its bounding box, so a fifth of the quadrants the box test lets through hold no passing pixel.  Here Q is
// minimised over the 8x8 block of pixel centres of each quadrant (convex when the conic is an ellipse: interior minimiser or
// the four clamped edges).  fp32 throughout, in tile-centred coordinates; the same cancellation as in the per-pixel
// evaluation applies (relative error up to ~4e-4 of tau for sub-pixel splats far from the principal point), so tau is
// inflated by 1 % + 0.01 -- half a percent of the footprint's extent.  Anything that is not a proper ellipse keeps the box
// mask.  tests/hostmath proves on the test scenes that no quadrant with a passing pixel is ever dropped.
DGS_HD uint32_t quad_mask_conic(const Quad& q0, const Quad& q1, const Quad& q2, uint32_t box_mask, float px0, float py0)
{
    if (box_mask == 0u) return 0u;
    const float tau = 2.0f * logf(255.0f * q2.w) * 1.01f + 0.01f;
    if (!(tau > 0.0f)) return 0u;                       // opacity below 1/255 (or not a number): no pixel can pass
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y, Twx = q1.z, Twy = q1.w, Twz = q2.x;
    const float Ax = Tuy * Tvz - Tuz * Tvy, Ay = Tuz * Tvx - Tux * Tvz, Az = Tux * Tvy - Tuy * Tvx;
    const float Bx = Tvy * Twz - Tvz * Twy, By = Tvz * Twx - Tvx * Twz, Bz = Tvx * Twy - Tvy * Twx;
    const float Cx = Twy * Tuz - Twz * Tuy, Cy = Twz * Tux - Twx * Tuz, Cz = Twx * Tuy - Twy * Tux;
    const float X0 = px0 + 8.0f, Y0 = py0 + 8.0f;       // pixel centres of the tile: (X0 + u, Y0 + v), u, v in -7.5 .. 7.5
    const float Px = Ax + X0 * Bx + Y0 * Cx, Py = Ay + X0 * By + Y0 * Cy, Pz = Az + X0 * Bz + Y0 * Cz;
    // Q(u, v) = a u^2 + 2 c u v + b v^2 + 2 d u + 2 e v + f
    const float a = Bx * Bx + By * By - tau * Bz * Bz, b = Cx * Cx + Cy * Cy - tau * Cz * Cz;
    const float c = Bx * Cx + By * Cy - tau * Bz * Cz;
    const float d = Px * Bx + Py * By - tau * Pz * Bz, e = Px * Cx + Py * Cy - tau * Pz * Cz;
    const float f = Px * Px + Py * Py - tau * Pz * Pz;
    const float det = a * b - c * c;
    if (!(a > 0.0f && b > 0.0f && det > 1e-6f * a * b)) return box_mask;   // not a bounded, well-conditioned ellipse
    const float inv_det = 1.0f / det, inv_a = 1.0f / a, inv_b = 1.0f / b;
    const float uc = (c * e - b * d) * inv_det, vc = (c * d - a * e) * inv_det;
    const float tol = 2e-6f * (64.0f * (a + b) + 16.0f * (fabsf(d) + fabsf(e)) + fabsf(f));
    // (the ellipse is never empty: rho3d = 0 at the splat's origin.  Its minimum VALUE is not used -- f + d uc + e vc
    // cancels catastrophically for thin tilted footprints.)
    const float r2 = 0.5f * tau, cx = q2.y - X0, cy = q2.z - Y0;
    uint32_t m = 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float u0 = (w & 1) ? 0.5f : -7.5f, u1 = u0 + 7.0f, v0 = (w & 2) ? 0.5f : -7.5f, v1 = v0 + 7.0f;
        // low-pass disc: distance from the centre to the block of pixel centres
        const float gx = fmaxf(fmaxf(u0 - cx, cx - u1), 0.0f), gy = fmaxf(fmaxf(v0 - cy, cy - v1), 0.0f);
        bool hit = gx * gx + gy * gy <= r2 * 1.0001f;
        hit |= (uc >= u0) & (uc <= u1) & (vc >= v0) & (vc <= v1);
        auto Qf = [&](float u, float v) { return u * (a * u + 2.0f * (c * v + d)) + v * (b * v + 2.0f * e) + f; };
        auto clampf = [](float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); };
        float qm = Qf(u0, clampf(-(c * u0 + e) * inv_b, v0, v1));
        qm = fminf(qm, Qf(u1, clampf(-(c * u1 + e) * inv_b, v0, v1)));
        qm = fminf(qm, Qf(clampf(-(c * v0 + d) * inv_a, u0, u1), v0));
        qm = fminf(qm, Qf(clampf(-(c * v1 + d) * inv_a, u0, u1), v1));
        hit |= qm <= tol;
        m |= hit ? (1u << w) : 0u;
    }
    return m & box_mask;
}

// Packed tile rectangle: x0 | x1 << 16, y0 | y1 << 16
struct TileRect { uint32_t xs, ys; };

// forward.cu:166-260 preprocessCUDA with computeTransMat (:75-128), computeAABB (:133-163) and the
// near cull of in_frustum (auxiliary.h:160-185).  Returns the integer radius (0 = culled) and the
// tile count.  Contraction is switched off so that the culling / ceil / (int) decisions are taken on
// the same IEEE values as the CPU oracle.
DGS_HD int preprocess_surfel(const Camera& cam, const float* pos, const float* scale, const float* quat, float opacity,
                             int deg, const float* sh /*or null*/, const float* color_precomp /*or null*/,
                             SurfelRec& rec, int& tiles, TileRect& trect, bool tight = true)
{
    tiles = 0;
    trect.xs = trect.ys = 0;
    const float* vm = cam.view;
    float pv[3];
    pv[0] = vm[0] * pos[0] + vm[4] * pos[1] + vm[8] * pos[2] + vm[12];
    pv[1] = vm[1] * pos[0] + vm[5] * pos[1] + vm[9] * pos[2] + vm[13];
    pv[2] = vm[2] * pos[0] + vm[6] * pos[1] + vm[10] * pos[2] + vm[14];
    if (pv[2] <= 0.2f) return 0;

    float c0[3], c1[3], c2[3];
    quat_columns(quat, c0, c1, c2);
    float rs0[3] = {c0[0] * scale[0], c0[1] * scale[0], c0[2] * scale[0]};
    float rs1[3] = {c1[0] * scale[1], c1[1] * scale[1], c1[2] * scale[1]};
    float m0[3], m1[3], tn[3];
    view_rotate(vm, rs0, m0);
    view_rotate(vm, rs1, m1);
    view_rotate(vm, c2, tn);
    float cosv = -tn[0] * pv[0] + -tn[1] * pv[1] + -tn[2] * pv[2];
    if (cosv == 0.0f) return 0;
    float flip = cosv > 0 ? 1.f : -1.f;

    const float cx = (float)cam.width / 2.0f, cy = (float)cam.height / 2.0f;
    float Tu[3] = {cam.focal_x * m0[0] + cx * m0[2], cam.focal_x * m1[0] + cx * m1[2], cam.focal_x * pv[0] + cx * pv[2]};
    float Tv[3] = {cam.focal_y * m0[1] + cy * m0[2], cam.focal_y * m1[1] + cy * m1[2], cam.focal_y * pv[1] + cy * pv[2]};
    float Tw[3] = {m0[2], m1[2], pv[2]};

    float d = Tw[0] * Tw[0] + Tw[1] * Tw[1] - Tw[2] * Tw[2];
    if (d == 0.0f) return 0;
    float inv = 1.0f / d;
    float f[3] = {inv, inv, -inv};
    float px = f[0] * (Tu[0] * Tw[0]) + f[1] * (Tu[1] * Tw[1]) + f[2] * (Tu[2] * Tw[2]);
    float py = f[0] * (Tv[0] * Tw[0]) + f[1] * (Tv[1] * Tw[1]) + f[2] * (Tv[2] * Tw[2]);
    float h0x = px * px - (f[0] * (Tu[0] * Tu[0]) + f[1] * (Tu[1] * Tu[1]) + f[2] * (Tu[2] * Tu[2]));
    float h0y = py * py - (f[0] * (Tv[0] * Tv[0]) + f[1] * (Tv[1] * Tv[1]) + f[2] * (Tv[2] * Tv[2]));
    float ex = sqrtf(h0x > 0.f ? h0x : 0.f), ey = sqrtf(h0y > 0.f ? h0y : 0.f);
    float emax = ex > ey ? ex : ey;
    // forward.cu:231: ceil(3.f * max(max(ex,ey), FilterSize)) with FilterSize a double literal
    double em = (double)emax > 0.7071067811865476 ? (double)emax : 0.7071067811865476;
    int radius = (int)(float)ceil(3.0 * em);
    int x0, y0, x1, y1;
    tile_rect(px, py, radius, cam.tiles_x, cam.tiles_y, x0, y0, x1, y1);
    int cnt = (x1 - x0) * (y1 - y0);
    if (cnt == 0) return 0;

    for (int c = 0; c < 3; c++) { rec.Tu[c] = Tu[c]; rec.Tv[c] = Tv[c]; rec.Tw[c] = Tw[c]; rec.normal[c] = tn[c] * flip; }
    rec.xy[0] = px; rec.xy[1] = py;
    rec.opacity = opacity;
    rec.depth = pv[2];
    if (color_precomp) {
        rec.rgb[0] = color_precomp[0]; rec.rgb[1] = color_precomp[1]; rec.rgb[2] = color_precomp[2];
        rec.flags = 0;
    } else {
        rec.flags = sh_to_rgb(deg, sh, pos, cam.campos, rec.rgb);
    }
    rec.bbox[0] = rec.bbox[2] = -3.0e38f; rec.bbox[1] = rec.bbox[3] = 3.0e38f;
    if (tight) tight_tile_rect(Tu, Tv, Tw, px, py, opacity, x0, y0, x1, y1, rec.bbox);
    trect.xs = (uint32_t)x0 | ((uint32_t)x1 << 16);
    trect.ys = (uint32_t)y0 | ((uint32_t)y1 << 16);
    tiles = (x1 - x0) * (y1 - y0);   // may be 0: visible (radius > 0) but unable to reach alpha >= 1/255 anywhere
    return radius;
}

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

// ---------------------------------------------------------------------------------------------
// Per-(pixel, surfel) evaluation shared by the forward and backward blend (forward.cu:359-399,
// backward.cu:283-323).
struct PairEval {
    float kx, ky, kz, lx, ly, lz;  // the two homogeneous planes
    float pz, inv_pz;              // z of their cross product and its reciprocal
    float sx, sy;                  // intersection in splat space
    float dx, dy;                  // projected centre minus pixel centre
    float G, alpha, depth;
    bool use3d;                    // rho3d <= rho2d
};

DGS_HD bool pair_eval(float pfx, float pfy, const Quad& q0, const Quad& q1, const Quad& q2, PairEval& e)
{
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y, Twx = q1.z, Twy = q1.w, Twz = q2.x;
    e.kx = pfx * Twx - Tux; e.ky = pfx * Twy - Tuy; e.kz = pfx * Twz - Tuz;
    e.lx = pfy * Twx - Tvx; e.ly = pfy * Twy - Tvy; e.lz = pfy * Twz - Tvz;
    float ppx = e.ky * e.lz - e.kz * e.ly;
    float ppy = e.kz * e.lx - e.kx * e.lz;
    e.pz = e.kx * e.ly - e.ky * e.lx;
    if (e.pz == 0.0f) return false;
    float inv = fast_rcp(e.pz);
    e.inv_pz = inv;
    e.sx = ppx * inv; e.sy = ppy * inv;
    float rho3d = e.sx * e.sx + e.sy * e.sy;
    e.dx = q2.y - pfx; e.dy = q2.z - pfy;
    float rho2d = 2.0f * (e.dx * e.dx + e.dy * e.dy);   // FilterInvSquare == 2, auxiliary.h:20-21
    e.use3d = rho3d <= rho2d;
    float rho = e.use3d ? rho3d : rho2d;
    e.depth = e.use3d ? (e.sx * Twx + e.sy * Twy) + Twz : Twz;
    if (e.depth < kNear) return false;                  // float 0.2f: same set as (double)depth < 0.2
    float power = -0.5f * rho;
    if (power > 0.0f) return false;
    e.G = fast_exp(power);
    float a = q2.w * e.G;
    e.alpha = a < kAlphaMax ? a : kAlphaMax;
    return !(e.alpha < kAlphaMin);
}

// Branch-free variant used by the kernels: everything is evaluated for every lane (no exec-mask nest,
// no serialised LDS round trips) and the four skips of forward.cu:369-399 become one predicate.
// Inf/NaN produced by pz == 0 never pass because pz != 0 is part of the predicate.
DGS_HD bool pair_eval_bf(float pfx, float pfy, const Quad& q0, const Quad& q1, const Quad& q2, PairEval& e)
{
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y, Twx = q1.z, Twy = q1.w, Twz = q2.x;
    e.kx = pfx * Twx - Tux; e.ky = pfx * Twy - Tuy; e.kz = pfx * Twz - Tuz;
    e.lx = pfy * Twx - Tvx; e.ly = pfy * Twy - Tvy; e.lz = pfy * Twz - Tvz;
    const float ppx = e.ky * e.lz - e.kz * e.ly;
    const float ppy = e.kz * e.lx - e.kx * e.lz;
    e.pz = e.kx * e.ly - e.ky * e.lx;
    const float inv = fast_rcp(e.pz);
    e.inv_pz = inv;
    e.sx = ppx * inv; e.sy = ppy * inv;
    const float rho3d = e.sx * e.sx + e.sy * e.sy;
    e.dx = q2.y - pfx; e.dy = q2.z - pfy;
    const float rho2d = 2.0f * (e.dx * e.dx + e.dy * e.dy);
    e.use3d = rho3d <= rho2d;
    const float rho = e.use3d ? rho3d : rho2d;
    const float d3 = (e.sx * Twx + e.sy * Twy) + Twz;
    e.depth = e.use3d ? d3 : Twz;
    const float power = -0.5f * rho;
    e.G = fast_exp(power);
    const float a = q2.w * e.G;
    e.alpha = a < kAlphaMax ? a : kAlphaMax;
    return (e.pz != 0.0f) & (e.depth >= kNear) & !(power > 0.0f) & (e.alpha >= kAlphaMin);
}

// ---------------------------------------------------------------------------------------------
// Affine form of the same evaluation around the splat's projected centre -- what the blend kernels run since round 3.
//
// forward.cu:359-368 builds, per (pixel, entry), the planes k = x Tw - Tu, l = y Tw - Tv and their cross product
// p = k x l: 6 FMAs + 6 products.  p is AFFINE in the pixel: with c = (cx, cy) the entry's projected centre (its `xy`),
//     kc = cx Tw - Tu,  lc = cy Tw - Tv            (the reference's own planes, taken at the centre)
//     p(x, y) = (kc + (x - cx) Tw) x (lc + (y - cy) Tw) = kc x lc + (x - cx) (Tw x lc) + (y - cy) (kc x Tw)
// so nine constants are computed ONCE per (tile, entry) by the thread that stages the entry and a pixel pays 6 FMAs.  The
// expansion point matters: p.xy vanishes at the centre of the splat, so around the centre no term is larger than the result
// (around the tile centre, 8 px away, a sub-pixel splat's terms cancel to 1/100 of their size and the error of a gradient-free
// constant lands on every pixel; the global form Tu x Tv + x (Tv x Tw) + y (Tw x Tu) is worse still).  The pixel offset is
// already there: the low-pass term (auxiliary.h:20-21, FilterInvSquare = 2) needs d = c - pixel, carried pre-scaled
// (ds = sqrt2 d, from tile-relative operands: cs = sqrt2 (c - tile centre), us = sqrt2 (pixel - tile centre)) so that
// rho2d = dxs^2 + dys^2 needs no doubling, and p = A + dxs B + dys C with B = -(Tw x lc) / sqrt2, C = -(kc x Tw) / sqrt2.
// Staged image of an entry (floats):
//   a0 = (A.x A.y A.z B.x)  a1 = (B.y B.z C.x C.y)  a2 = (C.z cxs cys opacity)       <- alpha part, read for every visit
//   Tw, normal, colour: the record's own q1/q2 (Tw), q3, q4                         <- read when some pixel blends the entry
constexpr float kSqrt2 = 1.41421356237309505f;
constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kNegHalfLog2e = -0.72134752044448170f;   // G = exp(-rho/2) = 2^(rho * this)
constexpr float kDepthC1 = 100.0f / 99.8f;               // mapped depth (FAR d - FAR NEAR) / ((FAR - NEAR) d) = C1 - C2 / d, forward.cu:412
constexpr float kDepthC2 = 20.0f / 99.8f;

DGS_HD float fast_exp2(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DGS_PRECISE_MATH)
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}

struct TileAffine {
    Quad a0, a1, a2;
    float kcx, kcy, lcx, lcy;   // x, y of the centre planes: the backward rebuilds k.xy, l.xy of a pixel from them
};

// X0, Y0: pixel-centre coordinates of the tile's centre (tile origin + 8; pixel centres are X0 + u, u = -7.5 .. 7.5)
DGS_HD TileAffine tile_affine(const Quad& q0, const Quad& q1, const Quad& q2, float X0, float Y0)
{
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y, Twx = q1.z, Twy = q1.w, Twz = q2.x;
    const float cx = q2.y, cy = q2.z;
    const float kx = cx * Twx - Tux, ky = cx * Twy - Tuy, kz = cx * Twz - Tuz;
    const float lx = cy * Twx - Tvx, ly = cy * Twy - Tvy, lz = cy * Twz - Tvz;
    TileAffine t;
    t.kcx = kx; t.kcy = ky; t.lcx = lx; t.lcy = ly;
    t.a0.x = ky * lz - kz * ly;   t.a0.y = kz * lx - kx * lz;   t.a0.z = kx * ly - ky * lx;                       // A = kc x lc
    t.a0.w = -kInvSqrt2 * (Twy * lz - Twz * ly); t.a1.x = -kInvSqrt2 * (Twz * lx - Twx * lz); t.a1.y = -kInvSqrt2 * (Twx * ly - Twy * lx);   // B
    t.a1.z = -kInvSqrt2 * (ky * Twz - kz * Twy); t.a1.w = -kInvSqrt2 * (kz * Twx - kx * Twz); t.a2.x = -kInvSqrt2 * (kx * Twy - ky * Twx);   // C
    t.a2.y = kSqrt2 * (cx - X0); t.a2.z = kSqrt2 * (cy - Y0); t.a2.w = q2.w;
    return t;
}

// Can the entry reach alpha >= 1/255 on the block of pixels whose scaled tile-relative coordinates span [us0, us1] x [vs0, vs1]
// (an 8x8 quadrant: us1 = us0 + 7 sqrt2)?  Same test as quad_mask_conic -- the footprint is the union of the conic
// |P.xy|^2 - tau P.z^2 <= 0 and the low-pass disc rho2d <= tau, tau = 2 ln(255 o) -- written in the variables of the affine form
// (x, y) = (dxs, dys) = cs - (us, vs), where P = A + x B + y C and rho2d = x^2 + y^2, so it costs no cross products of its own and
// inherits the conditioning of the centre expansion.  tau is inflated by 1 % + 0.01; anything that is not a proper ellipse
// counts as a hit.  tests/hostmath proves on the test scenes that no block with a passing pixel is ever dropped.
DGS_HD bool block_hit_affine(const TileAffine& t, float us0, float us1, float vs0, float vs1)
{
    const float tau = 2.0f * logf(255.0f * t.a2.w) * 1.01f + 0.01f;
    const bool can_pass = tau > 0.0f;                    // false: opacity below 1/255 (or not a number), no pixel can pass
    const float Ax = t.a0.x, Ay = t.a0.y, Az = t.a0.z, Bx = t.a0.w, By = t.a1.x, Bz = t.a1.y, Cx = t.a1.z, Cy = t.a1.w, Cz = t.a2.x;
    const float x0 = t.a2.y - us1, x1 = t.a2.y - us0, y0 = t.a2.z - vs1, y1 = t.a2.z - vs0;   // the block in (dxs, dys)
    // low-pass disc: distance from the centre (x = y = 0) to the block
    const float gx = fmaxf(fmaxf(x0, -x1), 0.0f), gy = fmaxf(fmaxf(y0, -y1), 0.0f);
    bool hit = gx * gx + gy * gy <= tau * 1.0001f;
    // Q(x, y) = a x^2 + 2 c x y + b y^2 + 2 d x + 2 e y + f
    const float a = Bx * Bx + By * By - tau * Bz * Bz, b = Cx * Cx + Cy * Cy - tau * Cz * Cz;
    const float c = Bx * Cx + By * Cy - tau * Bz * Cz;
    const float d = Ax * Bx + Ay * By - tau * Az * Bz, e = Ax * Cx + Ay * Cy - tau * Az * Cz;
    const float f = Ax * Ax + Ay * Ay - tau * Az * Az;
    const float det = a * b - c * c;
    const bool no_ellipse = !((a > 0.0f) & (b > 0.0f) & (det > 1e-6f * a * b));   // not a bounded, well-conditioned ellipse: a hit
    const float inv_det = 1.0f / det, inv_a = 1.0f / a, inv_b = 1.0f / b;
    const float xc = (c * e - b * d) * inv_det, yc = (c * d - a * e) * inv_det;
    const float X = fmaxf(fabsf(x0), fabsf(x1)), Y = fmaxf(fabsf(y0), fabsf(y1));
    const float tol = 2e-6f * (a * X * X + b * Y * Y + 2.0f * (fabsf(c) * X * Y + fabsf(d) * X + fabsf(e) * Y) + fabsf(f));
    hit |= (xc >= x0) & (xc <= x1) & (yc >= y0) & (yc <= y1);
    auto Qf = [&](float x, float y) { return x * (a * x + 2.0f * (c * y + d)) + y * (b * y + 2.0f * e) + f; };
    auto clampf = [](float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); };
    float qm = Qf(x0, clampf(-(c * x0 + e) * inv_b, y0, y1));
    qm = fminf(qm, Qf(x1, clampf(-(c * x1 + e) * inv_b, y0, y1)));
    qm = fminf(qm, Qf(clampf(-(c * y0 + d) * inv_a, x0, x1), y0));
    qm = fminf(qm, Qf(clampf(-(c * y1 + d) * inv_a, x0, x1), y1));
    hit |= qm <= tol;
    return can_pass & (hit | no_ellipse);
}

// Which of the four 4x4 pixel blocks of an 8x8 quadrant can the entry reach with alpha >= 1/255?  Bit b = block (b & 1, b >> 1);
// (us0, vs0) = scaled tile-relative coordinates of the quadrant's first pixel.  Used by the row-per-block blend kernels (round 4):
// every 16-lane row of a wave walks the list of ITS block.  Same footprint as block_hit_affine -- the conic Q = |P.xy|^2 - tau P.z^2
// <= 0, P = A + x B + y C, or the low-pass disc x^2 + y^2 <= tau -- but the four blocks are small (half extent h = 1.5 sqrt2 in
// these coordinates) and a splat that reaches the quadrant is mostly much larger, so instead of minimising Q over each block
// exactly (interior point + four clamped edges: ~75 operations per block) Q is bounded from below by its tangent plane at the
// block's centre: Q(c + d) >= Q(c) - h (|Qx(c)| + |Qy(c)|) for |d.x|, |d.y| <= h, valid because Q is convex when the conic is a
// proper ellipse (anything else counts as a hit).  Conservative by construction: a block is only dropped when the bound proves
// that no pixel centre of it lies inside the (1 % + 0.01 inflated) footprint; what the bound gives away against the exact
// minimum is ~4 % more (entry, block) pairs on the 200k / 800x800 scene (tools/blend_stats.py).
// bx = the record's exact pixel box (x_lo, x_hi, y_lo, y_hi); (qx, qy) = absolute coordinates of the quadrant's first pixel.
DGS_HD uint32_t blocks_hit_linear(const TileAffine& t, float us0, float vs0, const Quad& bx, float qx, float qy)
{
    const float tau = 2.0f * logf(255.0f * t.a2.w) * 1.01f + 0.01f;
    const bool can_pass = tau > 0.0f;                    // false: opacity below 1/255 (or not a number)
    const float Ax = t.a0.x, Ay = t.a0.y, Az = t.a0.z, Bx = t.a0.w, By = t.a1.x, Bz = t.a1.y, Cx = t.a1.z, Cy = t.a1.w, Cz = t.a2.x;
    const float a = Bx * Bx + By * By - tau * Bz * Bz, b = Cx * Cx + Cy * Cy - tau * Cz * Cz;
    const float c = Bx * Cx + By * Cy - tau * Bz * Cz;
    const float det = a * b - c * c;
    const bool no_ellipse = !((a > 0.0f) & (b > 0.0f) & (det > 1e-6f * a * b));   // not a bounded, well-conditioned ellipse: a hit
    const float h = 1.5f * kSqrt2;
    // block centres in (x, y) = cs - (us, vs): column i, row j
    const float xm[2] = {t.a2.y - (us0 + 1.5f * kSqrt2), t.a2.y - (us0 + 5.5f * kSqrt2)};
    const float ym[2] = {t.a2.z - (vs0 + 1.5f * kSqrt2), t.a2.z - (vs0 + 5.5f * kSqrt2)};
    // the record's pixel box against the two columns / rows of block centres (pixel centres qx + 0.5 .. + 3.5 and + 4.5 .. + 7.5)
    const bool bxc[2] = {bx.y >= qx + 0.5f && bx.x <= qx + 3.5f, bx.y >= qx + 4.5f && bx.x <= qx + 7.5f};
    const bool byr[2] = {bx.w >= qy + 0.5f && bx.z <= qy + 3.5f, bx.w >= qy + 4.5f && bx.z <= qy + 7.5f};
    // low-pass disc: squared distance from the centre (x = y = 0) to the block, per column / row
    float gx2[2], gy2[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float gx = fmaxf(fabsf(xm[i]) - h, 0.0f), gy = fmaxf(fabsf(ym[i]) - h, 0.0f);
        gx2[i] = gx * gx; gy2[i] = gy * gy;
    }
    const float tau_d = tau * 1.0001f;
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float Pxi = Ax + xm[i] * Bx, Pyi = Ay + xm[i] * By, Pzi = Az + xm[i] * Bz;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float Px = Pxi + ym[j] * Cx, Py = Pyi + ym[j] * Cy, Pz = Pzi + ym[j] * Cz;
            const float tPz = tau * Pz;
            const float Q = Px * Px + (Py * Py - tPz * Pz);
            const float hQx = Px * Bx + (Py * By - tPz * Bz), hQy = Px * Cx + (Py * Cy - tPz * Cz);   // half the partial derivatives
            const float lb = Q - (2.0f * h) * (fabsf(hQx) + fabsf(hQy));
            const float tol = 4e-6f * (Px * Px + Py * Py + tPz * Pz);
            const bool hit = (lb <= tol) | no_ellipse | (gx2[i] + gy2[j] <= tau_d);
            m |= (hit & bxc[i] & byr[j]) ? (1u << (2 * j + i)) : 0u;
        }
    }
    return can_pass ? m : 0u;
}

// quadrant w of the tile: us in [us_lo(w), us_lo(w) + 7 sqrt2]
DGS_HD bool quad_hit_affine(const TileAffine& t, int w)
{
    const float us0 = kSqrt2 * ((w & 1) ? 0.5f : -7.5f), vs0 = kSqrt2 * ((w & 2) ? 0.5f : -7.5f);
    return block_hit_affine(t, us0, us0 + 7.0f * kSqrt2, vs0, vs0 + 7.0f * kSqrt2);
}

struct AlphaEval {
    float pz, inv_pz, sx, sy;      // p.z, its reciprocal, the intersection in splat space
    float dxs, dys;                // sqrt2 * (projected centre - pixel)
    float rho3d, rho2d;
    float G, a, alpha;             // Gaussian weight, opacity * G before and after the 0.99 clamp (forward.cu:397)
};

// (us, vs) = sqrt2 (pixel - tile centre).  Returns whether the pixel passes the alpha test of forward.cu:369-399 EXCEPT the
// near-plane test on the depth (alpha_depth below), which needs Tw and is evaluated only when some pixel of the wave passes
// here.  Two comparisons decide (their masks meet in one scalar AND):
//   * p.z == 0 exactly (forward.cu:368) is the reference's own skip.  (Rounds 1-5 poisoned rho with `0 * rho3d` instead, which
//     also skipped a pair whose rho3d merely OVERFLOWS -- |s| > 1.8e19, a splat axis below ~1e-19 -- where the reference falls back
//     to the low-pass disc, forward.cu:381-382.  Now an infinite rho3d takes min(rho3d, rho2d) = rho2d like the reference; the
//     compare replaces the multiply-add, the VALU count of the visit is unchanged.)
//   * a lane that is done or outside the image carries us = NaN: every quantity below is NaN for it and fails `a >= 1/255`.
DGS_HD bool alpha_affine(float us, float vs, const Quad& a0, const Quad& a1, const Quad& a2, AlphaEval& e)
{
    e.dxs = a2.y - us; e.dys = a2.z - vs;
    const float px = e.dys * a1.z + (e.dxs * a0.w + a0.x);
    const float py = e.dys * a1.w + (e.dxs * a1.x + a0.y);
    e.pz = e.dys * a2.x + (e.dxs * a1.y + a0.z);
    e.inv_pz = fast_rcp(e.pz);
    e.sx = px * e.inv_pz; e.sy = py * e.inv_pz;
    e.rho3d = e.sy * e.sy + e.sx * e.sx;
    e.rho2d = e.dys * e.dys + e.dxs * e.dxs;
    const float rho = fminf(e.rho3d, e.rho2d);
    e.G = fast_exp2(rho * kNegHalfLog2e);
    e.a = a2.w * e.G;
    e.alpha = fminf(e.a, kAlphaMax);
    return (e.a >= kAlphaMin) & (e.pz != 0.0f);
}

// depth of the pair (forward.cu:385-387: the intersection's when the 3-D distance is the smaller one, the centre's otherwise)
DGS_HD float alpha_depth(const AlphaEval& e, float Twx, float Twy, float Twz, bool& use3d)
{
    use3d = e.rho3d <= e.rho2d;
    const float d3 = e.sy * Twy + (e.sx * Twx + Twz);
    return use3d ? d3 : Twz;
}

// (FAR*d - FAR*NEAR) / ((FAR-NEAR)*d), forward.cu:412, through one reciprocal (v_rcp_f32, 1 ulp) that the
// backward reuses for d(mapped)/d(depth) = FAR*NEAR / ((FAR-NEAR) d^2), backward.cu:352
DGS_HD float mapped_depth_r(float depth, float rd /* = fast_rcp(depth) */)
{
    return (100.0f * depth - 20.0f) * (rd * (1.0f / 99.8f));
}

DGS_HD float mapped_depth(float depth) { return mapped_depth_r(depth, fast_rcp(depth)); }

// Running per-pixel state of the forward blend (forward.cu:313-329).
struct PixFwd {
    float T;
    float C[3];
    float D;
    float N[3];
    float dist1, dist2, distortion;
    float med_d, med_w;
    uint32_t contributor, last, med_c;
};

DGS_HD void pixfwd_init(PixFwd& s)
{
    s.T = 1.f; s.C[0] = s.C[1] = s.C[2] = 0.f; s.D = 0.f; s.N[0] = s.N[1] = s.N[2] = 0.f;
    s.dist1 = s.dist2 = s.distortion = 0.f; s.med_d = s.med_w = 0.f; s.contributor = 0; s.last = 0; s.med_c = 0;
}

// forward.cu:400-438 for one contributing entry. Returns false when the pixel saturates (T < 1e-4).
DGS_HD bool pixfwd_blend(PixFwd& s, const PairEval& e, const Quad& q3, const Quad& q4)
{
    float test_T = s.T * (1.f - e.alpha);
    if (test_T < kTmin) return false;
    float w = e.alpha * s.T;
    float A = 1.f - s.T;
    float m = mapped_depth(e.depth);
    float err = m * m * A + s.dist2 - 2.f * m * s.dist1;
    s.distortion += err * w;
    if (s.T > 0.5f) { s.med_d = e.depth; s.med_w = w; s.med_c = s.contributor; }
    s.N[0] += q3.x * w; s.N[1] += q3.y * w; s.N[2] += q3.z * w;
    s.D += e.depth * w;
    s.dist1 += m * w;
    s.dist2 += m * m * w;
    s.C[0] += q3.w * w; s.C[1] += q4.x * w; s.C[2] += q4.y * w;
    s.T = test_T;
    s.last = s.contributor;
    return true;
}

// forward.cu:400-438 for one contributing entry, in two parts so that the kernel can evaluate the saturation test for the whole
// wave before it narrows the execution mask once.  w = alpha T and test_T = T - w replace T (1 - alpha): one rounding less on the
// same value.  pixfwd_weight: the pixel saturates (forward.cu:402-406: the entry is NOT blended) when test_T < 1e-4.
DGS_HD void pixfwd_weight(const PixFwd& s, float alpha, float& w, float& test_T)
{
    w = alpha * s.T;
    test_T = s.T - w;
}

// TRACK_MEDIAN = false: no pixel of the wave has T > 0.5 any more (forward.cu:421-425 would not fire)
template <bool TRACK_MEDIAN>
DGS_HD void pixfwd_accumulate(PixFwd& s, float w, float test_T, float depth, const Quad& q3, const Quad& q4)
{
    const float A = 1.f - s.T;
    const float m = kDepthC1 - kDepthC2 * fast_rcp(depth);
    const float mm = m * m;
    const float err = (mm * A + s.dist2) - (m + m) * s.dist1;
    s.distortion += err * w;
    if (TRACK_MEDIAN) {
        if (s.T > 0.5f) { s.med_d = depth; s.med_w = w; s.med_c = s.contributor; }
    }
    s.N[0] += q3.x * w; s.N[1] += q3.y * w; s.N[2] += q3.z * w;
    s.D += depth * w;
    s.dist1 += m * w;
    s.dist2 += mm * w;
    s.C[0] += q3.w * w; s.C[1] += q4.x * w; s.C[2] += q4.y * w;
    s.T = test_T;
    s.last = s.contributor;
}

// Running per-pixel state of the backward blend (backward.cu:191-247), reduced to what the recurrences need.
//
// The reference carries eight "colour behind this entry" recurrences (accum_rec[3], accum_depth_rec, accum_alpha_rec,
// accum_normal_rec[3]) plus last_dL_dT, each with its own `last_*` value, and adds (value_i - accum_i) * g per channel
// into dL_dalpha.  All nine follow the same linear blend  X <- a_prev * value_prev + (1 - a_prev) * X  (last_dL_dT is the
// same recurrence folded one entry early), so their g-weighted sum follows it too: with
//     u_i = colour_i . g_pix + depth_i g_depth + g_alpha + normal_i . g_normal + dL_dweight_i
// one scalar `acc` replaces the fifteen state values:  dL_dalpha_i = (u_i - acc) T_i + background term,
// acc <- acc + alpha_i (u_i - acc).  Same mathematics, different rounding order (not bit-identical to the oracle).
struct PixBwd {
    float T, T_final;
    float acc;           // sum over channels of g * "what lies behind the current entry" (see above)
    float final_D, final_D2, final_A;
    float g_pix[3];      // dL/dcolour
    float g_depth, g_alpha, g_normal[3], g_meddepth, g_dist, g_medw;
    float bg_dot;        // dot(bg, g_pix)
    int last_contributor, med_c;
};

DGS_HD void pixbwd_init(PixBwd& s, float T_final, float dist1, float dist2, int last, int med_c, const float* gpix,
                        const float* gothers /*8*/, const float* bg)
{
    s.T = s.T_final = T_final;
    s.acc = 0.f;
    for (int c = 0; c < 3; c++) s.g_pix[c] = gpix[c];
    s.final_D = dist1; s.final_D2 = dist2; s.final_A = 1.f - T_final;
    s.g_depth = gothers[0]; s.g_alpha = gothers[1];
    s.g_normal[0] = gothers[2]; s.g_normal[1] = gothers[3]; s.g_normal[2] = gothers[4];
    s.g_meddepth = gothers[5]; s.g_dist = gothers[6]; s.g_medw = gothers[7];
    s.bg_dot = bg[0] * gpix[0] + bg[1] * gpix[1] + bg[2] * gpix[2];
    s.last_contributor = last; s.med_c = med_c;
}

// backward.cu:325-446 for one list entry (`contributor` = 0-based list index), branch-free: `ok` says whether this pixel
// blends the entry (alpha test passed and the entry lies in front of the pixel's last contributor).  A pixel that does
// not blend it runs the same instructions on neutral values (alpha = G = 0, depth = 1, s = 0) and leaves its state and
// all sixteen outputs exactly unchanged / zero -- so a wave needs no divergent region, no zero-filled outputs for idle
// lanes and no register copies at the join.  The rho2d <= rho3d case (backward.cu:436-443) is folded in the same way:
// s = 0 and 1/p.z = 0 reduce the 3-D formulas to dL_dT[8] += dL_dz, and the screen-space gradient goes to out2d.
//   out[0..2] dL_dcolour, out[3..5] dL_dnormal, out[6..14] dL_dtransMat, out[15] dL_dopacity (AccSlot order), out2d = dL_dmean2D.
// dL_dk = cross(l, dL_dp) and dL_dl = cross(dL_dp, k) of the reference are parallel to (s.x, s.y, 1) because dL_dp, k and
// l are all orthogonal to p = k x l: dL_dk = (l.x dp.y - l.y dp.x) (s, 1), dL_dl = (dp.x k.y - dp.y k.x) (s, 1).
DGS_HD void pixbwd_step(PixBwd& s, const PairEval& e, bool ok, int contributor, float pfx, float pfy, const Quad& q1, const Quad& q2,
                        const Quad& q3, const Quad& q4, float* out /*[16]*/, float* out2d /*[2]*/)
{
    const bool m3 = ok & e.use3d;
    const float alpha = ok ? e.alpha : 0.f;
    const float G = ok ? e.G : 0.f;
    const float c_d = ok ? e.depth : 1.f;
    const float sx = m3 ? e.sx : 0.f, sy = m3 ? e.sy : 0.f, inv_pz = m3 ? e.inv_pz : 0.f;
    const float inv_1ma = ok ? fast_rcp(1.f - alpha) : 1.f;  // alpha <= 0.99: well conditioned
    s.T = s.T * inv_1ma;
    const float w = alpha * s.T;
    const float r_d = fast_rcp(c_d);
    const float m_d = mapped_depth_r(c_d, r_d);
    const float dmd_dd = (20.0f / 99.8f) * (r_d * r_d);
    const bool is_med = ok & (contributor == s.med_c - 1);
    float dL_dweight = (s.final_D2 + m_d * m_d * s.final_A - 2.f * m_d * s.final_D) * s.g_dist;
    dL_dweight += is_med ? s.g_medw : 0.f;
    float u = q3.w * s.g_pix[0] + q4.x * s.g_pix[1] + q4.y * s.g_pix[2];
    u += c_d * s.g_depth + s.g_alpha;
    u += q3.x * s.g_normal[0] + q3.y * s.g_normal[1] + q3.z * s.g_normal[2];
    u += dL_dweight;
    const float d = u - s.acc;
    const float dL_dalpha = d * s.T - (s.T_final * inv_1ma) * s.bg_dot;
    s.acc += alpha * d;
    float dL_dz = (2.0f * w * (m_d * s.final_A - s.final_D) * s.g_dist) * dmd_dd + w * s.g_depth;
    dL_dz += is_med ? s.g_meddepth : 0.f;
    out[kAccColor + 0] = w * s.g_pix[0]; out[kAccColor + 1] = w * s.g_pix[1]; out[kAccColor + 2] = w * s.g_pix[2];
    out[kAccNormal + 0] = w * s.g_normal[0]; out[kAccNormal + 1] = w * s.g_normal[1]; out[kAccNormal + 2] = w * s.g_normal[2];
    out[kAccOpacity] = G * dL_dalpha;
    const float nGdG = -(G * (q2.w * dL_dalpha));   // -G dL_dG
    const float dsx = nGdG * sx + dL_dz * q1.z, dsy = nGdG * sy + dL_dz * q1.w;
    const float ax = dsx * inv_pz, ay = dsy * inv_pz;   // dL_dp.xy; dL_dp.z = -(ax s.x + ay s.y)
    const float n1 = e.ly * ax - e.lx * ay;             // -dL_dk.z
    const float n2 = ay * e.kx - ax * e.ky;             // -dL_dl.z
    const float m3w = dL_dz - (pfx * n1 + pfy * n2);
    out[kAccT + 0] = n1 * sx; out[kAccT + 1] = n1 * sy; out[kAccT + 2] = n1;
    out[kAccT + 3] = n2 * sx; out[kAccT + 4] = n2 * sy; out[kAccT + 5] = n2;
    out[kAccT + 6] = m3w * sx; out[kAccT + 7] = m3w * sy; out[kAccT + 8] = m3w;
    const float f2 = (ok & !e.use3d) ? 2.0f * nGdG : 0.f;   // dL_dG * (-G * FilterInvSquare), backward.cu:436-443
    out2d[0] = f2 * e.dx; out2d[1] = f2 * e.dy;
}

// Backward of one list entry in the affine form (blend_bwd_kernel since round 3).  Same contract as pixbwd_step: branch-free, a
// pixel that does not blend the entry (`ok` false) runs on neutral values and contributes exact zeros.  Differences:
//   * the alpha part comes from alpha_affine (e: s, 1/p.z, G, a, ds) and the pair's depth from alpha_depth;
//   * the distortion terms use the per-pixel products gA = g_dist A, gD = g_dist D, gD2 = g_dist D2 (pixbwd_init_affine):
//         dL_dweight = g_dist (D2 + m^2 A - 2 m D) = gD2 + m (tz - gD),  tz = m gA - gD
//         dL_dz      = w (2 tz dm/dd + g_depth)
//   * u is one FMA chain; 1 / (1 - alpha) needs no select (alpha = 0 gives exactly 1).
// Round 5 (instruction count; the visit is VALU-issue bound):
//   * the median contributor's two extra terms cost one comparison and two selects: u's chain STARTS from g_alpha or from the
//     precomputed g_alpha + g_medw, and dL_dz ends in an FMA onto g_meddepth or 0 (were: compare, two selects, two adds);
//   * the background term uses the per-pixel product Tf_bg = T_final dot(bg, g_pix);
//   * dL_dk.z, dL_dl.z and the third row's factor without k.xy, l.xy: with (ax, ay) = dL_dp.xy and c_r = ax r.y - ay r.x for the rows
//     r = Tu, Tv, Tw of the matrix,  n1 = -dL_dk.z = pfy c_w - c_v,  n2 = -dL_dl.z = c_u - pfx c_w,  and
//     pfx n1 + pfy n2 = pfy c_u - pfx c_v  (the c_w terms cancel): 10 operations on shorter chains instead of 11;
struct PixBwdA {
    float T, acc;
    float gA, gD, gD2;            // g_dist * (final_A, final_D, final_D2)
    float g_pix[3], g_depth, g_alpha, g_normal[3];
    float g_alpha_med;            // g_alpha + g_medw: where the median contributor's u starts
    float g_meddepth;
    float Tf_bg;                  // T_final * dot(bg, g_pix)
    int last_contributor, med_e;  // med_e: 0-based list index of the median contributor (-1: none)
};

DGS_HD void pixbwd_init_affine(PixBwdA& s, float T_final, float dist1, float dist2, int last, int med_c, const float* gpix,
                               const float* gothers /*8*/, const float* bg)
{
    s.T = T_final;
    s.acc = 0.f;
    for (int c = 0; c < 3; c++) s.g_pix[c] = gpix[c];
    const float g_dist = gothers[6];
    s.gA = g_dist * (1.f - T_final); s.gD = g_dist * dist1; s.gD2 = g_dist * dist2;
    s.g_depth = gothers[0]; s.g_alpha = gothers[1];
    s.g_normal[0] = gothers[2]; s.g_normal[1] = gothers[3]; s.g_normal[2] = gothers[4];
    s.g_meddepth = gothers[5]; s.g_alpha_med = gothers[1] + gothers[7];
    s.Tf_bg = T_final * (bg[0] * gpix[0] + bg[1] * gpix[1] + bg[2] * gpix[2]);
    s.last_contributor = last; s.med_e = med_c - 1;
}

// The `u` of pixbwd_step_affine alone (same expressions): what the entry contributes to the "behind" recurrence acc <- acc + alpha (u - acc).
// Pass 1 of the long-tile backward needs it without the gradient arithmetic.
DGS_HD float pixbwd_u_affine(const PixBwdA& s, bool ok, float depth, int contributor, const Quad& q3, const Quad& q4)
{
    const float c_d = ok ? depth : 1.f;
    const float r_d = fast_rcp(c_d);
    const float m_d = kDepthC1 - kDepthC2 * r_d;
    const bool is_med = ok & (contributor == s.med_e);
    const float tz = m_d * s.gA - s.gD;
    float u = s.gD2 + m_d * (tz - s.gD) + (is_med ? s.g_alpha_med : s.g_alpha);
    u = q3.w * s.g_pix[0] + (q4.x * s.g_pix[1] + (q4.y * s.g_pix[2] + (q3.x * s.g_normal[0] + (q3.y * s.g_normal[1] + (q3.z * s.g_normal[2] + (c_d * s.g_depth + u))))));
    return u;
}

// tuv = (Tu.x, Tu.y, Tv.x, Tv.y); (pfx, pfy) absolute pixel centre.  out[16] in AccSlot order, out2d = dL_dmean2D (rare branch).
DGS_HD void pixbwd_step_affine(PixBwdA& s, const AlphaEval& e, bool ok, bool use3d, float depth, int contributor, float pfx, float pfy,
                               float Twx, float Twy, const Quad& tuv, float opacity, const Quad& q3, const Quad& q4, float* out /*[16]*/,
                               float* out2d /*[2]*/)
{
    const float alpha = ok ? e.alpha : 0.f;
    const float G = ok ? e.G : 0.f;
    const float c_d = ok ? depth : 1.f;
    const bool m3 = ok & use3d;
    const float sx = m3 ? e.sx : 0.f, sy = m3 ? e.sy : 0.f, inv_pz = m3 ? e.inv_pz : 0.f;
    // (one reciprocal for both -- r = 1 / ((1 - alpha) depth), two multiplies recover 1 / (1 - alpha) and 1 / depth -- was built and
    // measured in round 5: 0.257 / 0.256 ms against 0.259 / 0.255 for this form, same lease: three multiplies for one v_rcp is a wash)
    const float inv_1ma = fast_rcp(1.f - alpha);   // alpha <= 0.99: well conditioned; alpha = 0 gives exactly 1
    const float r_d = fast_rcp(c_d);
    s.T = s.T * inv_1ma;
    const float w = alpha * s.T;
    const float m_d = kDepthC1 - kDepthC2 * r_d;
    const float dmd2 = (2.0f * kDepthC2) * (r_d * r_d);     // 2 d(mapped)/d(depth), backward.cu:352
    const bool is_med = ok & (contributor == s.med_e);
    const float tz = m_d * s.gA - s.gD;
    float u = s.gD2 + m_d * (tz - s.gD) + (is_med ? s.g_alpha_med : s.g_alpha);
    u = q3.w * s.g_pix[0] + (q4.x * s.g_pix[1] + (q4.y * s.g_pix[2] + (q3.x * s.g_normal[0] + (q3.y * s.g_normal[1] + (q3.z * s.g_normal[2] + (c_d * s.g_depth + u))))));
    const float d = u - s.acc;
    const float dL_dalpha = d * s.T - inv_1ma * s.Tf_bg;
    s.acc += alpha * d;
    const float dL_dz = w * (tz * dmd2 + s.g_depth) + (is_med ? s.g_meddepth : 0.f);
    out[kAccColor + 0] = w * s.g_pix[0]; out[kAccColor + 1] = w * s.g_pix[1]; out[kAccColor + 2] = w * s.g_pix[2];
    out[kAccNormal + 0] = w * s.g_normal[0]; out[kAccNormal + 1] = w * s.g_normal[1]; out[kAccNormal + 2] = w * s.g_normal[2];
    out[kAccOpacity] = G * dL_dalpha;
    const float nGdG = -(G * (opacity * dL_dalpha));   // -G dL_dG
    const float dsx = nGdG * sx + dL_dz * Twx, dsy = nGdG * sy + dL_dz * Twy;
    const float ax = dsx * inv_pz, ay = dsy * inv_pz;   // dL_dp.xy; dL_dp.z = -(ax s.x + ay s.y)
    const float c_w = ax * Twy - ay * Twx, c_u = ax * tuv.y - ay * tuv.x, c_v = ax * tuv.w - ay * tuv.z;
    const float n1 = pfy * c_w - c_v;               // -dL_dk.z
    const float n2 = c_u - pfx * c_w;               // -dL_dl.z
    const float m3w = (dL_dz - pfy * c_u) + pfx * c_v;
    out[kAccT + 0] = n1 * sx; out[kAccT + 1] = n1 * sy; out[kAccT + 2] = n1;
    out[kAccT + 3] = n2 * sx; out[kAccT + 4] = n2 * sy; out[kAccT + 5] = n2;
    out[kAccT + 6] = m3w * sx; out[kAccT + 7] = m3w * sy; out[kAccT + 8] = m3w;
    const float f2 = (ok & !use3d) ? (2.0f * kInvSqrt2) * nGdG : 0.f;   // dL_dG * (-G * FilterInvSquare) * d, d = ds / sqrt2 (backward.cu:436-443)
    out2d[0] = f2 * e.dxs; out2d[1] = f2 * e.dys;
}

// ---------------------------------------------------------------------------------------------
// Per-surfel backward: computeAABB vjp (backward.cu:599-649), computeTransMat vjp (:451-529),
// quaternion vjp (auxiliary.h:213-257) and SH vjp (backward.cu:20-139).
struct SurfelGrads {
    float dmean3D[3];
    float dscale[2];
    float drot[4];
    float dmean2D[2];   // densification signal, backward.cu:645-648
    float dT[9];        // total dL/dtransMat
};

DGS_HD void sh_backward(int deg, const float* sh, const float* pos, const float* campos, uint32_t flags, const float* dL_dcolor,
                        float* dsh /*[M,3], entries up to (deg+1)^2 written*/, float* dmean /*+=*/)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    float ox = pos[0] - campos[0], oy = pos[1] - campos[1], oz = pos[2] - campos[2];
    float len = sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
    for (int c = 0; c < 3; c++) {
        const float g = ((flags >> c) & 1u) ? 0.f : dL_dcolor[c];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        dsh[c] = C0 * g;
        if (deg > 0) {
            dsh[3 + c] = (-C1 * y) * g; dsh[6 + c] = (C1 * z) * g; dsh[9 + c] = (-C1 * x) * g;
            gx = -C1 * sh[9 + c]; gy = -C1 * sh[3 + c]; gz = C1 * sh[6 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dsh[12 + c] = (C2[0] * xy) * g; dsh[15 + c] = (C2[1] * yz) * g; dsh[18 + c] = (C2[2] * (2.f * zz - xx - yy)) * g;
                dsh[21 + c] = (C2[3] * xz) * g; dsh[24 + c] = (C2[4] * (xx - yy)) * g;
                gx += C2[0] * y * sh[12 + c] + C2[2] * 2.f * -x * sh[18 + c] + C2[3] * z * sh[21 + c] + C2[4] * 2.f * x * sh[24 + c];
                gy += C2[0] * x * sh[12 + c] + C2[1] * z * sh[15 + c] + C2[2] * 2.f * -y * sh[18 + c] + C2[4] * 2.f * -y * sh[24 + c];
                gz += C2[1] * y * sh[15 + c] + C2[2] * 2.f * 2.f * z * sh[18 + c] + C2[3] * x * sh[21 + c];
                if (deg > 2) {
                    dsh[27 + c] = (C3[0] * y * (3.f * xx - yy)) * g; dsh[30 + c] = (C3[1] * xy * z) * g;
                    dsh[33 + c] = (C3[2] * y * (4.f * zz - xx - yy)) * g; dsh[36 + c] = (C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                    dsh[39 + c] = (C3[4] * x * (4.f * zz - xx - yy)) * g; dsh[42 + c] = (C3[5] * z * (xx - yy)) * g;
                    dsh[45 + c] = (C3[6] * x * (xx - 3.f * yy)) * g;
                    gx += C3[0] * sh[27 + c] * 3.f * 2.f * xy + C3[1] * sh[30 + c] * yz + C3[2] * sh[33 + c] * -2.f * xy +
                          C3[3] * sh[36 + c] * -3.f * 2.f * xz + C3[4] * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) +
                          C3[5] * sh[42 + c] * 2.f * xz + C3[6] * sh[45 + c] * 3.f * (xx - yy);
                    gy += C3[0] * sh[27 + c] * 3.f * (xx - yy) + C3[1] * sh[30 + c] * xz + C3[2] * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) +
                          C3[3] * sh[36 + c] * -3.f * 2.f * yz + C3[4] * sh[39 + c] * -2.f * xy + C3[5] * sh[42 + c] * -2.f * yz +
                          C3[6] * sh[45 + c] * -3.f * 2.f * xy;
                    gz += C3[1] * sh[30 + c] * xy + C3[2] * sh[33 + c] * 4.f * 2.f * yz + C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) +
                          C3[4] * sh[39 + c] * 4.f * 2.f * xz + C3[5] * sh[42 + c] * (xx - yy);
                }
            }
        }
        ddx += gx * g; ddy += gy * g; ddz += gz * g;
    }
    // through dir = v/|v| (auxiliary.h:126-137)
    float sum2 = ox * ox + oy * oy + oz * oz;
    float invs = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invs;
    dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invs;
    dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invs;
}

// `acc` = this surfel's accumulated row from the backward blend (AccSlot layout).
DGS_HD void surfel_backward(const Camera& cam, const float* pos, const float* scale, const float* quat, const SurfelRec& rec,
                            const float* acc, SurfelGrads& g)
{
    const float* vm = cam.view;
    const float* Tu = rec.Tu; const float* Tv = rec.Tv; const float* Tw = rec.Tw;
    float dT[9];
    for (int c = 0; c < 9; c++) dT[c] = acc[kAccT + c];
    // ---- AABB chain: d(centre)/dT applied to the raw screen-space gradient
    {
        const float gmx = acc[kAccMean2D + 0], gmy = acc[kAccMean2D + 1];
        const float d = Tw[0] * Tw[0] + Tw[1] * Tw[1] - Tw[2] * Tw[2];
        const float inv = 1.0f / d;
        const float f[3] = {inv, inv, -inv};
        const float sgn[3] = {1.f, 1.f, -1.f};
        float dLdf_dot_f = 0.f;
        for (int c = 0; c < 3; c++) dLdf_dot_f += (gmx * Tu[c] * Tw[c] + gmy * Tv[c] * Tw[c]) * f[c];
        const float dL_dd = dLdf_dot_f * (-1.0f / d);
        for (int c = 0; c < 3; c++) {
            dT[c] += gmx * f[c] * Tw[c];
            dT[3 + c] += gmy * f[c] * Tw[c];
            dT[6 + c] += gmx * f[c] * Tu[c] + gmy * f[c] * Tv[c] + dL_dd * (sgn[c] * Tw[c] * 2.0f);
        }
    }
    for (int c = 0; c < 9; c++) g.dT[c] = dT[c];
    const float Wh = cam.focal_x * cam.tan_fovx, Hh = cam.focal_y * cam.tan_fovy;  // backward.cu:680-681
    g.dmean2D[0] = dT[2] * Tw[2] * Wh;
    g.dmean2D[1] = dT[5] * Tw[2] * Hh;
    // ---- transMat chain
    const float cx = Wh, cy = Hh;  // backward.cu:565
    float dM0[3], dM1[3], dM2[3];
    dM0[0] = cam.focal_x * dT[0]; dM0[1] = cam.focal_y * dT[3]; dM0[2] = cx * dT[0] + cy * dT[3] + dT[6];
    dM1[0] = cam.focal_x * dT[1]; dM1[1] = cam.focal_y * dT[4]; dM1[2] = cx * dT[1] + cy * dT[4] + dT[7];
    dM2[0] = cam.focal_x * dT[2]; dM2[1] = cam.focal_y * dT[5]; dM2[2] = cx * dT[2] + cy * dT[5] + dT[8];
    float dRS0[3], dRS1[3], dpw[3], dtn[3];
    view_rotate_T(vm, dM0, dRS0);
    view_rotate_T(vm, dM1, dRS1);
    view_rotate_T(vm, dM2, dpw);
    view_rotate_T(vm, acc + kAccNormal, dtn);
    float c0[3], c1[3], c2[3];
    quat_columns(quat, c0, c1, c2);
    // dual-visible flip (backward.cu:509-514): recompute the sign from W*R2 . p_view
    float tn[3], wp[3];
    view_rotate(vm, c2, tn);
    view_rotate(vm, pos, wp);
    const float pvx = wp[0] + vm[12], pvy = wp[1] + vm[13], pvz = wp[2] + vm[14];
    const float cosv = -tn[0] * pvx + -tn[1] * pvy + -tn[2] * pvz;
    const float flip = cosv > 0 ? 1.f : -1.f;
    // v_R columns
    float v0[3] = {dRS0[0] * scale[0], dRS0[1] * scale[0], dRS0[2] * scale[0]};
    float v1[3] = {dRS1[0] * scale[1], dRS1[1] * scale[1], dRS1[2] * scale[1]};
    float v2[3] = {dtn[0] * flip, dtn[1] * flip, dtn[2] * flip};
    {
        float s = 1.0f / sqrtf(quat[3] * quat[3] + quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2]);
        float w = quat[0] * s, x = quat[1] * s, y = quat[2] * s, z = quat[3] * s;
        // vR[c][r] = v{c}[r]
        g.drot[0] = 2.f * (x * (v1[2] - v2[1]) + y * (v2[0] - v0[2]) + z * (v0[1] - v1[0]));
        g.drot[1] = 2.f * (-2.f * x * (v1[1] + v2[2]) + y * (v0[1] + v1[0]) + z * (v0[2] + v2[0]) + w * (v1[2] - v2[1]));
        g.drot[2] = 2.f * (x * (v0[1] + v1[0]) - 2.f * y * (v0[0] + v2[2]) + z * (v1[2] + v2[1]) + w * (v2[0] - v0[2]));
        g.drot[3] = 2.f * (x * (v0[2] + v2[0]) + y * (v1[2] + v2[1]) - 2.f * z * (v0[0] + v1[1]) + w * (v0[1] - v1[0]));
    }
    g.dscale[0] = dRS0[0] * c0[0] + dRS0[1] * c0[1] + dRS0[2] * c0[2];
    g.dscale[1] = dRS1[0] * c1[0] + dRS1[1] * c1[1] + dRS1[2] * c1[2];
    g.dmean3D[0] = dpw[0]; g.dmean3D[1] = dpw[1]; g.dmean3D[2] = dpw[2];
}

}  // namespace dgs
