/*
 * surfel_oracle.c -- CPU restatement of the 2D-Gaussian surfel rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * dynamic-2dgs_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product path never does.
 *
 * PARITY UNPINNED against the reference's own tests: hustvl/Dynamic-2DGS ships no tests,
 * golden vectors or fixtures for this path, and its CUDA sources cannot be
 * built in this image (no nvcc / CUB / cooperative_groups).  The oracle is instead pinned by
 *   (i)   fp64 torch.autograd of an independent dense re-derivation of the forward
 *         (tests/dense_torch_ref.py) against the ORACLE_F64 build of this file, and
 *   (ii)  for the one part of this path the reference also has in Python -- the SH colour evaluation, utils/sh_utils.py
 *         eval_sh -- golden vectors produced by importing that module (tests/golden/make_aux_golden.py,
 *         tests/test_aux_golden.py); the geometry / blending / backward parts exist in the reference only as CUDA and
 *         stay unpinned.
 *
 * Every function cites the reference file:line whose behaviour it restates; paths are relative
 * to /root/reference/submodules/diff-surfel-rasterization/.
 *
 * Build (see oracle/Makefile):
 *   gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC            surfel_oracle.c -o liboracle_f32.so
 *   gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC -DORACLE_F64 ...        -o liboracle_f64.so
 *
 * The f32 build follows the reference's float/double mix (e.g. the mapped depth is evaluated in
 * double and rounded, forward.cu:412); the f64 build evaluates everything in double and exists to
 * validate the hand-derived backward against autograd.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#endif

/* cuda_rasterizer/config.h:15-17, auxiliary.h:18-37 */
#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE 256
#define FILTER_SIZE 0.7071067811865476
#define FILTER_INV_SQUARE (1 / (FILTER_SIZE * FILTER_SIZE))
#define NEAR_PLANE 0.2
#define FAR_PLANE 100.0
#define DEPTH_OFFSET 0
#define ALPHA_OFFSET 1
#define NORMAL_OFFSET 2
#define MIDDEPTH_OFFSET 5
#define DISTORTION_OFFSET 6
#define MEDIAN_WEIGHT_OFFSET 7

/* auxiliary.h:40-58 */
static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
                              (real)0.3731763325901154, (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

typedef struct {
    int P, D, M, W, H;
    int R;              /* num_rendered, rasterizer_impl.cu:281-282 */
    int tiles_x, tiles_y;
    /* geometry state, rasterizer_impl.h GeometryState / rasterizer_impl.cu:155-170 */
    real* depths;       /* [P] */
    int* radii;         /* [P] */
    real* means2D;      /* [P,2] */
    real* transMat;     /* [P,9] */
    real* normal_opacity; /* [P,4] */
    real* rgb;          /* [P,3] */
    unsigned char* clamped; /* [P,3] */
    uint32_t* tiles_touched; /* [P] */
    uint32_t* point_offsets; /* [P] inclusive scan */
    /* binning state, rasterizer_impl.cu:183-194 */
    uint32_t* point_list; /* [R] */
    uint32_t* point_tile; /* [R] tile id of the sorted entry (key >> 32) */
    uint32_t* ranges;     /* [T,2] */
    /* image state, rasterizer_impl.cu:172-179 */
    real* final_T;        /* [3,H*W]: T, dist1, dist2 */
    uint32_t* n_contrib;  /* [2,H*W]: last, median */
} OracleState;

static void cross3(const real* a, const real* b, real* r) /* auxiliary.h:151-157 */
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

/* auxiliary.h:188-210: quaternion (r,x,y,z) -> rotation, columns R[c][row] */
static void quat_to_rotmat(const real* q, real R[3][3])
{
    real s = (real)1 / R_SQRT(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    R[0][0] = 1 - 2 * (y * y + z * z);
    R[0][1] = 2 * (x * y + w * z);
    R[0][2] = 2 * (x * z - w * y);
    R[1][0] = 2 * (x * y - w * z);
    R[1][1] = 1 - 2 * (x * x + z * z);
    R[1][2] = 2 * (y * z + w * x);
    R[2][0] = 2 * (x * z + w * y);
    R[2][1] = 2 * (y * z - w * x);
    R[2][2] = 1 - 2 * (x * x + y * y);
}

/* auxiliary.h:213-257: vjp of quat_to_rotmat w.r.t. the NORMALISED quaternion (no grad through
 * the normalisation), v_R[c][row] column-major, result in (r,x,y,z) order. */
static void quat_to_rotmat_vjp(const real* q, real vR[3][3], real* vq)
{
    real s = (real)1 / R_SQRT(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    vq[0] = 2 * (x * (vR[1][2] - vR[2][1]) + y * (vR[2][0] - vR[0][2]) + z * (vR[0][1] - vR[1][0]));
    vq[1] = 2 * (-2 * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) + z * (vR[0][2] + vR[2][0]) +
                 w * (vR[1][2] - vR[2][1]));
    vq[2] = 2 * (x * (vR[0][1] + vR[1][0]) - 2 * y * (vR[0][0] + vR[2][2]) + z * (vR[1][2] + vR[2][1]) +
                 w * (vR[2][0] - vR[0][2]));
    vq[3] = 2 * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) - 2 * z * (vR[0][0] + vR[1][1]) +
                 w * (vR[0][1] - vR[1][0]));
}

/* W * v with W = viewmat[:3,:3] in the reference's flat (column-major) storage,
 * forward.cu:79-83 / auxiliary.h:98-106 */
static void view_rot(const real* vm, const real* v, real* r)
{
    r[0] = vm[0] * v[0] + vm[4] * v[1] + vm[8] * v[2];
    r[1] = vm[1] * v[0] + vm[5] * v[1] + vm[9] * v[2];
    r[2] = vm[2] * v[0] + vm[6] * v[1] + vm[10] * v[2];
}

/* W^T * v, auxiliary.h:108-116 */
static void view_rot_T(const real* vm, const real* v, real* r)
{
    r[0] = vm[0] * v[0] + vm[1] * v[1] + vm[2] * v[2];
    r[1] = vm[4] * v[0] + vm[5] * v[1] + vm[6] * v[2];
    r[2] = vm[8] * v[0] + vm[9] * v[1] + vm[10] * v[2];
}

/* auxiliary.h:64-74 (getRect). max_radius is an int in the reference signature. */
static void get_rect(real px, real py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    int v;
    v = (int)((px - max_radius) / BLOCK_X); v = v < 0 ? 0 : v; rmin[0] = v < gx ? v : gx;
    v = (int)((py - max_radius) / BLOCK_Y); v = v < 0 ? 0 : v; rmin[1] = v < gy ? v : gy;
    v = (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X); v = v < 0 ? 0 : v; rmax[0] = v < gx ? v : gx;
    v = (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y); v = v < 0 ? 0 : v; rmax[1] = v < gy ? v : gy;
}

/* forward.cu:20-71 computeColorFromSH */
static void sh_to_rgb(int idx, int deg, int M, const real* means, const real* campos, const real* shs,
                      unsigned char* clamped, real* out)
{
    real dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    real len = R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    real x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    const real* sh = shs + (size_t)idx * M * 3;
    for (int c = 0; c < 3; c++) {
        real r = SH_C0 * sh[c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] +
                    SH_C2[2] * (2 * zz - xx - yy) * sh[18 + c] + SH_C2[3] * xz * sh[21 + c] +
                    SH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3 * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                        SH_C3[2] * y * (4 * zz - xx - yy) * sh[33 + c] +
                        SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[36 + c] +
                        SH_C3[4] * x * (4 * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                        SH_C3[6] * x * (xx - 3 * yy) * sh[45 + c];
                }
            }
        }
        r += (real)0.5;
        clamped[3 * idx + c] = (r < 0);
        out[c] = r > 0 ? r : 0;
    }
}

/* forward.cu:166-260 preprocessCUDA (+ computeTransMat :75-128, computeAABB :133-163,
 * in_frustum auxiliary.h:160-185) */
static void preprocess_fwd(OracleState* st, const real* means3D, const real* scales, const real* rotations,
                           const real* opacities, const real* shs, const real* colors_precomp, const real* vm,
                           const real* campos, real tanfovx, real tanfovy)
{
    const int P = st->P, W = st->W, H = st->H;
    const real focal_y = H / (2 * tanfovy), focal_x = W / (2 * tanfovx); /* rasterizer_impl.cu:223-224 */
    const real cx = (real)W / 2, cy = (real)H / 2;                       /* forward.cu:208 */
    for (int idx = 0; idx < P; idx++) {
        st->radii[idx] = 0;
        st->tiles_touched[idx] = 0;
        const real* pw = means3D + 3 * idx;
        real pv[3];
        pv[0] = vm[0] * pw[0] + vm[4] * pw[1] + vm[8] * pw[2] + vm[12];
        pv[1] = vm[1] * pw[0] + vm[5] * pw[1] + vm[9] * pw[2] + vm[13];
        pv[2] = vm[2] * pw[0] + vm[6] * pw[1] + vm[10] * pw[2] + vm[14];
        if (pv[2] <= (real)0.2) continue; /* auxiliary.h:174 */

        real R[3][3];
        quat_to_rotmat(rotations + 4 * idx, R);
        real sx = scales[2 * idx], sy = scales[2 * idx + 1];
        real RS0[3] = {R[0][0] * sx, R[0][1] * sx, R[0][2] * sx};
        real RS1[3] = {R[1][0] * sy, R[1][1] * sy, R[1][2] * sy};
        real M0[3], M1[3], tn[3];
        view_rot(vm, RS0, M0);
        view_rot(vm, RS1, M1);
        view_rot(vm, R[2], tn);
        real cosv = -tn[0] * pv[0] + -tn[1] * pv[1] + -tn[2] * pv[2];
        if (cosv == 0) continue; /* forward.cu:101-103 */
        real mult = cosv > 0 ? (real)1 : (real)-1;
        tn[0] *= mult; tn[1] *= mult; tn[2] *= mult;
        real* T = st->transMat + 9 * idx; /* forward.cu:111-126 */
        T[0] = focal_x * M0[0] + cx * M0[2]; T[1] = focal_x * M1[0] + cx * M1[2]; T[2] = focal_x * pv[0] + cx * pv[2];
        T[3] = focal_y * M0[1] + cy * M0[2]; T[4] = focal_y * M1[1] + cy * M1[2]; T[5] = focal_y * pv[1] + cy * pv[2];
        T[6] = M0[2]; T[7] = M1[2]; T[8] = pv[2];

        /* computeAABB forward.cu:133-163 */
        real d = T[6] * T[6] + T[7] * T[7] - T[8] * T[8];
        if (d == 0) continue;
        real inv = (real)1 / d;
        real f[3] = {inv, inv, -inv};
        real px = f[0] * (T[0] * T[6]) + f[1] * (T[1] * T[7]) + f[2] * (T[2] * T[8]);
        real py = f[0] * (T[3] * T[6]) + f[1] * (T[4] * T[7]) + f[2] * (T[5] * T[8]);
        real h0x = px * px - (f[0] * (T[0] * T[0]) + f[1] * (T[1] * T[1]) + f[2] * (T[2] * T[2]));
        real h0y = py * py - (f[0] * (T[3] * T[3]) + f[1] * (T[4] * T[4]) + f[2] * (T[5] * T[5]));
        real ex = R_SQRT(h0x > 0 ? h0x : 0), ey = R_SQRT(h0y > 0 ? h0y : 0);
        real emax = ex > ey ? ex : ey;
        /* forward.cu:231 : ceil(3.f * max(max(ex,ey), FilterSize)), FilterSize is a double literal */
        double em = (double)emax > FILTER_SIZE ? (double)emax : FILTER_SIZE;
        real radius = (real)ceil(3.0 * em);
        int rmin[2], rmax[2];
        get_rect(px, py, (int)radius, st->tiles_x, st->tiles_y, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (colors_precomp == NULL)
            sh_to_rgb(idx, st->D, st->M, means3D, campos, shs, st->clamped, st->rgb + 3 * idx);
        st->depths[idx] = pv[2];
        st->radii[idx] = (int)radius;
        st->means2D[2 * idx] = px;
        st->means2D[2 * idx + 1] = py;
        st->normal_opacity[4 * idx + 0] = tn[0];
        st->normal_opacity[4 * idx + 1] = tn[1];
        st->normal_opacity[4 * idx + 2] = tn[2];
        st->normal_opacity[4 * idx + 3] = opacities[idx];
        st->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

typedef struct { uint32_t tile; real depth; uint32_t seq; uint32_t id; } SortEnt;

static int cmp_ent(const void* a, const void* b)
{
    const SortEnt* x = (const SortEnt*)a; const SortEnt* y = (const SortEnt*)b;
    if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
    if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1; /* depth > 0.2: bit order == value order */
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);       /* stable radix sort keeps emission order */
}

/* rasterizer_impl.cu:70-138,278-319: inclusive scan, duplicateWithKeys, stable sort by
 * (tile<<32 | depth bits), identifyTileRanges */
static void bin_and_sort(OracleState* st)
{
    const int P = st->P;
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += st->tiles_touched[i]; st->point_offsets[i] = acc; }
    st->R = (int)acc;
    const int R = st->R, Tn = st->tiles_x * st->tiles_y;
    SortEnt* ents = (SortEnt*)malloc(sizeof(SortEnt) * (size_t)(R > 0 ? R : 1));
    for (int idx = 0; idx < P; idx++) {
        if (st->radii[idx] > 0) {
            uint32_t off = idx == 0 ? 0 : st->point_offsets[idx - 1];
            int rmin[2], rmax[2];
            get_rect(st->means2D[2 * idx], st->means2D[2 * idx + 1], st->radii[idx], st->tiles_x, st->tiles_y, rmin, rmax);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    ents[off].tile = (uint32_t)(y * st->tiles_x + x);
                    ents[off].depth = st->depths[idx];
                    ents[off].seq = off;
                    ents[off].id = (uint32_t)idx;
                    off++;
                }
        }
    }
    qsort(ents, (size_t)R, sizeof(SortEnt), cmp_ent);
    st->point_list = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    st->point_tile = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    for (int i = 0; i < R; i++) { st->point_list[i] = ents[i].id; st->point_tile[i] = ents[i].tile; }
    memset(st->ranges, 0, sizeof(uint32_t) * 2 * (size_t)Tn); /* rasterizer_impl.cu:311 */
    for (int i = 0; i < R; i++) {                              /* rasterizer_impl.cu:116-138 */
        uint32_t cur = ents[i].tile;
        if (i == 0) st->ranges[2 * cur] = 0;
        else {
            uint32_t prev = ents[i - 1].tile;
            if (cur != prev) { st->ranges[2 * prev + 1] = (uint32_t)i; st->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) st->ranges[2 * cur + 1] = (uint32_t)R;
    }
    free(ents);
}

/* mapped depth, forward.cu:412 / backward.cu:351 (double expression rounded to float) */
static real mapped_depth(real depth)
{
    return (real)((FAR_PLANE * (double)depth - FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * (double)depth));
}

/* forward.cu:265-463 renderCUDA, one pixel at a time (the tile-level batching of :331-351 does
 * not change per-pixel results). */
static void render_fwd(OracleState* st, const real* feat, const real* bg, real* out_color, real* out_others)
{
    const int W = st->W, H = st->H, HW = W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < st->tiles_x * st->tiles_y; tile++) {
        const int tx = tile % st->tiles_x, ty = tile / st->tiles_x;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const int pix_id = W * pyi + pxi;
                const real pfx = (real)pxi + (real)0.5, pfy = (real)pyi + (real)0.5;
                real T = 1, C[3] = {0, 0, 0}, D = 0, N[3] = {0, 0, 0};
                real dist1 = 0, dist2 = 0, distortion = 0, median_depth = 0, median_weight = 0;
                float median_contributor = -1; /* forward.cu:326, stored through a float */
                uint32_t contributor = 0, last_contributor = 0;
                for (uint32_t e = r0; e < r1; e++) {
                    contributor++;
                    const uint32_t id = st->point_list[e];
                    const real* Tm = st->transMat + 9 * id;
                    const real* Tu = Tm; const real* Tv = Tm + 3; const real* Tw = Tm + 6;
                    real k[3] = {-Tu[0] + pfx * Tw[0], -Tu[1] + pfx * Tw[1], -Tu[2] + pfx * Tw[2]};
                    real l[3] = {-Tv[0] + pfy * Tw[0], -Tv[1] + pfy * Tw[1], -Tv[2] + pfy * Tw[2]};
                    real p[3];
                    cross3(k, l, p);
                    if (p[2] == 0) continue;                           /* :369 */
                    real sx = p[0] / p[2], sy = p[1] / p[2];
                    real rho3d = sx * sx + sy * sy;
                    real dx = st->means2D[2 * id] - pfx, dy = st->means2D[2 * id + 1] - pfy;
                    real rho2d = (real)(FILTER_INV_SQUARE * (double)(dx * dx + dy * dy)); /* :381 */
                    real rho = rho3d < rho2d ? rho3d : rho2d;
                    real depth = (rho3d <= rho2d) ? (sx * Tw[0] + sy * Tw[1]) + Tw[2] : Tw[2];
                    if ((double)depth < NEAR_PLANE) continue;          /* :385 */
                    const real* no = st->normal_opacity + 4 * id;
                    real power = (real)-0.5 * rho;
                    if (power > 0) continue;
                    real alpha = no[3] * R_EXP(power);
                    if (alpha > (real)0.99) alpha = (real)0.99;
                    if (alpha < (real)1 / (real)255) continue;
                    real test_T = T * (1 - alpha);
                    if (test_T < (real)0.0001) break; /* done = true; :400-405 */
                    real A = 1 - T;
                    real m = mapped_depth(depth);
                    /* literal operand order of forward.cu:413-433: `x * alpha * T` is (x * alpha) * T, not x * (alpha * T) */
                    real error = m * m * A + dist2 - 2 * m * dist1;
                    distortion += error * alpha * T;
                    if ((double)T > 0.5) { median_depth = depth; median_weight = alpha * T; median_contributor = (float)contributor; }
                    for (int ch = 0; ch < 3; ch++) N[ch] += no[ch] * alpha * T;
                    D += depth * alpha * T;
                    dist1 += m * alpha * T;
                    dist2 += m * m * alpha * T;
                    for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * id + ch] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                st->final_T[pix_id] = T;
                st->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
                /* float -> uint32 store of -1 is UB in C; CUDA saturates to 0. Either way the backward
                 * test `contributor == median_contributor - 1` can never fire (backward.cu:353). */
                st->n_contrib[pix_id + HW] = median_contributor < 0 ? 0u : (uint32_t)median_contributor;
                st->final_T[pix_id + HW] = dist1;
                st->final_T[pix_id + 2 * HW] = dist2;
                out_others[pix_id + DEPTH_OFFSET * HW] = D;
                out_others[pix_id + ALPHA_OFFSET * HW] = 1 - T;
                for (int ch = 0; ch < 3; ch++) out_others[pix_id + (NORMAL_OFFSET + ch) * HW] = N[ch];
                out_others[pix_id + MIDDEPTH_OFFSET * HW] = median_depth;
                out_others[pix_id + DISTORTION_OFFSET * HW] = distortion;
                out_others[pix_id + MEDIAN_WEIGHT_OFFSET * HW] = median_weight;
            }
    }
}

/* Test aid: replay render_fwd for ONE pixel and list the entries it blends -- 1-based contributor index, then (depth, weight
 * alpha*T, T before the blend) per entry.  The two median channels of the allmap are the depth / weight of the LAST entry
 * blended while T > 0.5 (forward.cu:421-425), a discontinuous pick: a pixel whose T sits on 0.5 to rounding takes the
 * neighbouring contributor under any other fp32 evaluation order.  The parity tests use this trace to show that a median
 * that differs from the oracle's is exactly that (tests/gpu_utils.py median_flips).  Same arithmetic as render_fwd. */
int oracle_pixel_trace(const OracleState* st, int pxi, int pyi, int cap, uint32_t* out_contrib, real* out_vals /* [cap][3] */)
{
    if (pxi < 0 || pyi < 0 || pxi >= st->W || pyi >= st->H) return -1;
    const int tile = (pyi / BLOCK_Y) * st->tiles_x + pxi / BLOCK_X;
    const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
    const real pfx = (real)pxi + (real)0.5, pfy = (real)pyi + (real)0.5;
    real T = 1;
    uint32_t contributor = 0;
    int n = 0;
    for (uint32_t e = r0; e < r1; e++) {
        contributor++;
        const uint32_t id = st->point_list[e];
        const real* Tm = st->transMat + 9 * id;
        const real* Tu = Tm; const real* Tv = Tm + 3; const real* Tw = Tm + 6;
        real k[3] = {-Tu[0] + pfx * Tw[0], -Tu[1] + pfx * Tw[1], -Tu[2] + pfx * Tw[2]};
        real l[3] = {-Tv[0] + pfy * Tw[0], -Tv[1] + pfy * Tw[1], -Tv[2] + pfy * Tw[2]};
        real p[3];
        cross3(k, l, p);
        if (p[2] == 0) continue;
        real sx = p[0] / p[2], sy = p[1] / p[2];
        real rho3d = sx * sx + sy * sy;
        real dx = st->means2D[2 * id] - pfx, dy = st->means2D[2 * id + 1] - pfy;
        real rho2d = (real)(FILTER_INV_SQUARE * (double)(dx * dx + dy * dy));
        real rho = rho3d < rho2d ? rho3d : rho2d;
        real depth = (rho3d <= rho2d) ? (sx * Tw[0] + sy * Tw[1]) + Tw[2] : Tw[2];
        if ((double)depth < NEAR_PLANE) continue;
        real power = (real)-0.5 * rho;
        if (power > 0) continue;
        real alpha = st->normal_opacity[4 * id + 3] * R_EXP(power);
        if (alpha > (real)0.99) alpha = (real)0.99;
        if (alpha < (real)1 / (real)255) continue;
        real test_T = T * (1 - alpha);
        if (test_T < (real)0.0001) break;
        if (n < cap) {
            out_contrib[n] = contributor;
            out_vals[3 * n] = depth; out_vals[3 * n + 1] = alpha * T; out_vals[3 * n + 2] = T;
        }
        n++;
        T = test_T;
    }
    return n;
}

/* Gradient accumulators are double even in the f32 build: the reference sums fp32 atomics in a
 * non-deterministic order (backward.cu:345-446); the order-free double sum is the centre of that
 * noise band. */
typedef struct {
    double* dT;     /* [P,9] */
    double* dmean2D;/* [P,2] */
    double* dnormal;/* [P,3] */
    double* dopac;  /* [P]   */
    double* dcolor; /* [P,3] */
} GradAcc;

static inline void acc_add(double* p, double v)
{
#pragma omp atomic
    *p += v;
}

/* backward.cu:143-449 renderCUDA (backward) */
static void render_bwd(const OracleState* st, const real* feat, const real* bg, const real* dL_dpix,
                       const real* dL_dothers, GradAcc* g)
{
    const int W = st->W, H = st->H, HW = W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < st->tiles_x * st->tiles_y; tile++) {
        const int tx = tile % st->tiles_x, ty = tile / st->tiles_x;
        const uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        if (r1 == r0) continue;
        /* tile-local partial sums [entry][18] (dcolor 3, dnormal 3, dT 9, dopac 1, dmean2D 2), flushed once
         * per (tile, entry) below: same sums, 256x fewer atomics on many-core hosts */
        double* loc = (double*)calloc((size_t)(r1 - r0) * 18, sizeof(double));
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const int pix_id = W * pyi + pxi;
                const real pfx = (real)pxi + (real)0.5, pfy = (real)pyi + (real)0.5;
                const real T_final = st->final_T[pix_id];
                real T = T_final;
                const int last_contributor = (int)st->n_contrib[pix_id];
                const int median_contributor = (int)st->n_contrib[pix_id + HW];
                real accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
                real dL_dpixel[3] = {dL_dpix[pix_id], dL_dpix[HW + pix_id], dL_dpix[2 * HW + pix_id]};
                const real dL_ddepth = dL_dothers[DEPTH_OFFSET * HW + pix_id];
                const real dL_daccum = dL_dothers[ALPHA_OFFSET * HW + pix_id];
                const real dL_dreg = dL_dothers[DISTORTION_OFFSET * HW + pix_id];
                const real dL_dnormal2D[3] = {dL_dothers[(NORMAL_OFFSET + 0) * HW + pix_id],
                                              dL_dothers[(NORMAL_OFFSET + 1) * HW + pix_id],
                                              dL_dothers[(NORMAL_OFFSET + 2) * HW + pix_id]};
                const real dL_dmedian_depth = dL_dothers[MIDDEPTH_OFFSET * HW + pix_id];
                const real dL_dmax_dweight = dL_dothers[MEDIAN_WEIGHT_OFFSET * HW + pix_id];
                real last_depth = 0, last_normal[3] = {0, 0, 0}, accum_depth_rec = 0, accum_alpha_rec = 0;
                real accum_normal_rec[3] = {0, 0, 0};
                const real final_D = st->final_T[pix_id + HW], final_D2 = st->final_T[pix_id + 2 * HW];
                const real final_A = 1 - T_final;
                real last_dL_dT = 0, last_alpha = 0;
                real bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];

                /* contributor index counts down from the list length, backward.cu:197,276-279 */
                for (int contributor = (int)(r1 - r0) - 1; contributor >= 0; contributor--) {
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = st->point_list[r0 + (uint32_t)contributor];
                    double* la = loc + (size_t)contributor * 18;
                    const real* Tm = st->transMat + 9 * id;
                    const real* Tu = Tm; const real* Tv = Tm + 3; const real* Tw = Tm + 6;
                    real k[3] = {-Tu[0] + pfx * Tw[0], -Tu[1] + pfx * Tw[1], -Tu[2] + pfx * Tw[2]};
                    real l[3] = {-Tv[0] + pfy * Tw[0], -Tv[1] + pfy * Tw[1], -Tv[2] + pfy * Tw[2]};
                    real p[3];
                    cross3(k, l, p);
                    if (p[2] == 0) continue;
                    real sx = p[0] / p[2], sy = p[1] / p[2];
                    real rho3d = sx * sx + sy * sy;
                    real dx = st->means2D[2 * id] - pfx, dy = st->means2D[2 * id + 1] - pfy;
                    real rho2d = (real)(FILTER_INV_SQUARE * (double)(dx * dx + dy * dy));
                    real rho = rho3d < rho2d ? rho3d : rho2d;
                    real c_d = (rho3d <= rho2d) ? (sx * Tw[0] + sy * Tw[1]) + Tw[2] : Tw[2];
                    if ((double)c_d < NEAR_PLANE) continue;
                    const real* no = st->normal_opacity + 4 * id;
                    real power = (real)-0.5 * rho;
                    if (power > 0) continue;
                    const real G = R_EXP(power);
                    real alpha = no[3] * G;
                    if (alpha > (real)0.99) alpha = (real)0.99;
                    if (alpha < (real)1 / (real)255) continue;

                    T = T / (1 - alpha);                          /* :325 */
                    const real dchannel_dcolor = alpha * T;
                    real dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ch++) {             /* :333-346 */
                        const real c = feat[3 * id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dL_dpixel[ch];
                        la[ch] += (double)(dchannel_dcolor * dL_dpixel[ch]);
                    }
                    real dL_dz = 0, dL_dweight = 0;
                    const real m_d = mapped_depth(c_d);
                    const real dmd_dd = (real)((FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * (double)c_d * (double)c_d));
                    if (contributor == median_contributor - 1) { dL_dz += dL_dmedian_depth; dL_dweight += dL_dmax_dweight; }
                    dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg; /* :362 */
                    dL_dalpha += dL_dweight - last_dL_dT;
                    last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                    const real dL_dmd = 2 * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                    dL_dz += dL_dmd * dmd_dd;
                    accum_depth_rec = last_alpha * last_depth + (1 - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                    accum_alpha_rec = last_alpha * 1 + (1 - last_alpha) * accum_alpha_rec;
                    dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                    for (int ch = 0; ch < 3; ch++) {             /* :379-384 */
                        accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1 - last_alpha) * accum_normal_rec[ch];
                        last_normal[ch] = no[ch];
                        dL_dalpha += (no[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
                        la[3 + ch] += (double)(alpha * T * dL_dnormal2D[ch]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1 - alpha)) * bg_dot_dpixel; /* :393-396 */
                    const real dL_dG = no[3] * dL_dalpha;
                    dL_dz += alpha * T * dL_ddepth;

                    if (rho3d <= rho2d) {                         /* :405-435 */
                        real dL_ds[2] = {dL_dG * -G * sx + dL_dz * Tw[0], dL_dG * -G * sy + dL_dz * Tw[1]};
                        real dz_dTw[3] = {sx, sy, 1};
                        real dsx_pz = dL_ds[0] / p[2], dsy_pz = dL_ds[1] / p[2];
                        real dL_dp[3] = {dsx_pz, dsy_pz, -(dsx_pz * sx + dsy_pz * sy)};
                        real dL_dk[3], dL_dl[3];
                        cross3(l, dL_dp, dL_dk);
                        cross3(dL_dp, k, dL_dl);
                        for (int c = 0; c < 3; c++) {
                            la[6 + c] += (double)(-dL_dk[c]);
                            la[9 + c] += (double)(-dL_dl[c]);
                            la[12 + c] += (double)(pfx * dL_dk[c] + pfy * dL_dl[c] + dL_dz * dz_dTw[c]);
                        }
                    } else {                                      /* :436-443 */
                        const real dG_ddelx = (real)((double)(-G) * FILTER_INV_SQUARE * (double)dx);
                        const real dG_ddely = (real)((double)(-G) * FILTER_INV_SQUARE * (double)dy);
                        la[16] += (double)(dL_dG * dG_ddelx);
                        la[17] += (double)(dL_dG * dG_ddely);
                        la[14] += (double)dL_dz;
                    }
                    la[15] += (double)(G * dL_dalpha); /* :446 */
                }
            }
        for (uint32_t e = 0; e < r1 - r0; e++) {
            const double* la = loc + (size_t)e * 18;
            const uint32_t id = st->point_list[r0 + e];
            for (int c = 0; c < 3; c++) { if (la[c] != 0) acc_add(&g->dcolor[3 * id + c], la[c]); if (la[3 + c] != 0) acc_add(&g->dnormal[3 * id + c], la[3 + c]); }
            for (int c = 0; c < 9; c++) if (la[6 + c] != 0) acc_add(&g->dT[9 * id + c], la[6 + c]);
            if (la[15] != 0) acc_add(&g->dopac[id], la[15]);
            if (la[16] != 0) acc_add(&g->dmean2D[2 * id], la[16]);
            if (la[17] != 0) acc_add(&g->dmean2D[2 * id + 1], la[17]);
        }
        free(loc);
    }
}

/* backward.cu:20-139 computeColorFromSH (backward) */
static void sh_bwd(int idx, int deg, int M, const real* means, const real* campos, const real* shs,
                   const unsigned char* clamped, const real* dL_dcolor, real* dL_dmeans, real* dL_dshs)
{
    real dorig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    real len = R_SQRT(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
    real x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
    const real* sh = shs + (size_t)idx * M * 3;
    real* dsh = dL_dshs + (size_t)idx * M * 3;
    real dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0 : 1);
    real ddir[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        real gx = 0, gy = 0, gz = 0; /* dRGBdx/dy/dz for this channel */
        dsh[c] = SH_C0 * dRGB[c];
        if (deg > 0) {
            dsh[3 + c] = (-SH_C1 * y) * dRGB[c];
            dsh[6 + c] = (SH_C1 * z) * dRGB[c];
            dsh[9 + c] = (-SH_C1 * x) * dRGB[c];
            gx = -SH_C1 * sh[9 + c];
            gy = -SH_C1 * sh[3 + c];
            gz = SH_C1 * sh[6 + c];
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dsh[12 + c] = (SH_C2[0] * xy) * dRGB[c];
                dsh[15 + c] = (SH_C2[1] * yz) * dRGB[c];
                dsh[18 + c] = (SH_C2[2] * (2 * zz - xx - yy)) * dRGB[c];
                dsh[21 + c] = (SH_C2[3] * xz) * dRGB[c];
                dsh[24 + c] = (SH_C2[4] * (xx - yy)) * dRGB[c];
                gx += SH_C2[0] * y * sh[12 + c] + SH_C2[2] * 2 * -x * sh[18 + c] + SH_C2[3] * z * sh[21 + c] + SH_C2[4] * 2 * x * sh[24 + c];
                gy += SH_C2[0] * x * sh[12 + c] + SH_C2[1] * z * sh[15 + c] + SH_C2[2] * 2 * -y * sh[18 + c] + SH_C2[4] * 2 * -y * sh[24 + c];
                gz += SH_C2[1] * y * sh[15 + c] + SH_C2[2] * 2 * 2 * z * sh[18 + c] + SH_C2[3] * x * sh[21 + c];
                if (deg > 2) {
                    dsh[27 + c] = (SH_C3[0] * y * (3 * xx - yy)) * dRGB[c];
                    dsh[30 + c] = (SH_C3[1] * xy * z) * dRGB[c];
                    dsh[33 + c] = (SH_C3[2] * y * (4 * zz - xx - yy)) * dRGB[c];
                    dsh[36 + c] = (SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy)) * dRGB[c];
                    dsh[39 + c] = (SH_C3[4] * x * (4 * zz - xx - yy)) * dRGB[c];
                    dsh[42 + c] = (SH_C3[5] * z * (xx - yy)) * dRGB[c];
                    dsh[45 + c] = (SH_C3[6] * x * (xx - 3 * yy)) * dRGB[c];
                    gx += SH_C3[0] * sh[27 + c] * 3 * 2 * xy + SH_C3[1] * sh[30 + c] * yz + SH_C3[2] * sh[33 + c] * -2 * xy +
                          SH_C3[3] * sh[36 + c] * -3 * 2 * xz + SH_C3[4] * sh[39 + c] * (-3 * xx + 4 * zz - yy) +
                          SH_C3[5] * sh[42 + c] * 2 * xz + SH_C3[6] * sh[45 + c] * 3 * (xx - yy);
                    gy += SH_C3[0] * sh[27 + c] * 3 * (xx - yy) + SH_C3[1] * sh[30 + c] * xz +
                          SH_C3[2] * sh[33 + c] * (-3 * yy + 4 * zz - xx) + SH_C3[3] * sh[36 + c] * -3 * 2 * yz +
                          SH_C3[4] * sh[39 + c] * -2 * xy + SH_C3[5] * sh[42 + c] * -2 * yz + SH_C3[6] * sh[45 + c] * -3 * 2 * xy;
                    gz += SH_C3[1] * sh[30 + c] * xy + SH_C3[2] * sh[33 + c] * 4 * 2 * yz +
                          SH_C3[3] * sh[36 + c] * 3 * (2 * zz - xx - yy) + SH_C3[4] * sh[39 + c] * 4 * 2 * xz +
                          SH_C3[5] * sh[42 + c] * (xx - yy);
                }
            }
        }
        ddir[0] += gx * dRGB[c];
        ddir[1] += gy * dRGB[c];
        ddir[2] += gz * dRGB[c];
    }
    /* dnormvdv, auxiliary.h:126-137 */
    real sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
    real invsum32 = (real)1 / R_SQRT(sum2 * sum2 * sum2);
    real vx = dorig[0], vy = dorig[1], vz = dorig[2];
    dL_dmeans[3 * idx + 0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * invsum32;
}

/* backward.cu:599-649 computeAABB (backward) then :533-597 preprocessCUDA (backward) with
 * computeTransMat vjp :451-529.  dL_dT / dL_dmean2D / dL_dnormal / dL_dcolor are the accumulated
 * outputs of render_bwd (already rounded to `real`), modified in place like the reference. */
static void preprocess_bwd(const OracleState* st, const real* means3D, const real* scales, const real* rotations,
                           const real* shs, const real* vm, const real* campos, real tanfovx, real tanfovy,
                           real* dL_dmean2D /*[P,3]*/, const real* dL_dnormal, real* dL_dtransMat, const real* dL_dcolor,
                           real* dL_dsh, real* dL_dmean3D, real* dL_dscale, real* dL_drot)
{
    const int P = st->P, W = st->W, H = st->H;
    const real focal_y = H / (2 * tanfovy), focal_x = W / (2 * tanfovx);
    const real Wh = focal_x * tanfovx, Hh = focal_y * tanfovy; /* backward.cu:680-681 */
    for (int idx = 0; idx < P; idx++) {
        if (!(st->radii[idx] > 0)) continue;
        const real* T = st->transMat + 9 * idx;
        real* dT = dL_dtransMat + 9 * idx;
        /* ---- computeAABB backward ---- */
        {
            const real gmx = dL_dmean2D[3 * idx], gmy = dL_dmean2D[3 * idx + 1];
            real d = T[6] * T[6] + T[7] * T[7] - T[8] * T[8];
            real inv = (real)1 / d;
            real f[3] = {inv, inv, -inv};
            real dT0[3], dT1[3], dT3[3], dLdf[3];
            for (int c = 0; c < 3; c++) {
                dT0[c] = gmx * f[c] * T[6 + c];
                dT1[c] = gmy * f[c] * T[6 + c];
                dT3[c] = gmx * f[c] * T[c] + gmy * f[c] * T[3 + c];
                dLdf[c] = (gmx * T[c] * T[6 + c]) + (gmy * T[3 + c] * T[6 + c]);
            }
            real dL_dd = (real)((double)(dLdf[0] * f[0] + dLdf[1] * f[1] + dLdf[2] * f[2]) * (-1.0 / (double)d));
            const real sgn[3] = {1, 1, -1};
            for (int c = 0; c < 3; c++) dT3[c] += dL_dd * (sgn[c] * T[6 + c] * 2);
            for (int c = 0; c < 3; c++) { dT[c] += dT0[c]; dT[3 + c] += dT1[c]; dT[6 + c] += dT3[c]; }
            real z = T[8];
            dL_dmean2D[3 * idx] = dT[2] * z * Wh;     /* :645-648 densification signal */
            dL_dmean2D[3 * idx + 1] = dT[5] * z * Hh;
        }
        /* ---- computeTransMat backward ---- */
        {
            const real cx = focal_x * tanfovx, cy = focal_y * tanfovy; /* backward.cu:565 */
            real R[3][3];
            quat_to_rotmat(rotations + 4 * idx, R);
            const real sx = scales[2 * idx], sy = scales[2 * idx + 1];
            const real* pw = means3D + 3 * idx;
            real Wp[3], pv[3];
            view_rot(vm, pw, Wp);
            pv[0] = Wp[0] + vm[12]; pv[1] = Wp[1] + vm[13]; pv[2] = Wp[2] + vm[14];
            real dM[3][3]; /* dM[col][row], backward.cu:488-500 */
            for (int j = 0; j < 3; j++) {
                dM[j][0] = focal_x * dT[j];
                dM[j][1] = focal_y * dT[3 + j];
                dM[j][2] = cx * dT[j] + cy * dT[3 + j] + dT[6 + j];
            }
            real dRS0[3], dRS1[3], dpw[3], dtn[3];
            view_rot_T(vm, dM[0], dRS0);
            view_rot_T(vm, dM[1], dRS1);
            view_rot_T(vm, dM[2], dpw);
            view_rot_T(vm, dL_dnormal + 3 * idx, dtn);
            real tn[3];
            view_rot(vm, R[2], tn);
            real cosv = -tn[0] * pv[0] + -tn[1] * pv[1] + -tn[2] * pv[2];
            real mult = cosv > 0 ? (real)1 : (real)-1;
            real dR[3][3];
            for (int c = 0; c < 3; c++) { dR[0][c] = dRS0[c] * sx; dR[1][c] = dRS1[c] * sy; dR[2][c] = dtn[c] * mult; }
            quat_to_rotmat_vjp(rotations + 4 * idx, dR, dL_drot + 4 * idx);
            dL_dscale[2 * idx] = dRS0[0] * R[0][0] + dRS0[1] * R[0][1] + dRS0[2] * R[0][2];
            dL_dscale[2 * idx + 1] = dRS1[0] * R[1][0] + dRS1[1] * R[1][1] + dRS1[2] * R[1][2];
            dL_dmean3D[3 * idx] = dpw[0]; dL_dmean3D[3 * idx + 1] = dpw[1]; dL_dmean3D[3 * idx + 2] = dpw[2];
        }
        if (shs) sh_bwd(idx, st->D, st->M, means3D, campos, shs, st->clamped, dL_dcolor, dL_dmean3D, dL_dsh);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Public API (ctypes).  Mirrors CudaRasterizer::Rasterizer::forward/backward,               */
/* cuda_rasterizer/rasterizer.h:20-87 / rasterizer_impl.cu:198-448.                           */
/* ------------------------------------------------------------------------------------------ */

void oracle_free(OracleState* st)
{
    if (!st) return;
    free(st->depths); free(st->radii); free(st->means2D); free(st->transMat); free(st->normal_opacity);
    free(st->rgb); free(st->clamped); free(st->tiles_touched); free(st->point_offsets);
    free(st->point_list); free(st->point_tile); free(st->ranges); free(st->final_T); free(st->n_contrib);
    free(st);
}

OracleState* oracle_forward(int P, int D, int M, const real* background, int W, int H, const real* means3D,
                            const real* shs, const real* colors_precomp, const real* opacities, const real* scales,
                            const real* rotations, const real* viewmatrix, const real* projmatrix, const real* campos,
                            real tanfovx, real tanfovy, real* out_color, real* out_others, int* radii_out)
{
    (void)projmatrix; /* only feeds a dead expression in in_frustum, auxiliary.h:170-172 */
    OracleState* st = (OracleState*)calloc(1, sizeof(OracleState));
    st->P = P; st->D = D; st->M = M; st->W = W; st->H = H;
    st->tiles_x = (W + BLOCK_X - 1) / BLOCK_X;
    st->tiles_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t Pn = (size_t)(P > 0 ? P : 1), N = (size_t)W * H, Tn = (size_t)st->tiles_x * st->tiles_y;
    st->depths = (real*)calloc(Pn, sizeof(real));
    st->radii = (int*)calloc(Pn, sizeof(int));
    st->means2D = (real*)calloc(Pn * 2, sizeof(real));
    st->transMat = (real*)calloc(Pn * 9, sizeof(real));
    st->normal_opacity = (real*)calloc(Pn * 4, sizeof(real));
    st->rgb = (real*)calloc(Pn * 3, sizeof(real));
    st->clamped = (unsigned char*)calloc(Pn * 3, 1);
    st->tiles_touched = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    st->point_offsets = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    st->ranges = (uint32_t*)calloc(Tn * 2, sizeof(uint32_t));
    st->final_T = (real*)calloc(N * 3, sizeof(real));
    st->n_contrib = (uint32_t*)calloc(N * 2, sizeof(uint32_t));
    memset(out_color, 0, sizeof(real) * 3 * N);  /* rasterize_points.cu:87-88 */
    memset(out_others, 0, sizeof(real) * 8 * N);
    if (P == 0) {                                 /* rasterize_points.cu:106 */
        st->point_list = (uint32_t*)calloc(1, sizeof(uint32_t));
        st->point_tile = (uint32_t*)calloc(1, sizeof(uint32_t));
        return st;
    }
    preprocess_fwd(st, means3D, scales, rotations, opacities, shs, colors_precomp, viewmatrix, campos, tanfovx, tanfovy);
    bin_and_sort(st);
    const real* feat = colors_precomp ? colors_precomp : st->rgb; /* rasterizer_impl.cu:322 */
    render_fwd(st, feat, background, out_color, out_others);
    if (radii_out) memcpy(radii_out, st->radii, sizeof(int) * (size_t)P);
    return st;
}

/* All dL_* outputs are overwritten (the oracle zero-initialises internally, as
 * rasterize_points.cu:194-202 does for the reference). dL_dmean2D is [P,3]. */
void oracle_backward(const OracleState* st, const real* background, const real* means3D, const real* shs,
                     const real* colors_precomp, const real* scales, const real* rotations, const real* viewmatrix,
                     const real* campos, real tanfovx, real tanfovy, const real* dL_dpix, const real* dL_dothers,
                     real* dL_dmean2D, real* dL_dnormal, real* dL_dopacity, real* dL_dcolor, real* dL_dmean3D,
                     real* dL_dtransMat, real* dL_dsh, real* dL_dscale, real* dL_drot)
{
    const int P = st->P;
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    memset(dL_dmean2D, 0, sizeof(real) * 3 * Pn); memset(dL_dnormal, 0, sizeof(real) * 3 * Pn);
    memset(dL_dopacity, 0, sizeof(real) * Pn);    memset(dL_dcolor, 0, sizeof(real) * 3 * Pn);
    memset(dL_dmean3D, 0, sizeof(real) * 3 * Pn); memset(dL_dtransMat, 0, sizeof(real) * 9 * Pn);
    if (st->M > 0) memset(dL_dsh, 0, sizeof(real) * 3 * (size_t)st->M * Pn);
    memset(dL_dscale, 0, sizeof(real) * 2 * Pn);  memset(dL_drot, 0, sizeof(real) * 4 * Pn);
    if (P == 0) return;
    GradAcc g;
    g.dT = (double*)calloc(Pn * 9, sizeof(double));
    g.dmean2D = (double*)calloc(Pn * 2, sizeof(double));
    g.dnormal = (double*)calloc(Pn * 3, sizeof(double));
    g.dopac = (double*)calloc(Pn, sizeof(double));
    g.dcolor = (double*)calloc(Pn * 3, sizeof(double));
    const real* feat = colors_precomp ? colors_precomp : st->rgb;
    render_bwd(st, feat, background, dL_dpix, dL_dothers, &g);
    for (int i = 0; i < P; i++) {
        for (int c = 0; c < 9; c++) dL_dtransMat[9 * i + c] = (real)g.dT[9 * i + c];
        dL_dmean2D[3 * i] = (real)g.dmean2D[2 * i]; dL_dmean2D[3 * i + 1] = (real)g.dmean2D[2 * i + 1];
        for (int c = 0; c < 3; c++) { dL_dnormal[3 * i + c] = (real)g.dnormal[3 * i + c]; dL_dcolor[3 * i + c] = (real)g.dcolor[3 * i + c]; }
        dL_dopacity[i] = (real)g.dopac[i];
    }
    free(g.dT); free(g.dmean2D); free(g.dnormal); free(g.dopac); free(g.dcolor);
    preprocess_bwd(st, means3D, scales, rotations, colors_precomp ? NULL : shs, viewmatrix, campos, tanfovx, tanfovy,
                   dL_dmean2D, dL_dnormal, dL_dtransMat, dL_dcolor, dL_dsh, dL_dmean3D, dL_dscale, dL_drot);
}

/* rasterizer_impl.cu:54-66,141-153 checkFrustum / markVisible */
void oracle_mark_visible(int P, const real* means3D, const real* vm, unsigned char* present)
{
    for (int i = 0; i < P; i++) {
        const real* p = means3D + 3 * i;
        real z = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
        present[i] = !(z <= (real)0.2);
    }
}

/* thread count for the OpenMP loops (bench.py cpu_baseline reports it as "cores") */
#ifdef _OPENMP
#include <omp.h>
#endif
int oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* introspection for stage-by-stage tests */
int oracle_get_int(const OracleState* st, int what)
{
    switch (what) {
    case 0: return st->R;
    case 1: return st->tiles_x;
    case 2: return st->tiles_y;
    case 3: return (int)sizeof(real);
    default: return -1;
    }
}

const void* oracle_get_ptr(const OracleState* st, int what)
{
    switch (what) {
    case 0: return st->depths;
    case 1: return st->radii;
    case 2: return st->means2D;
    case 3: return st->transMat;
    case 4: return st->normal_opacity;
    case 5: return st->rgb;
    case 6: return st->clamped;
    case 7: return st->tiles_touched;
    case 8: return st->point_offsets;
    case 9: return st->point_list;
    case 10: return st->point_tile;
    case 11: return st->ranges;
    case 12: return st->final_T;
    case 13: return st->n_contrib;
    default: return NULL;
    }
}
