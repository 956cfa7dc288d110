// train_ops.hip -- fused SSIM (forward + backward) and brute-force KNN for gfx950 (include/dgs_train_ops.h).
//
// SSIM replaces utils/loss_utils.py:45-76 of the reference: five 11x11 depthwise convolutions plus autograd
// become one kernel per direction.  One 16x16 output tile per 256-thread workgroup; the 26x26 input halo tile
// of both images is staged in LDS once, the Gaussian is applied separably (11 taps horizontally into LDS, 11
// taps vertically from LDS), so each input pixel is read from HBM ~2.6x instead of 121x5.
// KNN replaces pytorch3d.ops.knn_points for the control-node lookup (utils/time_utils.py:950): control nodes
// (<= 1024 x 16 floats) live in LDS, one thread per query point keeps its K best in registers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/dgs_train_ops.h"
#include "node_mlp.h"
#include "wave_reduce.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

constexpr int kR = 5;             // window radius (11 taps)
constexpr int kTW = 54, kTH = 28; // output tile of a 256-thread workgroup
constexpr int kCols = kTW + 2 * kR;   // 64 input columns: one per lane
constexpr int kPV = 7;            // output rows per thread of the vertical pass (4 waves x 7 rows)
constexpr int kPH = 6;            // output columns per thread of the horizontal pass (28 rows x 9 groups = 252 threads)
constexpr int kLds = kCols + 1;
static_assert(kCols == 64 && kTH == 4 * kPV && kTW % kPH == 0 && kTH * (kTW / kPH) <= 256, "thread maps below");

struct Gauss { float w[11]; };

Gauss make_gauss()
{
    // loss_utils.py:33-35: exp(-(x - 5)^2 / (2 * 1.5^2)) normalised, evaluated in float like the reference
    Gauss g;
    float s = 0.f;
    for (int i = 0; i < 11; i++) { g.w[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g.w[i]; }
    for (int i = 0; i < 11; i++) g.w[i] /= s;
    return g;
}

constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

typedef const float __attribute__((address_space(1)))* GlobalF;   // global_load instead of flat_load for pointers that come out
                                                                  // of memory (the per-replay image slots)

// loss = (1 - lambda) * sum(l1 partials) / n + lambda * (1 - sum(ssim partials) / n) + sum(regulariser partials)
// photo = [ssim partial per workgroup (nphoto) | l1 partial per workgroup (nphoto)], reg = [nreg]; one 256-thread workgroup
// Guard of a training step (dgs_step_guard; one thread).  status: [0] skip flag of this step, [1] number of skipped steps so far,
// [2] guarded steps so far.  skip[0] != 0 means "this step must not change anything" (see step_guard_kernel below).
__device__ __forceinline__ void step_guard_body(const int* skip, float* step_count, float* status, float* host_ring, int ring_len, float loss)
{
    const bool sk = skip && skip[0] != 0;
    if (!sk) step_count[0] += 1.0f;
    const float n_skipped = status[1] + (sk ? 1.0f : 0.0f);
    const float n_steps = status[2] + 1.0f;
    status[0] = sk ? 1.0f : 0.0f;
    status[1] = n_skipped;
    status[2] = n_steps;
    if (host_ring) {   // pinned host memory: (step index, skip flag, skipped so far, loss) of the last ring_len steps
        float* e = host_ring + 4 * ((long long)n_steps % ring_len);
        e[1] = sk ? 1.0f : 0.0f;
        e[2] = n_skipped;
        e[3] = loss;   // the step's loss: the host can read a history without a copy kernel per step
        __threadfence_system();
        e[0] = n_steps;   // written last: a reader that sees the index sees the payload
    }
}

struct CombineArgs {
    const float* photo; int nphoto; const float* reg; int nreg; float inv_n; float lambda_dssim; float* out;
    // optional second rider: the step guard, run by the same thread right behind the loss it reports (g_step_count != nullptr)
    const int* g_skip; float* g_step_count; float* g_status; float* g_ring; int g_ring_len;
};

__device__ __forceinline__ void combine_partials(const CombineArgs& c, float (&s_red)[3][4])
{
    float a = 0.f, b = 0.f, r = 0.f;
    {
        float a4[4] = {0, 0, 0, 0}, b4[4] = {0, 0, 0, 0}, c4[4] = {0, 0, 0, 0};
        for (int i = threadIdx.x; i < c.nphoto; i += 1024)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = i + 256 * u;
                a4[u] += k < c.nphoto ? c.photo[k] : 0.f;
                b4[u] += k < c.nphoto ? c.photo[c.nphoto + k] : 0.f;
            }
        for (int i = threadIdx.x; i < c.nreg; i += 1024)
#pragma unroll
            for (int u = 0; u < 4; u++) c4[u] += i + 256 * u < c.nreg ? c.reg[i + 256 * u] : 0.f;
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]); b = (b4[0] + b4[1]) + (b4[2] + b4[3]); r = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    }
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); r += __shfl_xor(r, d, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = a; s_red[1][threadIdx.x >> 6] = b; s_red[2][threadIdx.x >> 6] = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        b = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        r = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
        const float loss = (1.0f - c.lambda_dssim) * b * c.inv_n + c.lambda_dssim * (1.0f - a * c.inv_n) + r;
        c.out[0] = loss;
        if (c.g_step_count) step_guard_body(c.g_skip, c.g_step_count, c.g_status, c.g_ring, c.g_ring_len, loss);
    }
}

__global__ void __launch_bounds__(256) loss_combine_kernel(CombineArgs c)
{
    __shared__ float s_red[3][4];
    combine_partials(c, s_red);
}

// Both SSIM kernels: separable 11-tap window over one 54 x 28 output tile per 256-thread workgroup.
//   pass 1, vertical, straight from global memory: a wave owns 7 output rows, a lane one of the tile's 64 input columns, and loads
//     its 17 input rows with fully coalesced 256-byte wave loads that are all in flight at once -- no staging of the inputs in LDS,
//     no staging barrier (the round-2 kernels staged a 42 x 42 window element by element: 14 dependent memory round trips per
//     workgroup, 39 us for the forward at 800 x 800 x 3; batching those loads gave 27 us, this layout 21: the two passes alone
//     are 11 us, writing the 23 MB of derivative maps the rest);
//   pass 2, horizontal, from LDS: thread = (row, 6 adjacent outputs), 16 reads per quantity.
// Every thread filters several adjacent outputs from one run of inputs held in registers (7 + 10 rows, 6 + 10 columns).  The
// window weights are copied into VGPRs: a VALU instruction with an SGPR source issues at 4.4 instead of 2.5 cycles on gfx950
// (profiles/r03_valu_issue_gfx950.txt).  LDS 36 KB forward / 22 KB backward.
#ifndef DGS_SSIM_DIAG
#define DGS_SSIM_DIAG 0   // development only: 1 no map stores, 2 no global loads, 4 no SSIM formula (tools/diag/loss_timing.py)
#endif
__device__ __forceinline__ void gauss_to_vgprs(const Gauss& g, float (&w)[11])
{
#pragma unroll
    for (int k = 0; k < 11; k++) { w[k] = g.w[k]; asm volatile("" : "+v"(w[k])); }
}

struct SsimFwdArgs {
    int H, W;
    const float* img1; const float* img2;
    float* ssim_sum; float* dm_dmu1; float* dm_ds11; float* dm_ds12; float* l1_sum; float* partial;
    const float* const* img2_slot;
};
constexpr int kSsimFwdLds = 5 * kTH * kLds + 8;   // floats

// workgroup (bx, by, bz) of a (gx, gy, gz) grid; `lds` = kSsimFwdLds floats
__device__ __forceinline__ void ssim_fwd_body(const SsimFwdArgs& A, const Gauss& g, int bx, int by, int bz, int gx, int gy, int gz,
                                              float* __restrict__ lds)
{
    const int H = A.H, W = A.W;
    const float* __restrict__ img1 = A.img1;
    const float* __restrict__ img2 = A.img2_slot ? *A.img2_slot : A.img2;   // indirection: the comparison image is chosen per graph replay by rewriting one pointer
    float* __restrict__ dm_dmu1 = A.dm_dmu1; float* __restrict__ dm_ds11 = A.dm_ds11; float* __restrict__ dm_ds12 = A.dm_ds12;
    float (*s_v)[kTH][kLds] = reinterpret_cast<float (*)[kTH][kLds]>(lds);
    float* s_red = lds + 5 * kTH * kLds;
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    const int x0 = bx * kTW, y0 = by * kTH;
    const size_t plane = (size_t)bz * H * W;
    float w[11];
    gauss_to_vgprs(g, w);
    float l1 = 0.f;
    {
        const GlobalF p1 = (GlobalF)(img1 + plane), p2 = (GlobalF)(img2 + plane);
        const int x = x0 + col - kR;
        const bool xin = x >= 0 && x < W;
        const unsigned xc = (unsigned)min(max(x, 0), W - 1);
        float a[kPV + 10], b[kPV + 10];
#pragma unroll
        for (int j = 0; j < kPV + 10; j++) {   // clamped addresses, zeros (the conv2d padding) selected afterwards
            const int y = y0 + rg * kPV + j - kR;
            const unsigned o = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + xc;
            const bool in = xin && y >= 0 && y < H;
#if DGS_SSIM_DIAG & 2
            const float av = (float)(o & 255) * 0.003f, bv = (float)(o & 127) * 0.005f;
#else
            const float av = p1[o], bv = p2[o];
#endif
            a[j] = in ? av : 0.f;
            b[j] = in ? bv : 0.f;
        }
        const bool mine = col >= kR && col < kR + kTW && xin;   // the tile's own pixels: the mean-|.| term
#pragma unroll
        for (int o = 0; o < kPV; o++)
            if (mine && y0 + rg * kPV + o < H) l1 += fabsf(a[o + kR] - b[o + kR]);
        // two sweeps keep the live set near 100 registers (4 workgroups per CU): means and the cross term from a, b, a b; then the
        // squares in place of a, b
        {
            float ab[kPV + 10];
#pragma unroll
            for (int j = 0; j < kPV + 10; j++) ab[j] = a[j] * b[j];
#pragma unroll
            for (int o = 0; o < kPV; o++) {
                float m1 = 0.f, m2 = 0.f, q12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) { m1 += w[k] * a[o + k]; m2 += w[k] * b[o + k]; q12 += w[k] * ab[o + k]; }
                const int r = rg * kPV + o;
                s_v[0][r][col] = m1; s_v[1][r][col] = m2; s_v[4][r][col] = q12;
            }
        }
#pragma unroll
        for (int j = 0; j < kPV + 10; j++) { a[j] *= a[j]; b[j] *= b[j]; asm volatile("" : "+v"(a[j]), "+v"(b[j])); }
#pragma unroll
        for (int o = 0; o < kPV; o++) {
            float q11 = 0.f, q22 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) { q11 += w[k] * a[o + k]; q22 += w[k] * b[o + k]; }
            const int r = rg * kPV + o;
            s_v[2][r][col] = q11; s_v[3][r][col] = q22;
        }
    }
    __syncthreads();
    float val = 0.f;
    float res[5][kPH];
    if (tid < kTH * (kTW / kPH)) {
        const int r = tid / (kTW / kPH), c0 = (tid - r * (kTW / kPH)) * kPH;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            float v[kPH + 10];
#pragma unroll
            for (int j = 0; j < kPH + 10; j++) v[j] = s_v[q][r][c0 + j];
#pragma unroll
            for (int o = 0; o < kPH; o++) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) t += w[k] * v[o + k];
                res[q][o] = t;
            }
            // one quantity's 16 reads and 66 FMAs at a time (the compiler hoists all 80 reads otherwise, and spills)
            asm volatile("" : "+v"(res[q][0]), "+v"(res[q][1]), "+v"(res[q][2]), "+v"(res[q][3]), "+v"(res[q][4]), "+v"(res[q][5]) :: "memory");
        }
        const int y = y0 + r;
#pragma unroll
        for (int o = 0; o < kPH; o++) {
            const int x = x0 + c0 + o;
            const float mu1 = res[0][o], mu2 = res[1][o], s11 = res[2][o], s22 = res[3][o], s12 = res[4][o];
#if DGS_SSIM_DIAG & 4
            val += mu1 + mu2 + s11 + s22 + s12;
            continue;
#endif
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sg1 = s11 - mu1_sq, sg2 = s22 - mu2_sq, sg12 = s12 - mu12;
            const float A = 2.f * mu12 + kC1, B = 2.f * sg12 + kC2, Cc = mu1_sq + mu2_sq + kC1, D = sg1 + sg2 + kC2;
            const float inv_cd = 1.0f / (Cc * D);
            const float m = A * B * inv_cd;
            if (x < W && y < H) val += m;
            // map = A B / (Cc D) with sigma1^2 = s11 - mu1^2, sigma12 = s12 - mu1 mu2 (loss_utils.py:59-71)
            res[0][o] = (2.f * mu2 * B - 2.f * mu2 * A) * inv_cd - m * (2.f * mu1 / Cc - 2.f * mu1 / D);
            res[1][o] = -m / D;
            res[2][o] = 2.f * A * inv_cd;
        }
    }
    if (dm_dmu1 && !(DGS_SSIM_DIAG & 1)) {
        // The derivative maps leave through LDS: a thread's 6 adjacent outputs would be 4-byte stores 24 bytes apart (18 partial
        // cache lines per wave store; the three maps cost 12 of the kernel's 25 us that way), rows of 54 floats are 2-3 lines.
        __syncthreads();                       // every thread is done reading s_v
        if (tid < kTH * (kTW / kPH)) {
            const int r = tid / (kTW / kPH), c0 = (tid - r * (kTW / kPH)) * kPH;
#pragma unroll
            for (int o = 0; o < kPH; o++) { s_v[0][r][c0 + o] = res[0][o]; s_v[1][r][c0 + o] = res[1][o]; s_v[2][r][c0 + o] = res[2][o]; }
        }
        __syncthreads();
        typedef float __attribute__((address_space(1)))* GlobalW;
        const GlobalW d0 = (GlobalW)(dm_dmu1 + plane), d1 = (GlobalW)(dm_ds11 + plane), d2 = (GlobalW)(dm_ds12 + plane);
#pragma unroll
        for (int t = 0; t < (kTH * kTW + 255) / 256; t++) {
            const int i = tid + 256 * t, r = i / kTW, c = i - r * kTW;
            const int y = y0 + r, x = x0 + c;
            if (i < kTH * kTW && x < W && y < H) {
                const unsigned oo = (unsigned)y * (unsigned)W + (unsigned)x;
                d0[oo] = s_v[0][r][c]; d1[oo] = s_v[1][r][c]; d2[oo] = s_v[2][r][c];
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { val += __shfl_xor(val, d, 64); l1 += __shfl_xor(l1, d, 64); }
    if ((tid & 63) == 0) { s_red[tid >> 6] = val; s_red[4 + (tid >> 6)] = l1; }
    __syncthreads();
    if (tid == 0) {
        const float vs = s_red[0] + s_red[1] + s_red[2] + s_red[3], vl = s_red[4] + s_red[5] + s_red[6] + s_red[7];
        if (A.partial) {
            // one slot per workgroup, summed by loss_combine_kernel: thousands of atomics on ONE address serialise in a
            // single L2 channel (~13 ns each) and were most of this kernel's run time
            const int nb = gx * gy * gz;
            const int b = (bz * gy + by) * gx + bx;
            A.partial[b] = vs;
            A.partial[nb + b] = vl;
        } else {
            atomicAdd(A.ssim_sum, vs);
            if (A.l1_sum) atomicAdd(A.l1_sum, vl);
        }
    }
}

__global__ void __launch_bounds__(256) ssim_fwd_kernel(SsimFwdArgs A, Gauss g)
{
    __shared__ float lds[kSsimFwdLds];
    ssim_fwd_body(A, g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z, lds);
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(int H, int W, float inv_n, float l1_coef, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, Gauss g, const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_ds11, const float* __restrict__ dm_ds12,
                                                       const float* __restrict__ dL_dmean, float* __restrict__ dL_dimg1,
                                                       const float* const* __restrict__ img2_slot, CombineArgs comb)
{
    if (img2_slot) img2 = *img2_slot;
    __shared__ float s_v[3][kTH][kLds];
    const int tid = threadIdx.x, col = tid & 63, rg = tid >> 6;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    float w[11];
    gauss_to_vgprs(g, w);
    {
        const GlobalF p0 = (GlobalF)(dm_dmu1 + plane), p1 = (GlobalF)(dm_ds11 + plane), p2 = (GlobalF)(dm_ds12 + plane);
        const int x = x0 + col - kR;
        const bool xin = x >= 0 && x < W;
        const unsigned xc = (unsigned)min(max(x, 0), W - 1);
        float v0[kPV + 10], v1[kPV + 10], v2[kPV + 10];
#pragma unroll
        for (int j = 0; j < kPV + 10; j++) {
            const int y = y0 + rg * kPV + j - kR;
            const unsigned o = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + xc;
            const bool in = xin && y >= 0 && y < H;
            const float a = p0[o], b = p1[o], c = p2[o];
            v0[j] = in ? a : 0.f;
            v1[j] = in ? b : 0.f;
            v2[j] = in ? c : 0.f;
        }
#pragma unroll
        for (int o = 0; o < kPV; o++) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) { t0 += w[k] * v0[o + k]; t1 += w[k] * v1[o + k]; t2 += w[k] * v2[o + k]; }
            const int r = rg * kPV + o;
            s_v[0][r][col] = t0; s_v[1][r][col] = t1; s_v[2][r][col] = t2;
        }
    }
    __syncthreads();
    // the epilogue works on row-contiguous elements (coalesced image reads and gradient stores: see ssim_fwd_kernel); its image
    // reads go out before the LDS pass
    constexpr int kEp = (kTH * kTW + 255) / 256;
    const GlobalF q1 = (GlobalF)(img1 + plane), q2 = (GlobalF)(img2 + plane);
    float i1[kEp], i2[kEp];
#pragma unroll
    for (int t = 0; t < kEp; t++) {
        const int i = min(tid + 256 * t, kTH * kTW - 1), r = i / kTW, c = i - r * kTW;
        const unsigned oo = (unsigned)min(y0 + r, H - 1) * (unsigned)W + (unsigned)min(x0 + c, W - 1);
        i1[t] = q1[oo]; i2[t] = q2[oo];
    }
    float res[3][kPH];
    if (tid < kTH * (kTW / kPH)) {
        const int r = tid / (kTW / kPH), c0 = (tid - r * (kTW / kPH)) * kPH;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float v[kPH + 10];
#pragma unroll
            for (int j = 0; j < kPH + 10; j++) v[j] = s_v[q][r][c0 + j];
#pragma unroll
            for (int o = 0; o < kPH; o++) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) t += w[k] * v[o + k];
                res[q][o] = t;
            }
            asm volatile("" : "+v"(res[q][0]), "+v"(res[q][1]), "+v"(res[q][2]), "+v"(res[q][3]), "+v"(res[q][4]), "+v"(res[q][5]) :: "memory");
        }
    }
    __syncthreads();                           // every thread is done reading s_v
    if (tid < kTH * (kTW / kPH)) {
        const int r = tid / (kTW / kPH), c0 = (tid - r * (kTW / kPH)) * kPH;
#pragma unroll
        for (int o = 0; o < kPH; o++) { s_v[0][r][c0 + o] = res[0][o]; s_v[1][r][c0 + o] = res[1][o]; s_v[2][r][c0 + o] = res[2][o]; }
    }
    __syncthreads();
    const float gm = dL_dmean[0];
    typedef float __attribute__((address_space(1)))* GlobalW;
    const GlobalW dst = (GlobalW)(dL_dimg1 + plane);
#pragma unroll
    for (int t = 0; t < kEp; t++) {
        const int i = tid + 256 * t, r = i / kTW, c = i - r * kTW;
        const int y = y0 + r, x = x0 + c;
        if (i < kTH * kTW && x < W && y < H) {
            // the zero-padded symmetric window is its own adjoint
            // inv_n scales the SSIM-map adjoint, l1_coef the sign(img1 - img2) of an optional mean-|.| term
            const float df = i1[t] - i2[t];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            dst[(unsigned)y * (unsigned)W + (unsigned)x] = ((s_v[0][r][c] + 2.f * i1[t] * s_v[1][r][c] + i2[t] * s_v[2][r][c]) * inv_n + l1_coef * sg) * gm;
        }
    }
    // optional rider: the LAST workgroup of the grid also sums the forward kernels' partials into the loss value (the train step
    // launches this kernel after both of them; a one-workgroup kernel of its own cost 5-8 us of the replayed step)
    if (comb.out && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) {
        __shared__ float s_red[3][4];
        combine_partials(comb, s_red);
    }
}

// ---- KNN ------------------------------------------------------------------------------------------------------
constexpr int kKnnChunk = 1024;  // nodes staged per pass
constexpr int kKnnDpad = 16;

// Q = number of float4 per (zero-padded) node row: D <= 4*Q.  Compile-time so that the distance loop fully
// unrolls and the wave-uniform node reads become ds_read_b128 broadcasts.  Every thread owns kKnnPts query points:
// one set of LDS reads feeds kKnnPts independent distance chains (the loop is latency-, not throughput-bound).
constexpr int kKnnPts = 2;
constexpr int kKnnGrp = 4;

template <int K, int Q>
__global__ void __launch_bounds__(256) knn_kernel(int N, int M, int D, const float* __restrict__ x, const float* __restrict__ nodes,
                                                  long long* __restrict__ idx, float* __restrict__ dist2,
                                                  const float* __restrict__ x2, int D1, int stride2)
{
    __shared__ float4 s_nodes[kKnnChunk * Q];
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * kKnnPts;
    float xv[kKnnPts][4 * Q];
#pragma unroll
    for (int u = 0; u < kKnnPts; u++)
#pragma unroll
        for (int d = 0; d < 4 * Q; d++) {
            // coordinates [0, D1) come from x (row stride D1), [D1, D) from x2 (row stride stride2); x2 == nullptr: D1 = D
            float v = 0.f;
            if (p0 + u < N && d < D) v = d < D1 ? x[(size_t)(p0 + u) * D1 + d] : x2[(size_t)(p0 + u) * stride2 + d - D1];
            xv[u][d] = v;
        }
    float bd[kKnnPts][K];
    int bi[kKnnPts][K];
#pragma unroll
    for (int u = 0; u < kKnnPts; u++)
#pragma unroll
        for (int k = 0; k < K; k++) { bd[u][k] = INFINITY; bi[u][k] = 0; }
    for (int base = 0; base < M; base += kKnnChunk) {
        const int cnt = (M - base) < kKnnChunk ? (M - base) : kKnnChunk;
        __syncthreads();
        // one node row per thread and pass, every load of the pass issued before the first LDS store (a flat
        // element-wise copy is a chain of dependent global-load latencies and used to cost more than the scan)
        for (int r0 = 0; r0 < cnt; r0 += 1024) {
            float v[4][4 * Q];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * 256 + threadIdx.x;
#pragma unroll
                for (int d = 0; d < 4 * Q; d++) v[i][d] = (r < cnt && d < D) ? nodes[(size_t)(base + r) * D + d] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i * 256 + threadIdx.x;
                if (r < cnt)
#pragma unroll
                    for (int q = 0; q < Q; q++) s_nodes[r * Q + q] = make_float4(v[i][4 * q], v[i][4 * q + 1], v[i][4 * q + 2], v[i][4 * q + 3]);
            }
        }
        __syncthreads();
        // groups of kKnnGrp nodes: all LDS reads of a group are issued before the first use, the kKnnPts x kKnnGrp
        // distances are independent FMA chains, and the (rare) insertions come last
        for (int j0 = 0; j0 < cnt; j0 += kKnnGrp) {
            float4 nd[kKnnGrp][Q];
#pragma unroll
            for (int g = 0; g < kKnnGrp; g++)
#pragma unroll
                for (int q = 0; q < Q; q++) nd[g][q] = s_nodes[min(j0 + g, cnt - 1) * Q + q];  // wave-uniform: LDS broadcast
            // (A v_pk_add_f32 / v_pk_fma_f32 formulation pairing the two points was measured at the same 0.23 ms: packed
            // fp32 does not issue faster than two scalar ops on gfx950 and needs extra moves to splat the node value.)
            float acc[kKnnPts][kKnnGrp];
#pragma unroll
            for (int u = 0; u < kKnnPts; u++)
#pragma unroll
                for (int g = 0; g < kKnnGrp; g++) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        float t;
                        t = xv[u][4 * q + 0] - nd[g][q].x; a += t * t;
                        t = xv[u][4 * q + 1] - nd[g][q].y; a += t * t;
                        t = xv[u][4 * q + 2] - nd[g][q].z; a += t * t;
                        t = xv[u][4 * q + 3] - nd[g][q].w; a += t * t;
                    }
                    acc[u][g] = (j0 + g < cnt) ? a : INFINITY;
                }
#pragma unroll
            for (int u = 0; u < kKnnPts; u++) {
                float best = acc[u][0];
#pragma unroll
                for (int g = 1; g < kKnnGrp; g++) best = fminf(best, acc[u][g]);
                if (best < bd[u][K - 1]) {
#pragma unroll
                    for (int g = 0; g < kKnnGrp; g++) {
                        // insertion into the sorted K best; strict < keeps the lower index on ties
                        if (acc[u][g] < bd[u][K - 1]) {
                            bd[u][K - 1] = acc[u][g]; bi[u][K - 1] = base + j0 + g;
#pragma unroll
                            for (int k = K - 1; k > 0; k--) {
                                if (bd[u][k] < bd[u][k - 1]) {
                                    const float td = bd[u][k]; bd[u][k] = bd[u][k - 1]; bd[u][k - 1] = td;
                                    const int ti = bi[u][k]; bi[u][k] = bi[u][k - 1]; bi[u][k - 1] = ti;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kKnnPts; u++)
        if (p0 + u < N) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                idx[(size_t)(p0 + u) * K + k] = bi[u][k];
                if (dist2) dist2[(size_t)(p0 + u) * K + k] = bd[u][k];
            }
        }
}

template <int K, int Q>
int launch_knn_q(int N, int M, int D, const float* x, const float* nodes, long long* idx, float* dist2, hipStream_t s,
                 const float* x2, int D1, int stride2)
{
    const int per_block = 256 * kKnnPts;
    hipLaunchKernelGGL((knn_kernel<K, Q>), dim3((N + per_block - 1) / per_block), dim3(256), 0, s, N, M, D, x, nodes, idx, dist2,
                       x2, D1, stride2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("knn_kernel: ") + hipGetErrorString(e));
    return 0;
}

template <int K>
int launch_knn(int N, int M, int D, const float* x, const float* nodes, long long* idx, float* dist2, hipStream_t s,
               const float* x2 = nullptr, int D1 = -1, int stride2 = 0)
{
    if (!x2) D1 = D;
    switch ((D + 3) / 4) {
    case 1: return launch_knn_q<K, 1>(N, M, D, x, nodes, idx, dist2, s, x2, D1, stride2);
    case 2: return launch_knn_q<K, 2>(N, M, D, x, nodes, idx, dist2, s, x2, D1, stride2);
    case 3: return launch_knn_q<K, 3>(N, M, D, x, nodes, idx, dist2, s, x2, D1, stride2);
    default: return launch_knn_q<K, 4>(N, M, D, x, nodes, idx, dist2, s, x2, D1, stride2);
    }
}

// ---- KNN refinement -----------------------------------------------------------------------------------------------
// Exact K nearest neighbours again, but seeded with a previous answer (last step's indices: surfels and nodes move by
// ~1e-6 per step).  The K seed nodes give an upper bound T on the K-th smallest distance; the 3-D part of the distance
// (coordinates 0..2 of D) is a lower bound of the full one, so only nodes with d3 <= T can be in the answer.  The scan
// over all M nodes therefore needs 3 of the D coordinates (7 instead of 2*D VALU operations per node) and just records
// the few candidates; full distances are evaluated for those only.  Any seed (stale, random, duplicated) gives the exact
// result: a bad seed only makes T large, a full candidate list falls back to the plain scan for that point.
//
// The scan skips whole 32-node blocks: every block has a bounding box (built next to the staged nodes), every wave the box of
// its points' search spheres (centre x, radius sqrt(T)); a block whose box misses the wave's cannot hold a candidate of any
// lane, and the test is one lane per block + one ballot.  Pays when both sides are spatially coherent -- surfels stored in the
// order of their nearest node and nodes stored along a space-filling curve (Trainer.sort_surfels / sort_nodes: 32 blocks ->
// ~4 per wave at 200 k surfels / 1024 nodes); any order gives the same, exact result.
constexpr int kKnnCap = 12;
// 512 threads x 1 point: 200k points are 3125 waves (3 per SIMD) instead of the 1563 of 256 threads x 2 points, and the node
// table (48 KB) is shared by twice the points per workgroup, so two workgroups still fit a CU.
constexpr int kRefThreads = 512;
constexpr int kRefPts = 1;
constexpr int kRefGrp = 4;   // nodes per group of LDS broadcasts in flight (8: no change)

template <int K, int Q>
__global__ void __launch_bounds__(kRefThreads) knn_refine_kernel(int N, int M, int D, const float* __restrict__ x, const float* __restrict__ nodes,
                                                         long long* __restrict__ idx, const float* __restrict__ x2, int D1, int stride2)
{
    extern __shared__ float4 s_dyn[];
    const int Mp = (M + 31) & ~31;                                        // rows padded to the 32-node scan blocks (zeros, masked)
    const int nblk = Mp >> 5;
    float4* s_nodes = s_dyn;                                              // [Mp][Q]
    float4* s_box = s_dyn + (size_t)Mp * Q;                               // [nblk][2]: min, max of the block's nodes (coordinates 0..2)
    int* s_list = reinterpret_cast<int*>(s_box + 2 * nblk);               // [kRefThreads * kRefPts][kKnnCap]
    // node table -> LDS rows of 4 Q floats (zero padded).  The table is read as a flat stream of 16-byte vectors (a 4-byte load
    // occupies the address unit as long as a 16-byte one: 24 loads per thread became 6) and scattered into the padded rows
    {
        float* s_f = reinterpret_cast<float*>(s_nodes);
        const int F = M * D, nvec = (reinterpret_cast<size_t>(nodes) & 15) == 0 ? F >> 2 : 0;
        for (int r = threadIdx.x; r < Mp; r += kRefThreads)
            for (int c = (r < M ? D : 0); c < 4 * Q; c++) s_f[r * 4 * Q + c] = 0.f;
        for (int base = 0; base < nvec; base += 8 * kRefThreads) {
            float4 q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int v = base + i * kRefThreads + threadIdx.x;
                q[i] = v < nvec ? reinterpret_cast<const float4*>(nodes)[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int v = base + i * kRefThreads + threadIdx.x;
                if (v >= nvec) continue;
                int r = (4 * v) / D, c = 4 * v - r * D;
                const float e[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s_f[r * 4 * Q + c] = e[k];
                    if (++c == D) { c = 0; r++; }
                }
            }
        }
        for (int e = 4 * nvec + threadIdx.x; e < F; e += kRefThreads) {   // the last F % 4 elements, or all of an unaligned table
            const int r = e / D, c = e - r * D;
            s_f[r * 4 * Q + c] = nodes[e];
        }
    }
    __syncthreads();
    // bounding boxes of the 32-node blocks: 16 lanes per block, two nodes each, min / max over the row of 16 with DPP shifts
    // (32 threads walking 32 nodes each left the other 480 waiting at the barrier)
    for (int b = threadIdx.x >> 4; b < nblk; b += kRefThreads >> 4) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = b * 32 + 2 * (threadIdx.x & 15) + h;
            if (j < M) {
                const float4 nd = s_nodes[j * Q];
                lo[0] = fminf(lo[0], nd.x); lo[1] = fminf(lo[1], nd.y); lo[2] = fminf(lo[2], nd.z);
                hi[0] = fmaxf(hi[0], nd.x); hi[1] = fmaxf(hi[1], nd.y); hi[2] = fmaxf(hi[2], nd.z);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
#define KNN_ROW_STEP(n)                                                                                                                         \
            lo[c] = fminf(lo[c], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lo[c]), __builtin_bit_cast(int, lo[c]), \
                                                                                       0x110 + (n), 0xf, 0xf, false)));                         \
            hi[c] = fmaxf(hi[c], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, hi[c]), __builtin_bit_cast(int, hi[c]), \
                                                                                       0x110 + (n), 0xf, 0xf, false)))
            KNN_ROW_STEP(1); KNN_ROW_STEP(2); KNN_ROW_STEP(4); KNN_ROW_STEP(8);
#undef KNN_ROW_STEP
        }
        if ((threadIdx.x & 15) == 15) {
            s_box[2 * b] = make_float4(lo[0], lo[1], lo[2], 0.f);
            s_box[2 * b + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
    __syncthreads();
    const int p0 = (blockIdx.x * kRefThreads + threadIdx.x) * kRefPts;
    float xv[kRefPts][4 * Q];
    float T[kRefPts];
    int cnt[kRefPts];
    auto full_dist = [&](int u, int j) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const float4 nd = s_nodes[j * Q + q];
            float t;
            t = xv[u][4 * q + 0] - nd.x; a += t * t;
            t = xv[u][4 * q + 1] - nd.y; a += t * t;
            t = xv[u][4 * q + 2] - nd.z; a += t * t;
            t = xv[u][4 * q + 3] - nd.w; a += t * t;
        }
        return a;
    };
#pragma unroll
    for (int u = 0; u < kRefPts; u++) {
        const bool in = p0 + u < N;
#pragma unroll
        for (int d = 0; d < 4 * Q; d++) {
            float v = 0.f;
            if (in && d < D) v = d < D1 ? x[(size_t)(p0 + u) * D1 + d] : x2[(size_t)(p0 + u) * stride2 + d - D1];
            xv[u][d] = v;
        }
        // bound from the seed (K distinct valid nodes), slightly inflated never hurts: it is only a filter
        int sj[K];
        bool ok = in;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const long long j = in ? idx[(size_t)(p0 + u) * K + k] : 0;
            ok = ok && j >= 0 && j < M;
            sj[k] = (int)(j < 0 ? 0 : (j >= M ? M - 1 : j));
        }
#pragma unroll
        for (int k = 1; k < K; k++)
#pragma unroll
            for (int k2 = 0; k2 < k; k2++) ok = ok && sj[k] != sj[k2];
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) t = fmaxf(t, full_dist(u, sj[k]));
        T[u] = in ? (ok ? t : INFINITY) : -1.f;
        cnt[u] = 0;
    }
    // ---- scan: 3-D lower bound only.  Branch-free inner loop: the sign of d3 - T' is shifted into a 32-node hit word
    // (v_alignbit), 7 VALU operations per (node, point); the words are drained once per 32 nodes.
    float Tn[kRefPts];
#pragma unroll
    for (int u = 0; u < kRefPts; u++) Tn[u] = -(T[u] * (1.0f + 1e-6f) + 1e-30f);   // inflated: d3 == T must stay a hit
    // box of the wave's search spheres (lanes past N have T = -1: no sphere).  |x_c - n_c| <= sqrt(d3) <= sqrt(-Tn) on every axis
    // for a hit; the radius is rounded up generously, the box only filters
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < kRefPts; u++)
        if (T[u] >= 0.f) {
            const float r = sqrtf(-Tn[u]) * 1.0001f + 1e-30f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                blo[c] = fminf(blo[c], xv[u][c] - r);
                bhi[c] = fmaxf(bhi[c], xv[u][c] + r);
            }
        }
    // FOUR boxes per wave, one per 16 lanes: where consecutive points change their nearest node across a jump of the node order,
    // one box over all 64 lanes spans the jump and touches most of the blocks (mean 8 of 32 but up to 23: those waves set the
    // kernel's time); the union of four tight boxes does not
    // min / max over each row of 16 lanes with DPP row shifts (lane 15 of a row ends up with the row's result; a lane without a
    // source keeps its own value), then the four results to scalar registers: no LDS traffic (48 ds_bpermute before)
    const int lane = threadIdx.x & 63;
    float qlo[4][3], qhi[4][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float lo = blo[c], hi = bhi[c];
#define KNN_ROW_STEP(n)                                                                                                                   \
        lo = fminf(lo, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lo), __builtin_bit_cast(int, lo),   \
                                                                             0x110 + (n), 0xf, 0xf, false)));                             \
        hi = fmaxf(hi, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, hi), __builtin_bit_cast(int, hi),   \
                                                                             0x110 + (n), 0xf, 0xf, false)))
        KNN_ROW_STEP(1); KNN_ROW_STEP(2); KNN_ROW_STEP(4); KNN_ROW_STEP(8);
#undef KNN_ROW_STEP
#pragma unroll
        for (int q = 0; q < 4; q++) {
            qlo[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 16 * q + 15));
            qhi[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 16 * q + 15));
        }
    }
    for (int bb = 0; bb < nblk; bb += 64) {
    bool touch = false;
    if (bb + lane < nblk) {
        const float4 lo = s_box[2 * (bb + lane)], hi = s_box[2 * (bb + lane) + 1];
#pragma unroll
        for (int q = 0; q < 4; q++)
            touch = touch || (lo.x <= qhi[q][0] && hi.x >= qlo[q][0] && lo.y <= qhi[q][1] && hi.y >= qlo[q][1] && lo.z <= qhi[q][2] &&
                              hi.z >= qlo[q][2]);
    }
    unsigned long long blocks = __ballot(touch);
    while (blocks) {
        const int j0 = (bb + __builtin_ctzll(blocks)) * 32;
        blocks &= blocks - 1;
        // ONE LDS read per block: lane g fetches node j0 + g (a wave-wide broadcast read of a node occupies the LDS pipe like any
        // other 1-KB read, and 32 of them per block were what bounded the scan: 28 us whatever the instruction count).  Lane g also
        // tests its node against the wave's boxes once for all lanes; the survivors (a third of a touched block) go to scalar
        // registers (v_readlane) and are tested per point
        float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
        bool inb = false;
        if (lane < 32 && j0 + lane < M) {
            mine = s_nodes[(j0 + lane) * Q];
#pragma unroll
            for (int q = 0; q < 4; q++)
                inb = inb || (mine.x >= qlo[q][0] && mine.x <= qhi[q][0] && mine.y >= qlo[q][1] && mine.y <= qhi[q][1] &&
                              mine.z >= qlo[q][2] && mine.z <= qhi[q][2]);
        }
        const unsigned sv = (unsigned)__ballot(inb);   // bit g <-> node j0 + g
        if (sv == 0u) continue;
        for (unsigned m = sv; m; m &= m - 1) {
            const int g = __builtin_ctz(m);
            const float nx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), g));
            const float ny = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), g));
            const float nz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.z), g));
#pragma unroll
            for (int u = 0; u < kRefPts; u++) {
                float t, a;
                t = xv[u][0] - nx; a = fmaf(t, t, Tn[u]);
                t = xv[u][1] - ny; a = fmaf(t, t, a);
                t = xv[u][2] - nz; a = fmaf(t, t, a);
                if (a < 0.f) {   // a candidate of this point (3 per point on average: the branch is skipped by most waves)
                    if (cnt[u] < kKnnCap) s_list[(threadIdx.x * kRefPts + u) * kKnnCap + cnt[u]] = j0 + g;
                    cnt[u]++;
                }
            }
        }
    }
    }
    // ---- candidates (ascending index, strict < on insertion: ties keep the lower index like the plain scan)
#pragma unroll
    for (int u = 0; u < kRefPts; u++) {
        if (p0 + u >= N) continue;
        float bd[K];
        int bi[K];
#pragma unroll
        for (int k = 0; k < K; k++) { bd[k] = INFINITY; bi[k] = 0; }
        const bool listed = cnt[u] <= kKnnCap;
        const int n = listed ? cnt[u] : M;
        for (int c = 0; c < n; c++) {
            const int j = listed ? s_list[(threadIdx.x * kRefPts + u) * kKnnCap + c] : c;
            const float dj = full_dist(u, j);
            if (dj < bd[K - 1]) {
                bd[K - 1] = dj; bi[K - 1] = j;
#pragma unroll
                for (int k = K - 1; k > 0; k--) {
                    if (bd[k] < bd[k - 1]) {
                        const float td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                        const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) idx[(size_t)(p0 + u) * K + k] = bi[k];
    }
}

template <int K, int Q>
int launch_knn_refine_q(int N, int M, int D, const float* x, const float* nodes, long long* idx, hipStream_t s, const float* x2, int D1,
                        int stride2)
{
    const int per_block = kRefThreads * kRefPts;
    const size_t lds = (size_t)((M + 31) & ~31) * Q * sizeof(float4) + (size_t)((M + 31) >> 5) * 2 * sizeof(float4) +
                       (size_t)kRefThreads * kRefPts * kKnnCap * sizeof(int);
    hipLaunchKernelGGL((knn_refine_kernel<K, Q>), dim3((N + per_block - 1) / per_block), dim3(kRefThreads), lds, s, N, M, D, x, nodes, idx, x2, D1,
                       stride2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("knn_refine_kernel: ") + hipGetErrorString(e));
    return 0;
}

// ---- seeded refine on the matrix cores ---------------------------------------------------------------------------------
// The 3-D block / box culling of knn_refine_kernel needs the K-th seed distance to be a SPATIAL radius.  In a trained scene it is
// not: the 8 hyper coordinates of surfels and nodes drift apart, the 11-D distance of the third neighbour exceeds the node spacing
// several times, every block is touched, candidate lists overflow, and the kernel is slower than the plain scan (172 us at 125 k
// surfels x 512 nodes against 36 us on the untrained scene).  This kernel filters with the FULL distance instead, dense and
// data-independent:   score[j][i] = |n_j|^2 - 2 x_i . n_j  (= d^2 - |x_i|^2)  for 32 nodes x 32 points per matrix instruction.
//   * f32-input MFMA runs at the vector rate on gfx950 (64 cycles per 32x32x2), bf16 MFMA sixteen times faster, so both operands
//     are split  v = hi + lo  (two bf16, 16 mantissa bits) and  x.n ~ xh.nh + xh.nl + xl.nh : three v_mfma_f32_32x32x16_bf16 per
//     tile (K = 16 slots: 11 coordinates, |n|^2 as hi + lo against 1, 3 spare) instead of six f32 ones at four times the cycles
//     each.  What is dropped (xl.nl and the rounding of the lo parts) is below 1.2e-5 (|x|^2 + |n|^2); products are exact in the
//     f32 accumulator.  This is only the FILTER: a node is a candidate of point i when  score <= T_i - |x_i|^2 + eps,  T_i the
//     exact distance of the K-th seed neighbour, eps = 1e-4 (|x|^2 + max |n|^2), so no node within the seed's bound is missed.
//   * the accumulator starts at -(threshold), a hit is a SIGN BIT, and the 16 results of a tile are shifted into a per-lane hit
//     word with one v_alignbit_b32 each: no compare, no branch, no list in LDS.
//   * the K plus few candidates are evaluated exactly (f32 differences, the same operation order as the plain scan) and ranked
//     (distance, then index).  Two lanes (l, l + 32) share a point and own alternating groups of 4 rows of every 32-node tile (the
//     D layout of the instruction); their top-K lists are merged at the end.  A garbage seed (T = inf) sets every bit: that lane
//     scans its rows itself.
#ifndef DGS_KNN_DIAG
#define DGS_KNN_DIAG 0   // development only: 1 no candidate evaluation, 2 no MFMA loop, 8 report the candidate count
#endif
constexpr int kRmThreads = 512;   // 8 waves x 32 points
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float v)          // round to nearest even; inputs are finite
{
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// v -> (hi, lo) bf16 bit patterns with hi + lo ~ v to 16 mantissa bits
__device__ __forceinline__ void bf16_split(float v, unsigned& hi, unsigned& lo)
{
    hi = bf16_rne(v);
    lo = bf16_rne(v - __uint_as_float(hi << 16));
}

// TP: pairs of 32-node tiles (one 32-bit hit word per pair and lane), Mp <= 64 TP
template <int K, int TP>
__global__ void __launch_bounds__(kRmThreads) knn_refine_mfma_kernel(int N, int M, int D, const float* __restrict__ x, const float* __restrict__ nodes,
                                                                     long long* __restrict__ idx, const float* __restrict__ x2, int D1, int stride2)
{
    extern __shared__ float s_n[];                                      // [Mp][12] f32: n_0 .. n_10 (zero padded), |n|^2
    const int Mp = (M + 31) & ~31, ntiles = Mp >> 5;
    uint4* s_hi = reinterpret_cast<uint4*>(s_n + (size_t)Mp * 12);      // [Mp][2] x 8 bf16: hi parts of n_0 .. n_10, |n|^2 hi, |n|^2 lo, 0 0 0
    uint4* s_lo = s_hi + (size_t)Mp * 2;                                // [Mp][2] x 8 bf16: lo parts of n_0 .. n_10, 0 ...
    __shared__ float s_max[kRmThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    float n2max = 0.f;
    for (int j = tid; j < Mp; j += kRmThreads) {                         // one node per thread: its row's loads are all in flight at once
        float v[11];
#pragma unroll
        for (int c = 0; c < 11; c++) v[c] = (j < M && c < D) ? nodes[(size_t)j * D + c] : 0.f;
        float n2 = 0.f;
#pragma unroll
        for (int c = 0; c < 11; c++) n2 += v[c] * v[c];
        if (j < M) n2max = fmaxf(n2max, n2); else n2 = 3.0e38f;          // padded rows never qualify
        float4* row = reinterpret_cast<float4*>(s_n + (size_t)j * 12);
        row[0] = make_float4(v[0], v[1], v[2], v[3]); row[1] = make_float4(v[4], v[5], v[6], v[7]); row[2] = make_float4(v[8], v[9], v[10], n2);
        unsigned h[13], l[13];
#pragma unroll
        for (int c = 0; c < 11; c++) bf16_split(v[c], h[c], l[c]);
        bf16_split(n2, h[11], h[12]);
        s_hi[2 * j] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        s_hi[2 * j + 1] = make_uint4(h[8] | (h[9] << 16), h[10] | (h[11] << 16), h[12], 0u);
        s_lo[2 * j] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        s_lo[2 * j + 1] = make_uint4(l[8] | (l[9] << 16), l[10], 0u, 0u);
    }
    for (int d = 32; d >= 1; d >>= 1) n2max = fmaxf(n2max, __shfl_xor(n2max, d, 64));
    if (lane == 0) s_max[tid >> 6] = n2max;
    __syncthreads();
    n2max = s_max[0];
#pragma unroll
    for (int w = 1; w < kRmThreads / 64; w++) n2max = fmaxf(n2max, s_max[w]);

    // a wave takes groups of 32 points; the host sizes the grid so that every wave gets the same number of groups and all
    // workgroups are resident at once (782 workgroups on 768 slots ran as two rounds: twice the time)
    const int ngroups = (N + 31) >> 5, nwaves = gridDim.x * (kRmThreads / 64);
    for (int grp = blockIdx.x * (kRmThreads / 64) + (tid >> 6); grp < ngroups; grp += nwaves) {
    const int p = grp * 32 + (lane & 31);
    const bool in = p < N;
    float xv[11];
#pragma unroll
    for (int d = 0; d < 11; d++) {
        float v = 0.f;
        if (in && d < D) v = d < D1 ? x[(size_t)p * D1 + d] : x2[(size_t)p * stride2 + d - D1];
        xv[d] = v;
    }
    int sj[K];
    bool ok = in;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long long j = in ? idx[(size_t)p * K + k] : 0;
        ok = ok && j >= 0 && j < M;
        sj[k] = (int)(j < 0 ? 0 : (j >= M ? M - 1 : j));
    }
#pragma unroll
    for (int k = 1; k < K; k++)
#pragma unroll
        for (int k2 = 0; k2 < k; k2++) ok = ok && sj[k] != sj[k2];
    auto full_dist = [&](int j) {
        const float4 n0 = *reinterpret_cast<const float4*>(s_n + j * 12), n1 = *reinterpret_cast<const float4*>(s_n + j * 12 + 4),
                     n2 = *reinterpret_cast<const float4*>(s_n + j * 12 + 8);
        float a = 0.f, t;
        // same order of operations as knn_kernel / knn_refine_kernel (groups of four coordinates): identical distances, identical ties
        t = xv[0] - n0.x; a += t * t; t = xv[1] - n0.y; a += t * t; t = xv[2] - n0.z; a += t * t; t = xv[3] - n0.w; a += t * t;
        t = xv[4] - n1.x; a += t * t; t = xv[5] - n1.y; a += t * t; t = xv[6] - n1.z; a += t * t; t = xv[7] - n1.w; a += t * t;
        t = xv[8] - n2.x; a += t * t; t = xv[9] - n2.y; a += t * t; t = xv[10] - n2.z; a += t * t;
        return a;
    };
    // bound from the seed (K distinct valid nodes): exact distance of its farthest member
    float T = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) T = fmaxf(T, full_dist(sj[k]));
    if (!ok) T = INFINITY;
    float xx = 0.f;
#pragma unroll
    for (int d = 0; d < 11; d++) xx += xv[d] * xv[d];
    // hit <=> score - thr < 0.  The subtraction rides in the accumulator: the first MFMA of a tile starts from C = -thr
    const float nthr = in ? -(T * (1.0f + 1e-6f) - xx + (1e-4f * (xx + n2max) + 1e-30f)) : INFINITY;
    f32x16 cthr;
#pragma unroll
    for (int v = 0; v < 16; v++) cthr[v] = nthr;
    // B operands of this lane's half of the K slots: slots 0..7 = coordinates 0..7 | slots 8..15 = coordinates 8..10, 1, 1, 0, 0, 0
    bf16x8 bh, bl;
    {
        unsigned h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = half ? (e < 3 ? -2.0f * xv[8 + (e < 3 ? e : 0)] : 0.f) : -2.0f * xv[e];      // (no dynamic register index)
            bf16_split(v, h[e], l[e]);
        }
        if (half) { h[3] = 0x3f80u; h[4] = 0x3f80u; }                     // 1.0 against |n|^2 hi and lo
        bh = __builtin_bit_cast(bf16x8, make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)));
        bl = __builtin_bit_cast(bf16x8, make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)));
    }
    unsigned hits[TP];
    const uint4* ahi = s_hi + 2 * (lane & 31) + half;
    const uint4* alo = s_lo + 2 * (lane & 31) + half;
#pragma unroll
    for (int w = 0; w < TP; w++) {
        unsigned word = 0u;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int t = 2 * w + u;
            if (t < ntiles && !(DGS_KNN_DIAG & 2)) {                     // wave-uniform
                const bf16x8 nh = __builtin_bit_cast(bf16x8, ahi[t * 64]), nl = __builtin_bit_cast(bf16x8, alo[t * 64]);
                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nh, bh, cthr, 0, 0, 0);     // nh.xh + |n|^2 - thr
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nl, bh, acc, 0, 0, 0);             // nl.xh   (|n|^2 slots of nl are 0)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nh, bl, acc, 0, 0, 0);             // nh.xl   (those slots of xl are 0)
                // the sign bits of the 16 results are shifted into the hit word, first result ends highest (v_alignbit_b32)
#pragma unroll
                for (int v = 0; v < 16; v++) word = __builtin_amdgcn_alignbit(word, __float_as_uint(acc[v]), 31);
            } else {
                word <<= 16;
            }
        }
        hits[w] = word;
    }
    // ---- exact evaluation of the candidates (ties keep the lower index like the plain scan)
    // D[i][j] of the instruction: lane = j + 32 ((i / 4) % 2), register v = 4 (i / 8) + i % 4  =>  row i = 8 (v / 4) + 4 half + v % 4;
    // bit 31 - (16 u + v) of word w is row i of tile 2 w + u
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = INFINITY; bi[k] = 0x7fffffff; }
    auto offer = [&](float dj, int j) {
        if (dj < bd[K - 1] || (dj == bd[K - 1] && j < bi[K - 1])) {
            bd[K - 1] = dj; bi[K - 1] = j;
#pragma unroll
            for (int k = K - 1; k > 0; k--) {
                if (bd[k] < bd[k - 1] || (bd[k] == bd[k - 1] && bi[k] < bi[k - 1])) {
                    const float td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                    const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
                }
            }
        }
    };
#if DGS_KNN_DIAG & 1
#pragma unroll
    for (int w = 0; w < TP; w++) bi[w % K] ^= (int)hits[w];      // (keeps the hit words alive)
#else
    // every trip each lane takes ITS next candidate, whichever word it is in: the number of trips is the largest candidate count of a
    // lane (5-6), not the sum over the words of the largest count per word (14 with 1.6 candidates per lane spread over 8 words)
    for (;;) {
        unsigned w = 0u;
        int wi = 0;
#pragma unroll
        for (int k = TP - 1; k >= 0; k--) { const bool nz = hits[k] != 0u; w = nz ? hits[k] : w; wi = nz ? k : wi; }
        if (__ballot(w != 0u) == 0ull) break;
        if (w != 0u) {
            const int q = __builtin_clz(w);                  // 16 u + v
            const unsigned bit = 0x80000000u >> q;
#pragma unroll
            for (int k = 0; k < TP; k++) hits[k] &= k == wi ? ~bit : ~0u;
            const int v = q & 15;
            const int j = ((2 * wi + (q >> 4)) << 5) + 8 * (v >> 2) + 4 * half + (v & 3);
            if (j < M) offer(full_dist(j), j);
        }
    }
#endif
    // merge with the partner lane's list
    float od[K];
    int oi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { od[k] = __shfl_xor(bd[k], 32, 64); oi[k] = __shfl_xor(bi[k], 32, 64); }
    if (in && half == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) offer(od[k], oi[k]);
#pragma unroll
        for (int k = 0; k < K; k++) idx[(size_t)p * K + k] = bi[k];
    }
    }   // groups
}

template <int K, int TP>
int launch_knn_refine_mfma_tp(int N, int M, int D, const float* x, const float* nodes, long long* idx, hipStream_t s, const float* x2, int D1, int stride2)
{
    const int Mp = (M + 31) & ~31;
    const size_t lds = (size_t)Mp * (12 * sizeof(float) + 4 * sizeof(uint4));
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                                return n > 0 ? n : 256; }();
    const int per_cu = lds > 80 * 1024 ? 1 : 2;                                         // resident workgroups per CU (LDS: 112 B per node; VGPRs: 2)
    const int wpb = kRmThreads / 64, ngroups = (N + 31) / 32, slots = cus * per_cu * wpb;
    const int iters = (ngroups + slots - 1) / slots, waves = (ngroups + iters - 1) / iters;
    hipLaunchKernelGGL((knn_refine_mfma_kernel<K, TP>), dim3((waves + wpb - 1) / wpb), dim3(kRmThreads), lds, s, N, M, D, x, nodes, idx, x2, D1,
                       stride2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("knn_refine_mfma_kernel: ") + hipGetErrorString(e));
    return 0;
}

template <int K>
int launch_knn_refine_mfma(int N, int M, int D, const float* x, const float* nodes, long long* idx, hipStream_t s, const float* x2, int D1, int stride2)
{
    if (M <= 256) return launch_knn_refine_mfma_tp<K, 4>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    if (M <= 512) return launch_knn_refine_mfma_tp<K, 8>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    return launch_knn_refine_mfma_tp<K, 16>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
}

template <int K>
int launch_knn_refine(int N, int M, int D, const float* x, const float* nodes, long long* idx, hipStream_t s, const float* x2, int D1,
                      int stride2, bool mfma)
{
    if (!x2) D1 = D;
    if (mfma && D <= 11 && M <= 1024) return launch_knn_refine_mfma<K>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    switch ((D + 3) / 4) {
    case 1: return launch_knn_refine_q<K, 1>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    case 2: return launch_knn_refine_q<K, 2>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    case 3: return launch_knn_refine_q<K, 3>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    default: return launch_knn_refine_q<K, 4>(N, M, D, x, nodes, idx, s, x2, D1, stride2);
    }
}

// ---- control-node LBS -------------------------------------------------------------------------------------------
constexpr int kLbsK = 3;
constexpr int kLbsHmax = 13;
constexpr int kLbsAttr = 13;       // quaternion 4 | trans 3 | rot 4 | scale 2
constexpr int kLbsBlocks = 256;    // backward: one partial gradient table per workgroup

struct LbsArgs {
    int N, M, H, fstride;
    const float* x; const float* feature; const long long* idx; const float* ntab; const float* attrs; const float* mask;
    // node table rows are tstride floats apart.  rad_raw / w_raw non-null: the table holds only [xyz | hyper] and the
    // kernel radius / weight are exp(rad_raw[j]) / sigmoid(w_raw[j]) (ControlNodeWarp.node_radius / node_weight)
    int tstride; const float* rad_raw; const float* w_raw;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// activations of the surfel parameters that render() applies around the deformation
// (gaussian_renderer/__init__.py:60-75: means3D = xyz + d_xyz, scales = exp(_scaling) + d_scaling,
//  rotations = normalize(_rotation + d_rotation), opacity = sigmoid(_opacity); scene/gaussian_model.py:60-78)
struct AsmArgs {
    const float* scaling_raw; const float* rotation_raw; const float* opacity_raw;
    float* means3D; float* scales; float* rotations; float* opacity;                                   // forward out
    const float* g_means3D; const float* g_scales; const float* g_rotations; const float* g_opacity;   // backward in
    float* g_xyz; float* g_scaling_raw; float* g_rotation_raw; float* g_opacity_raw;                   // backward out
};

// quaternion (r,i,j,k), not necessarily unit -> rotation matrix, utils/time_utils.py:115-132
__device__ __forceinline__ void quat_to_mat(const float* q, float* R, float& two_s)
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

// Table rows are gathered per lane (every lane another row): 4-byte aligned 16-byte loads fetch a row in 3-4 instructions
// instead of 11-13 scalar ones; the texture path spends its time per instruction and per cache line touched, not per byte.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load_row(const float* __restrict__ src, int n, float* __restrict__ out /*[16]*/)
{
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        if (c + 4 <= n) {
            const f4u v = *reinterpret_cast<const f4u*>(src + c);
            out[c] = v.x; out[c + 1] = v.y; out[c + 2] = v.z; out[c + 3] = v.w;
        } else {
#pragma unroll
            for (int d = c; d < c + 4; d++) out[d] = d < n ? src[d] : 0.f;
        }
    }
}

// per-point evaluation shared by forward and backward
struct LbsPoint {
    float w[kLbsK], e[kLbsK], dist[kLbsK], Ax[kLbsK][3], rad[kLbsK], wg[kLbsK];
    float W;
    int j[kLbsK];
};

template <int HM = kLbsHmax>
__device__ __forceinline__ void lbs_eval(const LbsArgs& a, int n, LbsPoint& p, float* xq /*[3+HM]*/)
{
    const int T = a.tstride;
    xq[0] = a.x[3 * n]; xq[1] = a.x[3 * n + 1]; xq[2] = a.x[3 * n + 2];
    {
        float fr[16];
        load_row(a.feature + (size_t)n * a.fstride, a.H < 16 ? a.H : 16, fr);
        for (int h = 0; h < HM; h++) xq[3 + h] = fr[h];
    }
    p.W = 0.f;
#pragma unroll
    for (int k = 0; k < kLbsK; k++) {
        const int j = (int)a.idx[(size_t)n * kLbsK + k];
        p.j[k] = j;
        float nd[16], at[16];
        load_row(a.ntab + (size_t)j * T, T < 16 ? T : 16, nd);
        load_row(a.attrs + (size_t)j * kLbsAttr, kLbsAttr, at);
        float dist = 0.f;
        for (int c = 0; c < 3 + HM; c++)
            if (c < 3 + a.H) { const float t = xq[c] - nd[c]; dist += t * t; }
        const float r = a.rad_raw ? expf(a.rad_raw[j]) : a.ntab[(size_t)j * T + 3 + a.H];
        const float wg = a.w_raw ? sigmoidf_(a.w_raw[j]) : a.ntab[(size_t)j * T + 3 + a.H + 1];
        p.rad[k] = r; p.wg[k] = wg;
        p.dist[k] = dist;
        p.e[k] = expf(-dist / (2.f * r * r));
        p.w[k] = p.e[k] * wg + 1e-7f;
        p.W += p.w[k];
        float R[9], two_s;
        quat_to_mat(at, R, two_s);
        const float dx = xq[0] - nd[0], dy = xq[1] - nd[1], dz = xq[2] - nd[2];
        p.Ax[k][0] = R[0] * dx + R[1] * dy + R[2] * dz + nd[0] + at[4];
        p.Ax[k][1] = R[3] * dx + R[4] * dy + R[5] * dz + nd[1] + at[5];
        p.Ax[k][2] = R[6] * dx + R[7] * dy + R[8] * dz + nd[2] + at[6];
    }
}

template <bool ASM>
__global__ void __launch_bounds__(256) lbs_fwd_kernel(LbsArgs a, float* d_xyz, float* d_rot, float* d_scale, AsmArgs s_)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    LbsPoint p;
    float xq[3 + kLbsHmax];
    lbs_eval(a, n, p, xq);
    const float inv = 1.0f / p.W, m = a.mask ? a.mask[n] : 1.0f;
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 0}, s[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < kLbsK; k++) {
        const float w = p.w[k] * inv;
        float at[16];
        load_row(a.attrs + (size_t)p.j[k] * kLbsAttr, kLbsAttr, at);
        for (int c = 0; c < 3; c++) t[c] += w * p.Ax[k][c];
        for (int c = 0; c < 4; c++) q[c] += w * at[7 + c];
        for (int c = 0; c < 2; c++) s[c] += w * at[11 + c];
    }
    if (!ASM) {
        for (int c = 0; c < 3; c++) d_xyz[3 * n + c] = (t[c] - xq[c]) * m;
        for (int c = 0; c < 4; c++) d_rot[4 * n + c] = q[c] * m;
        for (int c = 0; c < 2; c++) d_scale[2 * n + c] = s[c] * m;
    } else {
        for (int c = 0; c < 3; c++) s_.means3D[3 * n + c] = xq[c] + (t[c] - xq[c]) * m;
        for (int c = 0; c < 2; c++) s_.scales[2 * n + c] = expf(s_.scaling_raw[2 * n + c]) + s[c] * m;
        float v[4], n2 = 0.f;
        for (int c = 0; c < 4; c++) { v[c] = s_.rotation_raw[4 * n + c] + q[c] * m; n2 += v[c] * v[c]; }
        const float invn = 1.0f / fmaxf(sqrtf(n2), 1e-12f);   // F.normalize
        for (int c = 0; c < 4; c++) s_.rotations[4 * n + c] = v[c] * invn;
        s_.opacity[n] = sigmoidf_(s_.opacity_raw[n]);
    }
}

// Backward.  Per-node gradients (13 attribute + H+2 table columns) of the ~782 points of a workgroup are accumulated
// in an LDS table with ds_add_f32 and written once as that workgroup's partial table; lbs_reduce_kernel sums the
// kLbsBlocks partials.  (Direct global atomics would be ~7 M adds onto ~24 k hot addresses.)
constexpr int kLbsBwdThreads = 512;    // one workgroup per CU (the LDS table is ~94 KB); 8 waves keep 256 VGPRs per lane (no spills)

// LDS float atomics are the wrong tool here: ds_add_f32 retires ~1 lane per 2.4 clocks on gfx950 (measured: 13.8 M lane
// adds = 55 of this kernel's 104 us; the same pattern with ds_add_u32 takes 8 us, tools/micro/lds_atomic_bench.hip; the
// bucketed delivery below costs ~40 us, LDS-instruction bound, and makes the sums deterministic).
// The per-node sums are therefore built without float atomics: for each of the K neighbour slots the workgroup's threads
// park their G contributions in an LDS exchange buffer and take a slot in the target node's small bucket with an integer
// atomic (fast); after one barrier the thread that OWNS a node adds the parked rows into the node's table row with plain
// read-modify-writes.  Buckets hold kLbsSlots rows; the rare excess is added with float atomics in a second, guarded phase.
constexpr int kLbsSlots = 4;   // parked rows per node and delivery round; the rare excess goes through float atomics afterwards

__device__ __forceinline__ void lbs_deliver(bool valid, int j, const float* cv, int G, int GS, int M, float* s_tab, float* s_exch,
                                            int* s_cnt, unsigned short* s_slot, int* s_over)
{
    // on entry: s_cnt[] == 0, *s_over == 0 (left that way by the previous round)
    const int tid = threadIdx.x;
    int pos = 0;
    if (valid) {
        pos = atomicAdd(&s_cnt[j], 1);                       // integer LDS atomic: fast
        if (pos < kLbsSlots) {
            s_slot[j * kLbsSlots + pos] = (unsigned short)tid;
#pragma unroll
            for (int c = 0; c < kLbsAttr + kLbsHmax + 2; c++)
                if (c < G) s_exch[tid * GS + c] = cv[c];
        } else {
            *s_over = 1;
        }
    }
    __syncthreads();
    for (int node = tid; node < M; node += kLbsBwdThreads) {
        const int cn = min(s_cnt[node], kLbsSlots);
        if (cn == 0) continue;
        s_cnt[node] = 0;
        float* row = s_tab + (size_t)node * G;
        // all reads of a row are independent (static unroll): one LDS latency per row instead of one per element
        float accv[kLbsAttr + kLbsHmax + 2];
#pragma unroll
        for (int c = 0; c < kLbsAttr + kLbsHmax + 2; c++) accv[c] = c < G ? row[c] : 0.f;
        for (int e = 0; e < cn; e++) {
            const float* src = s_exch + (int)s_slot[node * kLbsSlots + e] * GS;
#pragma unroll
            for (int c = 0; c < kLbsAttr + kLbsHmax + 2; c++) accv[c] += c < G ? src[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < kLbsAttr + kLbsHmax + 2; c++)
            if (c < G) row[c] = accv[c];
    }
    __syncthreads();
    if (*s_over) {   // workgroup-uniform; ~1 round in 6 on the metric scene has a node with more than kLbsSlots rows
        if (valid && pos >= kLbsSlots) {
            float* row = s_tab + (size_t)j * G;
#pragma unroll
            for (int c = 0; c < kLbsAttr + kLbsHmax + 2; c++)
                if (c < G) atomicAdd(row + c, cv[c]);
        }
        __syncthreads();
        if (tid == 0) *s_over = 0;
        // the owners above zero only the counters they served; counters of overfull nodes were zeroed too (cn > 0)
        __syncthreads();
    }
}

__host__ __device__ inline int lbs_exch_stride(int G) { return G | 1; }   // odd row stride: conflict-free row writes
inline size_t lbs_bwd_lds_bytes(int M, int H)
{
    const int G = kLbsAttr + H + 2;
    return ((size_t)M * G + (size_t)kLbsBwdThreads * lbs_exch_stride(G)) * sizeof(float) + (size_t)M * sizeof(int) +
           (size_t)M * kLbsSlots * sizeof(unsigned short) + 16;
}

// Coherent variant (COH): for surfels STORED IN THE ORDER OF THEIR NEAREST CONTROL NODE (Trainer.sort_surfels) the 64 points
// of a wave share one or two nodes in the first neighbour slot and ~10 in the others.  The wave sums each node's
// contributions across its lanes (wave_reduce.h: permlane swaps + DPP, no LDS) and issues ONE 23-lane global atomic per
// (wave, node) into a single [M][G] table -- no per-workgroup tables (24 MB of partials to write and re-read), no LDS, 782
// small workgroups instead of 256 large ones.  Measured at 200 k surfels / 1024 nodes: 97 + 15 us (LDS tables + reduction of
// the partials) -> 77 + 5 us; the kernel is memory-latency bound either way (3 waves per SIMD in total, PMC: 62 % of the
// wave cycles parked on s_waitcnt); a variant that first combined the waves of a 512-thread workgroup in an LDS hash table
// (4x fewer global atomics) measured 96 us, issuing every atomic of a point after its last load / store 78 us.
// What it waits for is RESIDENCY: 200 k surfels are 3125 waves, 153 VGPRs allow 3 waves per SIMD = 3072 -- the last 53 waves
// (14 workgroups) run in a second round and the kernel takes two wave lifetimes.  Forcing 128 VGPRs spilled 68 registers
// (82 us); specialising the kernel for the trainer's hyper dimension (template HT = 8: arrays sized for it, static indices)
// needs 127 without a spill: all waves resident in one round, 77 -> 53 us.
// Correct for any order; an unsorted cloud makes the loop below run once per DISTINCT node of a wave (up to 64 times).
constexpr int kCohThreads = 256;

// FIXED (dgs_deform_backward accumulate bit 4): the table holds 64-bit fixed-point sums (units of 2^-44) added with INTEGER
// atomics -- order-free, so the node gradients are bit-identical from run to run (Trainer.set_deterministic); the reduce converts.
__device__ __forceinline__ unsigned long long lbs_to_fixed44(float v)
{
    v = fminf(fmaxf(v, -262144.0f), 262144.0f);
    return (unsigned long long)__float2ll_rn(v * 17592186044416.0f);
}
__device__ __forceinline__ float lbs_from_fixed44(unsigned long long v) { return (float)((double)(long long)v * (1.0 / 17592186044416.0)); }

template <bool FIXED, int CVN>
__device__ __forceinline__ void lbs_combine(bool valid, int j, const float (&cv)[CVN], int G, float* __restrict__ table)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {
        const int jl = __builtin_amdgcn_readlane(j, __builtin_ctzll(todo));   // wave-uniform node id
        const bool sel = valid && j == jl;
        // columns 0..15, then 16..23 (G <= 24 here: 13 attributes + H <= 9 + 2; wider tables take the generic path below)
        float r0, r1 = 0.f, r2 = 0.f;
        {
            float lo[16];
#pragma unroll
            for (int c = 0; c < 16; c++) lo[c] = (sel && c < G && c < CVN) ? cv[c < CVN ? c : 0] : 0.f;
            r0 = dgs::wave_reduce16_dpp(lo);   // quad q holds the wave total of column q
        }
        {
            float hi[8];
#pragma unroll
            for (int c = 0; c < 8; c++) hi[c] = (sel && 16 + c < G && 16 + c < CVN) ? cv[16 + c < CVN ? 16 + c : 0] : 0.f;
            r1 = dgs::wave_reduce8_dpp(hi);    // lanes 8 k .. 8 k + 7 hold the total of column 16 + k
        }
        if (CVN > 24 && G > 24) {              // hyper_dim > 9: four more columns (wave-uniform, never taken by the trainer)
            float hi[8];
#pragma unroll
            for (int c = 0; c < 8; c++) hi[c] = (sel && 24 + c < G && 24 + c < CVN) ? cv[24 + c < CVN ? 24 + c : 0] : 0.f;
            r2 = dgs::wave_reduce8_dpp(hi);
        }
        const int sub4 = lane & 3, sub8 = lane & 7;
        const int col = sub4 == 0 ? (lane >> 2) : (sub8 == 1 ? 16 + (lane >> 3) : (sub8 == 2 ? 24 + (lane >> 3) : -1));
        if (col >= 0 && col < G) {
            const float tot = sub4 == 0 ? r0 : (sub8 == 1 ? r1 : r2);
            if (FIXED) atomicAdd(reinterpret_cast<unsigned long long*>(table) + (size_t)jl * G + col, lbs_to_fixed44(tot));
            else atomicAdd(table + (size_t)jl * G + col, tot);
        }
        todo &= ~__ballot(sel);
    }
}

// HT > 0: hyper dimension known at compile time (a.H == HT): arrays sized for it, static indexing (the trainer's H = 8)
template <bool ASM, bool COH, int HT = 0, bool FIXED = false>
__global__ void __launch_bounds__(COH ? kCohThreads : kLbsBwdThreads) lbs_bwd_kernel(LbsArgs a, const float* g_xyz, const float* g_rot, const float* g_scale,
                                                      float* g_feature, int gf_stride, int accumulate,
                                                      float* partial /*[kLbsBlocks][M][G], COH: [M][G] zeroed*/, int chunk, AsmArgs s_)
{
    extern __shared__ float s_tab[];  // [M][G], G = 13 + H + 2, then the exchange buffer and the integer arrays of lbs_deliver
    constexpr int HM = HT > 0 ? HT : kLbsHmax;
    const int H = HT > 0 ? HT : a.H;
    const int G = kLbsAttr + H + 2, GS = lbs_exch_stride(G);
    const int T = a.tstride;
    float* s_exch = s_tab + (size_t)a.M * G;
    int* s_cnt = reinterpret_cast<int*>(s_exch + (size_t)kLbsBwdThreads * GS);
    int* s_over = s_cnt + a.M;
    unsigned short* s_slot = reinterpret_cast<unsigned short*>(s_over + 2);
    constexpr int kThreads = COH ? kCohThreads : kLbsBwdThreads;
    if (!COH) {
        for (int i = threadIdx.x; i < a.M * G; i += kLbsBwdThreads) s_tab[i] = 0.f;
        for (int i = threadIdx.x; i < a.M; i += kLbsBwdThreads) s_cnt[i] = 0;
        if (threadIdx.x == 0) *s_over = 0;
        __syncthreads();
    }
    const int begin = blockIdx.x * chunk, end = min(a.N, (int)(blockIdx.x + 1) * chunk);
    for (int n0 = begin; n0 < end; n0 += kThreads) {   // uniform trip count: lbs_deliver synchronises the workgroup
        const int n = n0 + threadIdx.x;
        const bool valid = n < end;
        LbsPoint p;
        float xq[3 + HM];
        float gx[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, gs[2] = {0, 0};
        float inv = 0.f, m = 0.f;
        if (valid) {
            lbs_eval<HM>(a, n, p, xq);
            inv = 1.0f / p.W;
            m = a.mask ? a.mask[n] : 1.0f;
        if (!ASM) {
            for (int c = 0; c < 3; c++) gx[c] = g_xyz[3 * n + c] * m;
            for (int c = 0; c < 4; c++) gq[c] = g_rot[4 * n + c] * m;
            for (int c = 0; c < 2; c++) gs[c] = g_scale[2 * n + c] * m;
        } else {
            // adjoint of the activations (see AsmArgs); the deformation sees the detached centre, so the centre's own
            // gradient is just the incoming one
            for (int c = 0; c < 3; c++) {
                const float g = s_.g_means3D[3 * n + c];
                gx[c] = g * m;
                s_.g_xyz[3 * n + c] = accumulate ? s_.g_xyz[3 * n + c] + g : g;
            }
            for (int c = 0; c < 2; c++) {
                const float g = s_.g_scales[2 * n + c];
                gs[c] = g * m;
                const float v = g * expf(s_.scaling_raw[2 * n + c]);
                s_.g_scaling_raw[2 * n + c] = accumulate ? s_.g_scaling_raw[2 * n + c] + v : v;
            }
            float q[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < kLbsK; k++) {
                const f4u r4 = *reinterpret_cast<const f4u*>(a.attrs + (size_t)p.j[k] * kLbsAttr + 7);
                const float wn = p.w[k] * inv;
                q[0] += wn * r4.x; q[1] += wn * r4.y; q[2] += wn * r4.z; q[3] += wn * r4.w;
            }
            float v[4], n2 = 0.f, dot = 0.f;
            for (int c = 0; c < 4; c++) { v[c] = s_.rotation_raw[4 * n + c] + q[c] * m; n2 += v[c] * v[c]; }
            const float nrm = sqrtf(n2);
            const bool tiny = nrm < 1e-12f;
            const float invn = 1.0f / fmaxf(nrm, 1e-12f);
            for (int c = 0; c < 4; c++) dot += v[c] * invn * s_.g_rotations[4 * n + c];
            for (int c = 0; c < 4; c++) {
                const float g = tiny ? s_.g_rotations[4 * n + c] * invn : (s_.g_rotations[4 * n + c] - v[c] * invn * dot) * invn;
                gq[c] = g * m;
                s_.g_rotation_raw[4 * n + c] = accumulate ? s_.g_rotation_raw[4 * n + c] + g : g;
            }
            const float o = sigmoidf_(s_.opacity_raw[n]);
            const float go = s_.g_opacity[n] * o * (1.0f - o);
            s_.g_opacity_raw[n] = accumulate ? s_.g_opacity_raw[n] + go : go;
        }
        }
        float dwh[kLbsK] = {0, 0, 0}, mean = 0.f;  // d loss / d (normalised weight)
        if (valid) {
#pragma unroll
        for (int k = 0; k < kLbsK; k++) {
            float at[16];
            load_row(a.attrs + (size_t)p.j[k] * kLbsAttr, kLbsAttr, at);
            float v = p.Ax[k][0] * gx[0] + p.Ax[k][1] * gx[1] + p.Ax[k][2] * gx[2];
            for (int c = 0; c < 4; c++) v += at[7 + c] * gq[c];
            for (int c = 0; c < 2; c++) v += at[11 + c] * gs[c];
            dwh[k] = v;
            mean += p.w[k] * inv * v;
        }
        }
        float gfeat[HM];
        for (int h = 0; h < HM; h++) gfeat[h] = 0.f;
#pragma unroll
        for (int k = 0; k < kLbsK; k++) {
            float cv[kLbsAttr + HM + 2];   // this point's contribution to node p.j[k]: [attrs 13 | hyper H | radius | weight]
            int j = 0;
            if (valid) {
            j = p.j[k];
            const float wn = p.w[k] * inv;
            float nd[16], at[16];
            load_row(a.ntab + (size_t)j * T, T < 16 ? T : 16, nd);
            load_row(a.attrs + (size_t)j * kLbsAttr, 4, at);   // only the local-frame quaternion is needed here
            // ---- attributes: rotation quaternion through R, translation, rotation/scale residuals
            const float dA[3] = {wn * gx[0], wn * gx[1], wn * gx[2]};
            const float dl[3] = {xq[0] - nd[0], xq[1] - nd[1], xq[2] - nd[2]};
            float Gm[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Gm[3 * r + c] = dA[r] * dl[c];
            {
                const float r = at[0], i = at[1], jq = at[2], kq = at[3];
                const float n2 = r * r + i * i + jq * jq + kq * kq, two_s = 2.0f / n2;
                const float B[9] = {-(jq * jq + kq * kq), i * jq - kq * r, i * kq + jq * r, i * jq + kq * r, -(i * i + kq * kq),
                                    jq * kq - i * r, i * kq - jq * r, jq * kq + i * r, -(i * i + jq * jq)};
                float BG = 0.f;
                for (int c = 0; c < 9; c++) BG += B[c] * Gm[c];
                const float dBr = -kq * Gm[1] + jq * Gm[2] + kq * Gm[3] - i * Gm[5] - jq * Gm[6] + i * Gm[7];
                const float dBi = jq * (Gm[1] + Gm[3]) + kq * (Gm[2] + Gm[6]) - 2.f * i * (Gm[4] + Gm[8]) + r * (Gm[7] - Gm[5]);
                const float dBj = -2.f * jq * (Gm[0] + Gm[8]) + i * (Gm[1] + Gm[3]) + r * (Gm[2] - Gm[6]) + kq * (Gm[5] + Gm[7]);
                const float dBk = -2.f * kq * (Gm[0] + Gm[4]) + r * (Gm[3] - Gm[1]) + i * (Gm[2] + Gm[6]) + jq * (Gm[5] + Gm[7]);
                const float cs = -4.0f * BG / (n2 * n2);
                cv[0] = two_s * dBr + cs * r;
                cv[1] = two_s * dBi + cs * i;
                cv[2] = two_s * dBj + cs * jq;
                cv[3] = two_s * dBk + cs * kq;
            }
            for (int c = 0; c < 3; c++) cv[4 + c] = dA[c];
            for (int c = 0; c < 4; c++) cv[7 + c] = wn * gq[c];
            for (int c = 0; c < 2; c++) cv[11 + c] = wn * gs[c];
            // ---- weights: w = e * weight + 1e-7, e = exp(-dist / (2 r^2)), normalised over the K neighbours
            const float dw = (dwh[k] - mean) * inv;
            const float rad = p.rad[k], wg = p.wg[k];
            const float de = dw * wg * p.e[k];
            const float ddist = -de / (2.f * rad * rad);
            const float d_rad = de * p.dist[k] / (rad * rad * rad), d_w = dw * p.e[k];
#pragma unroll
            for (int h = 0; h < HM; h++) {
                const float gd = h < H ? 2.f * (xq[3 + h] - nd[3 + h]) * ddist : 0.f;
                gfeat[h] += gd;
                // columns 13 .. 13+H-1: node hyper coordinates, then radius, weight (static register indices only)
                cv[kLbsAttr + h] = h < H ? -gd : (h == H ? d_rad : (h == H + 1 ? d_w : 0.f));
            }
#pragma unroll
            for (int h = HM; h < HM + 2; h++) cv[kLbsAttr + h] = h == H ? d_rad : (h == H + 1 ? d_w : 0.f);
            }
            if (COH) lbs_combine<FIXED>(valid, j, cv, G, partial);
            else lbs_deliver(valid, j, cv, G, GS, a.M, s_tab, s_exch, s_cnt, s_slot, s_over);
        }
        if (valid)
        for (int h = 0; h < HM; h++)
            if (h < H) {
                float* dst = g_feature + (size_t)n * gf_stride + h;
                *dst = accumulate ? *dst + gfeat[h] : gfeat[h];
            }

    }
    if (COH) return;
    __syncthreads();
    float* dst = partial + (size_t)blockIdx.x * a.M * G;
    for (int i = threadIdx.x; i < a.M * G; i += kLbsBwdThreads) dst[i] = s_tab[i];
}

__global__ void __launch_bounds__(256) lbs_reduce_kernel(const float* partial, int M, int H, float* g_ntab, float* g_attrs, int nparts)
{
    const int G = kLbsAttr + H + 2, T = 3 + H + 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * G) return;
    float acc = 0.f;
    for (int b = 0; b < nparts; b++) acc += partial[(size_t)b * M * G + i];
    const int node = i / G, c = i - node * G;
    if (c < kLbsAttr) g_attrs[(size_t)node * kLbsAttr + c] = acc;
    else g_ntab[(size_t)node * T + 3 + (c - kLbsAttr)] = acc;
    if (c < 3) g_ntab[(size_t)node * T + c] = 0.f;  // node positions are detached in the reference
}

// raw-parameter variant: gradients of nodes[M, 3+H] (hyper columns), _node_radius (through exp) and _node_weight
// (through sigmoid), written or added in place; the attribute gradients are always written (the node MLP consumes them)
__global__ void __launch_bounds__(256) lbs_reduce_raw_kernel(const float* partial, int M, int H, const float* rad_raw,
                                                             const float* w_raw, float* g_nodes, float* g_rad_raw, float* g_w_raw,
                                                             float* g_attrs, int accumulate, int nparts, float* clear, int fixed = 0)
{
    const int G = kLbsAttr + H + 2, T = 3 + H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * G) return;
    float acc = 0.f;
    if (fixed) {   // one [M][G] table of 64-bit fixed-point sums (lbs_combine<FIXED>)
        unsigned long long* t64 = reinterpret_cast<unsigned long long*>(const_cast<float*>(partial));
        acc = lbs_from_fixed44(t64[i]);
        if (clear) t64[i] = 0ull;
    } else {
    for (int b = 0; b < nparts; b++) acc += partial[(size_t)b * M * G + i];
    if (clear) clear[i] = 0.f;   // coherent variant with a persistent table: leave it zeroed for the next backward (no memset launch)
    }
    const int node = i / G, c = i - node * G;
    if (c < kLbsAttr) { g_attrs[(size_t)node * kLbsAttr + c] = acc; }
    else if (c < kLbsAttr + H) {
        float* d = g_nodes + (size_t)node * T + 3 + (c - kLbsAttr);
        *d = accumulate ? *d + acc : acc;
    } else if (c == kLbsAttr + H) {
        const float v = acc * expf(rad_raw[node]);
        g_rad_raw[node] = accumulate ? g_rad_raw[node] + v : v;
    } else {
        const float w = sigmoidf_(w_raw[node]);
        const float v = acc * w * (1.0f - w);
        g_w_raw[node] = accumulate ? g_w_raw[node] + v : v;
    }
    if (c < 3 && !accumulate) g_nodes[(size_t)node * T + c] = 0.f;
}

// ---- densification statistics (train_gui.py:411, gaussian_model.py:484-486) ----------------------------------------
// one view: visible = radii > 0; grad_norm = |dL/dmeans2D[:, :2]| where visible
__global__ void __launch_bounds__(256) densify_view_kernel(int P, const int* radii, const float* g_means2D, float* grad_norm,
                                                           float* visible, int* radii_vis)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    const bool v = r > 0;
    const float gx = g_means2D[3 * i], gy = g_means2D[3 * i + 1];
    grad_norm[i] = v ? sqrtf(gx * gx + gy * gy) : 0.f;
    visible[i] = v ? 1.f : 0.f;
    radii_vis[i] = v ? r : 0;
}
// running statistics: xyz_gradient_accum += grad_norm, denom += visible, max_radii2D = max(max_radii2D, radii_vis)
__global__ void __launch_bounds__(256) densify_accum_kernel(int P, const float* grad_norm, const float* visible, const int* radii_vis,
                                                            float* accum, float* denom, int* max_radii, const int* skip)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || (skip && skip[0] != 0)) return;
    accum[i] += grad_norm[i];
    denom[i] += visible[i];
    max_radii[i] = max(max_radii[i], radii_vis[i]);
}

// ---- fused regulariser loss -----------------------------------------------------------------------------------------
__device__ __forceinline__ float clean_depth(float d)  // torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest
{
    if (d != d) return 0.f;
    if (d == INFINITY) return 0.f;
    if (d == -INFINITY) return -3.4028234663852886e38f;
    return d;
}

struct RegArgs {
    int H, W;
    const float* allmap; const float* rays_d; const float* rays_o; const float* wvt;
    float ln, ld;
    const float* const* rays_slot;   // non-null: rays_d = *rays_slot (chosen per graph replay by rewriting one pointer)
    int write_all;                   // backward: also store the zeros of planes 0, 1, 7 and of the border (caller zero-fills plane 5 only)
    float* zero_plane;               // forward: optional [H,W] plane to clear (the backward's atomics target: saves its fill launch)
};

// back-projected point of pixel (y, x)
__device__ __forceinline__ void reg_point(const RegArgs& a, int y, int x, float* p)
{
    const size_t q = (size_t)y * a.W + x;
    const float d = clean_depth(a.allmap[5 * (size_t)a.H * a.W + q]);
    p[0] = d * a.rays_d[3 * q] + a.rays_o[0];
    p[1] = d * a.rays_d[3 * q + 1] + a.rays_o[1];
    p[2] = d * a.rays_d[3 * q + 2] + a.rays_o[2];
}

// un-normalised normal v = dx x dy at an interior pixel, dx = p[y+1] - p[y-1], dy = p[x+1] - p[x-1]
__device__ __forceinline__ void reg_cross(const RegArgs& a, int y, int x, float* dx, float* dy, float* v)
{
    float pu[3], pd[3], pl[3], pr[3];
    reg_point(a, y + 1, x, pd); reg_point(a, y - 1, x, pu); reg_point(a, y, x + 1, pr); reg_point(a, y, x - 1, pl);
    for (int c = 0; c < 3; c++) { dx[c] = pd[c] - pu[c]; dy[c] = pr[c] - pl[c]; }
    v[0] = dx[1] * dy[2] - dx[2] * dy[1];
    v[1] = dx[2] * dy[0] - dx[0] * dy[2];
    v[2] = dx[0] * dy[1] - dx[1] * dy[0];
}

__global__ void __launch_bounds__(256) regloss_fwd_kernel(RegArgs a, float* loss, float* partial)
{
    if (a.rays_slot) a.rays_d = *a.rays_slot;
    __shared__ float s_red[4];
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    const size_t HW = (size_t)a.H * a.W;
    float val = 0.f;
    if (x < a.W && y < a.H) {
        const size_t q = (size_t)y * a.W + x;
        if (a.zero_plane) a.zero_plane[q] = 0.f;
        float dot = 0.f;
        if (x >= 1 && y >= 1 && x < a.W - 1 && y < a.H - 1) {
            float dx[3], dy[3], v[3];
            reg_cross(a, y, x, dx, dy, v);
            const float L = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float inv = a.allmap[HW + q] / fmaxf(L, 1e-12f);  // normalize(), then * alpha
            const float nv[3] = {a.allmap[2 * HW + q], a.allmap[3 * HW + q], a.allmap[4 * HW + q]};
            for (int c = 0; c < 3; c++) {
                const float nw = nv[0] * a.wvt[4 * c] + nv[1] * a.wvt[4 * c + 1] + nv[2] * a.wvt[4 * c + 2];  // n_view @ wvt[:3,:3].T
                dot += nw * v[c] * inv;
            }
        }
        val = (a.ln * (1.f - dot) + a.ld * a.allmap[6 * HW + q]) / (float)HW;
    }
    for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (partial) partial[blockIdx.y * gridDim.x + blockIdx.x] = v;   // see ssim_fwd_kernel
        else atomicAdd(loss, v);
    }
}

__global__ void __launch_bounds__(256) regloss_bwd_kernel(RegArgs a, const float* g, float* d_allmap)
{
    if (a.rays_slot) a.rays_d = *a.rays_slot;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.W || y >= a.H) return;
    const size_t HW = (size_t)a.H * a.W, q = (size_t)y * a.W + x;
    const float gs = g[0] / (float)HW;
    d_allmap[6 * HW + q] = gs * a.ld;
    const bool interior = x >= 1 && y >= 1 && x < a.W - 1 && y < a.H - 1;
    if (a.write_all) {
        d_allmap[q] = 0.f; d_allmap[HW + q] = 0.f; d_allmap[7 * HW + q] = 0.f;
        if (!interior) { d_allmap[2 * HW + q] = 0.f; d_allmap[3 * HW + q] = 0.f; d_allmap[4 * HW + q] = 0.f; }
    }
    if (!interior) return;
    float dx[3], dy[3], v[3];
    reg_cross(a, y, x, dx, dy, v);
    const float L = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float alpha = a.allmap[HW + q];
    const float denom = fmaxf(L, 1e-12f);
    const float n[3] = {v[0] / denom, v[1] / denom, v[2] / denom};
    const float nv[3] = {a.allmap[2 * HW + q], a.allmap[3 * HW + q], a.allmap[4 * HW + q]};
    float nw[3];
    for (int c = 0; c < 3; c++) nw[c] = nv[0] * a.wvt[4 * c] + nv[1] * a.wvt[4 * c + 1] + nv[2] * a.wvt[4 * c + 2];
    const float k = -gs * a.ln;
    // d / d rend_normal (view space): -lambda/HW * wvt[:3,:3]^T-rotated surf_normal
    for (int kk = 0; kk < 3; kk++) {
        float acc = 0.f;
        for (int c = 0; c < 3; c++) acc += a.wvt[4 * c + kk] * (n[c] * alpha);
        d_allmap[(2 + kk) * HW + q] = k * acc;
    }
    // d / d n (alpha is detached), then through F.normalize and the cross product
    float dn[3] = {k * nw[0] * alpha, k * nw[1] * alpha, k * nw[2] * alpha}, dv[3];
    if (L >= 1e-12f) {
        const float nd = n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2];
        for (int c = 0; c < 3; c++) dv[c] = (dn[c] - n[c] * nd) / L;
    } else {
        for (int c = 0; c < 3; c++) dv[c] = dn[c] / 1e-12f;
    }
    const float ddx[3] = {dy[1] * dv[2] - dy[2] * dv[1], dy[2] * dv[0] - dy[0] * dv[2], dy[0] * dv[1] - dy[1] * dv[0]};  // dy x dv
    const float ddy[3] = {dv[1] * dx[2] - dv[2] * dx[1], dv[2] * dx[0] - dv[0] * dx[2], dv[0] * dx[1] - dv[1] * dx[0]};  // dv x dx
    float* dd = d_allmap + 5 * HW;
    const int ys[4] = {y + 1, y - 1, y, y}, xs[4] = {x, x, x + 1, x - 1};
    const float sg[4] = {1.f, -1.f, 1.f, -1.f};
    for (int i = 0; i < 4; i++) {
        const size_t qq = (size_t)ys[i] * a.W + xs[i];
        const float raw = a.allmap[5 * HW + qq];
        if (raw != raw || raw == INFINITY || raw == -INFINITY) continue;  // nan_to_num has zero gradient there
        const float* gd = i < 2 ? ddx : ddy;
        atomicAdd(dd + qq, sg[i] * (gd[0] * a.rays_d[3 * qq] + gd[1] * a.rays_d[3 * qq + 1] + gd[2] * a.rays_d[3 * qq + 2]));
    }
}

// ---- regularisers, value AND gradient in one kernel (unit upstream gradient) -----------------------------------------
// The train step differentiates loss = photometric + regularisers with dL/dloss = 1, so the regularisers' gradient image depends on
// the rasterizer outputs only and can be produced next to the value: one pass over the allmap instead of two (regloss_fwd_kernel +
// regloss_bwd_kernel read the same planes twice), and the depth gradient is GATHERED -- every pixel sums the four neighbouring
// normals' contributions from LDS -- instead of 4 float atomics per pixel into a pre-cleared plane (2.6 M atomics at 800 x 800).
//   workgroup = 30 x 14 pixels; normals ("centres") are needed on 32 x 16, back-projected points on 34 x 18
//   phase 1: points of the 34 x 18 region -> LDS (every global load of the 3 trips in flight before the first LDS store)
//   phase 2: thread = centre (2 trips of 32 x 8): cross product, normalisation, loss term, d/d rend_normal, and the two
//            vectors ddx = dy x dv, ddy = dv x dx its four neighbours' points receive -> LDS
//   phase 3: the centre's own thread gathers  +ddx(y-1) - ddx(y+1) + ddy(x-1) - ddy(x+1),  dots with its ray, stores all 8 planes
// 11 + 23 -> 17 us at 800 x 800 (forward + backward kernels -> this one).
constexpr int kRW = 30, kRH = 14;                 // own pixels of a workgroup
constexpr int kCW = kRW + 2, kCH = kRH + 2;       // centres: 32 x 16
constexpr int kQW = kRW + 4, kQH = kRH + 4;       // points: 34 x 18
static_assert(kCW == 32 && kCH == 16, "thread maps below");

constexpr int kRegLds = 3 * kQH * (kQW + 1) + 6 * kCH * (kCW + 1) + 4;   // floats

// workgroup (bx, by) of a grid gx wide; `lds` = kRegLds floats
__device__ __forceinline__ void regloss_fused_body(RegArgs a, float* __restrict__ partial, float* __restrict__ d_allmap, int bx, int by, int gx,
                                                   float* __restrict__ lds)
{
    if (a.rays_slot) a.rays_d = *a.rays_slot;
    float (*s_p)[kQH][kQW + 1] = reinterpret_cast<float (*)[kQH][kQW + 1]>(lds);
    float (*s_g)[kCH][kCW + 1] = reinterpret_cast<float (*)[kCH][kCW + 1]>(lds + 3 * kQH * (kQW + 1));
    float* s_red = lds + 3 * kQH * (kQW + 1) + 6 * kCH * (kCW + 1);
    const int tid = threadIdx.x;
    const int x0 = bx * kRW, y0 = by * kRH;      // first own pixel
    const unsigned HW = (unsigned)a.H * (unsigned)a.W;
    const GlobalF am = (GlobalF)a.allmap, rd = (GlobalF)a.rays_d;
    const float ox = a.rays_o[0], oy = a.rays_o[1], oz = a.rays_o[2];
    {
        constexpr int kTrips = (kQW * kQH + 255) / 256;          // 3
        float d[kTrips], r0[kTrips], r1[kTrips], r2[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; t++) {
            const int i = min(tid + 256 * t, kQW * kQH - 1), r = i / kQW, c = i - r * kQW;
            const unsigned q = (unsigned)min(max(y0 + r - 2, 0), a.H - 1) * (unsigned)a.W + (unsigned)min(max(x0 + c - 2, 0), a.W - 1);
            d[t] = am[5 * HW + q]; r0[t] = rd[3 * q]; r1[t] = rd[3 * q + 1]; r2[t] = rd[3 * q + 2];
        }
#pragma unroll
        for (int t = 0; t < kTrips; t++) {
            const int i = tid + 256 * t, r = i / kQW, c = i - r * kQW;
            if (i < kQW * kQH) {       // points outside the image are only read by centres that are not interior (their vectors are 0)
                const float dc = clean_depth(d[t]);
                s_p[0][r][c] = dc * r0[t] + ox; s_p[1][r][c] = dc * r1[t] + oy; s_p[2][r][c] = dc * r2[t] + oz;
            }
        }
    }
    const int cx = tid & 31;
    float wv[9];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) wv[3 * c + k] = a.wvt[4 * c + k];
    const float inv_hw = 1.0f / (float)HW;
    const float kn = -inv_hw * a.ln;
    // the centres' own planes and (for phase 3) the own pixels' ray and raw depth: issued before the barrier
    float al[2], n0[2], n1[2], n2[2], ds[2], q0[2], q1[2], q2[2], raw[2];
    bool own[2], interior[2];
    unsigned qq[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int cy = (tid >> 5) + 8 * t;
        const int x = x0 + cx - 1, y = y0 + cy - 1;
        own[t] = cx >= 1 && cx <= kRW && cy >= 1 && cy <= kRH && x < a.W && y < a.H;
        interior[t] = x >= 1 && y >= 1 && x < a.W - 1 && y < a.H - 1;
        qq[t] = (unsigned)min(max(y, 0), a.H - 1) * (unsigned)a.W + (unsigned)min(max(x, 0), a.W - 1);
        al[t] = am[HW + qq[t]]; n0[t] = am[2 * HW + qq[t]]; n1[t] = am[3 * HW + qq[t]]; n2[t] = am[4 * HW + qq[t]];
        ds[t] = am[6 * HW + qq[t]]; raw[t] = am[5 * HW + qq[t]];
        q0[t] = rd[3 * qq[t]]; q1[t] = rd[3 * qq[t] + 1]; q2[t] = rd[3 * qq[t] + 2];
    }
    __syncthreads();
    float val = 0.f;
    float gn[2][3];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int cy = (tid >> 5) + 8 * t;
        float ddx[3] = {0.f, 0.f, 0.f}, ddy[3] = {0.f, 0.f, 0.f};
        float dot = 0.f;
        gn[t][0] = gn[t][1] = gn[t][2] = 0.f;
        if (interior[t]) {
            // centre (cy, cx) is point (cy + 1, cx + 1) of the region
            float dx[3], dy[3], v[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dx[c] = s_p[c][cy + 2][cx + 1] - s_p[c][cy][cx + 1];
                dy[c] = s_p[c][cy + 1][cx + 2] - s_p[c][cy + 1][cx];
            }
            v[0] = dx[1] * dy[2] - dx[2] * dy[1];
            v[1] = dx[2] * dy[0] - dx[0] * dy[2];
            v[2] = dx[0] * dy[1] - dx[1] * dy[0];
            const float L = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float denom = fmaxf(L, 1e-12f);
            const float n[3] = {v[0] / denom, v[1] / denom, v[2] / denom};
            float nw[3];
#pragma unroll
            for (int c = 0; c < 3; c++) nw[c] = n0[t] * wv[3 * c] + n1[t] * wv[3 * c + 1] + n2[t] * wv[3 * c + 2];   // n_view @ wvt[:3,:3].T
            const float inv = al[t] / denom;                                     // normalize(), then * alpha
#pragma unroll
            for (int c = 0; c < 3; c++) dot += nw[c] * v[c] * inv;
            // d / d rend_normal (view space): -lambda/HW * wvt[:3,:3]^T-rotated surf_normal
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 3; c++) acc += wv[3 * c + kk] * (n[c] * al[t]);
                gn[t][kk] = kn * acc;
            }
            // d / d n (alpha is detached), then through F.normalize and the cross product
            const float dn[3] = {kn * nw[0] * al[t], kn * nw[1] * al[t], kn * nw[2] * al[t]};
            float dv[3];
            if (L >= 1e-12f) {
                const float nd = n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2];
#pragma unroll
                for (int c = 0; c < 3; c++) dv[c] = (dn[c] - n[c] * nd) / L;
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) dv[c] = dn[c] / 1e-12f;
            }
            ddx[0] = dy[1] * dv[2] - dy[2] * dv[1]; ddx[1] = dy[2] * dv[0] - dy[0] * dv[2]; ddx[2] = dy[0] * dv[1] - dy[1] * dv[0];   // dy x dv
            ddy[0] = dv[1] * dx[2] - dv[2] * dx[1]; ddy[1] = dv[2] * dx[0] - dv[0] * dx[2]; ddy[2] = dv[0] * dx[1] - dv[1] * dx[0];   // dv x dx
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { s_g[c][cy][cx] = ddx[c]; s_g[3 + c][cy][cx] = ddy[c]; }
        if (own[t]) val += (a.ln * (1.f - dot) + a.ld * ds[t]) * inv_hw;
    }
    __syncthreads();
    typedef float __attribute__((address_space(1)))* GlobalW;
    const GlobalW out = (GlobalW)d_allmap;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!own[t]) continue;
        const int cy = (tid >> 5) + 8 * t;
        float g3[3];
#pragma unroll
        for (int c = 0; c < 3; c++) g3[c] = s_g[c][cy - 1][cx] - s_g[c][cy + 1][cx] + s_g[3 + c][cy][cx - 1] - s_g[3 + c][cy][cx + 1];
        const float r = raw[t];
        const bool finite = !(r != r || r == INFINITY || r == -INFINITY);        // nan_to_num has zero gradient there
        const unsigned q = qq[t];
        out[q] = 0.f; out[HW + q] = 0.f;
        out[2 * HW + q] = gn[t][0]; out[3 * HW + q] = gn[t][1]; out[4 * HW + q] = gn[t][2];
        out[5 * HW + q] = finite ? g3[0] * q0[t] + g3[1] * q1[t] + g3[2] * q2[t] : 0.f;
        out[6 * HW + q] = inv_hw * a.ld;
        out[7 * HW + q] = 0.f;
    }
    for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) partial[by * gx + bx] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ void __launch_bounds__(256) regloss_fused_kernel(RegArgs a, float* __restrict__ partial, float* __restrict__ d_allmap)
{
    __shared__ float lds[kRegLds];
    regloss_fused_body(a, partial, d_allmap, blockIdx.x, blockIdx.y, gridDim.x, lds);
}

// Both forward halves of the loss in ONE launch (dgs_loss_forward_merged).  The photometric kernel (SSIM windows: FMA-bound with a
// store-heavy epilogue) and the regulariser kernel (three short phases between barriers: latency-bound) read different
// rasterizer outputs and write different buffers; launched one after the other each leaves the chip partly idle (1305 and 1566
// workgroups at 800 x 800: a second partial round of workgroups each) and pays its own launch gap.  Here the two kinds of
// workgroup alternate in one grid while both last, so that every CU holds both at once.
__global__ void __launch_bounds__(256) loss_fwd_merged_kernel(SsimFwdArgs A, Gauss g, int sgx, int sgy, int sgz, RegArgs a,
                                                              float* __restrict__ reg_partial, float* __restrict__ d_allmap, int rgx, int rgy)
{
    constexpr int kMergedLds = kSsimFwdLds > kRegLds ? kSsimFwdLds : kRegLds;
    __shared__ float lds[kMergedLds];
    const int ns = sgx * sgy * sgz, nr = rgx * rgy, both = 2 * min(ns, nr);
    const int bid = blockIdx.x;
    bool photo; int i;
    if (bid < both) { photo = !(bid & 1); i = bid >> 1; }
    else { photo = ns > nr; i = bid - both + min(ns, nr); }
    if (photo) {
        const int bz = i / (sgx * sgy), r = i - bz * (sgx * sgy), by = r / sgx, bx = r - by * sgx;
        ssim_fwd_body(A, g, bx, by, bz, sgx, sgy, sgz, lds);
    } else {
        const int by = i / rgx, bx = i - by * rgx;
        regloss_fused_body(a, reg_partial, d_allmap, bx, by, rgx, lds);
    }
}

// ---- flat Adam --------------------------------------------------------------------------------------------------
constexpr int kAdamSeg = 64;
constexpr int kAdamChunk = 1024;   // elements per workgroup (256 threads x 4: every load of a thread in flight at once -- with 16 the
                                   // update of the 0.5 M deformation parameters was four dependent memory round trips, 19 us)

struct AdamSegs {
    float* p[kAdamSeg];
    long long off[kAdamSeg + 1];
    float lr[kAdamSeg];
    // optional periodic learning-rate pattern inside a segment: element i uses lr2 when (i % period) >= split
    // (e.g. SH coefficients stored [P,16,3]: the DC term and the higher bands have different rates); period 0 = off
    float lr2[kAdamSeg];
    int period[kAdamSeg];
    int split[kAdamSeg];
    // optional exponential schedule of lr (get_expon_lr_func, utils/general_utils.py:49-83, lr_delay_steps = 0), evaluated on
    // the device from the step counter so that a captured step needs no host update: sched_steps 0 = constant
    float lr_final[kAdamSeg];
    float sched_steps[kAdamSeg];
    float sched_t0;
    // step origin of a segment: its bias corrections use t - t_origin.  torch.optim.Adam counts steps PER PARAMETER and skips
    // parameters without a gradient, so a parameter that joins the optimisation late (the reference's deformation and `feature`
    // after the warm-up, train_gui.py:281-285) starts at step 1 while the run's counter t is in the thousands
    float t_origin[kAdamSeg];
    float gscale;   // gradients are read as grad * gscale (data parallel: the bucket holds the SUM over ranks, gscale = 1 / world)
    int zero_grad;  // the gradient is cleared behind the read (optimizer.step() + zero_grad() in one pass; also on a skipped step)
};

// Guard of a training step (dgs_step_guard): skip[0] != 0 means "this step must not change anything" -- a rank's rasterizer
// ran out of list capacity and rendered background (under data parallelism the flag rides in the MAX all-reduce of the radii,
// so every rank sees the same value) -- and the update kernels return without touching parameters, moments, statistics or
// the step count.
// First node of a captured step: the view of this replay.  row_out <- table[v] with v = override[0] if it is >= 0 (then reset to -1),
// else (counter[0] * stride + offset) mod nrows; counter[0] += 1.  A step that walks its views in the default order needs no host
// copy in front of the replay (the 256-byte row copy + the idle device behind it were ~10 us of every 0.8 ms step).
__global__ void __launch_bounds__(64) select_row_kernel(mlp::SelectArgs q) { mlp::select_row_body(q); }

__global__ void step_guard_kernel(const int* __restrict__ skip, float* __restrict__ step_count, float* __restrict__ status,
                                  float* __restrict__ host_ring, int ring_len, const float* __restrict__ loss)
{
    step_guard_body(skip, step_count, status, host_ring, ring_len, loss ? loss[0] : 0.0f);   // one thread
}

// grad2 (nullable): a second gradient buffer of the same layout, ADDED to the first on the fly -- the two views of a step that were
// rendered concurrently keep a bucket each (Trainer.concurrent_views) and the update reads their sum without an adding pass.
// NT (DGS_ADAM_NT: 0 off, 1 moments + gradient = the default, 2 the parameter store too): the update's streams carry the non-temporal
// hint -- 320 MB per step that nothing reads again before the next step, passing through the L2s next to the working set of the node-MLP
// backward chain that runs beside the update (the step's critical path)
typedef float adam_f4 __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void adam_block(const AdamSegs& sg, const int2 pl, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                           const float t, float b1, float b2, float eps, const bool sk, const float* __restrict__ grad2)
{
    // pl = (segment, first element of this block inside the segment)
    const int s = pl.x;
    const long long seg_len = sg.off[s + 1] - sg.off[s];
    const float ts = fmaxf(t - sg.t_origin[s], 1.0f);
    const float bc1 = 1.0f - powf(b1, ts), bc2 = 1.0f - powf(b2, ts);
    float lr = sg.lr[s];
    if (sg.sched_steps[s] > 0.0f) {
        // the reference sets the rate AFTER optimizer.step(): step t runs at schedule(t - 1)
        const float tau = fminf(fmaxf((t - 1.0f + sg.sched_t0) / sg.sched_steps[s], 0.0f), 1.0f);
        lr = expf(logf(lr) * (1.0f - tau) + logf(sg.lr_final[s]) * tau);
    }
    const float step_size = lr / bc1, step_size2 = sg.lr2[s] / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const unsigned period = (unsigned)sg.period[s], split = (unsigned)sg.split[s];
    float* __restrict__ p = sg.p[s];
    const long long base = sg.off[s];
    auto update = [&](float g, float& mi, float& vi, float& pi, long long i) {
        mi = b1 * mi + (1.0f - b1) * g;
        vi = b2 * vi + (1.0f - b2) * g * g;
        const float ss = (period && (unsigned)(i % period) >= split) ? step_size2 : step_size;
        pi -= ss * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    };
    static_assert(kAdamChunk == 4 * 256, "one 16-byte vector per thread and stream");
    const long long i0 = (long long)pl.y + 4 * threadIdx.x;
    // a thread owns 4 consecutive elements: one 16-byte access per stream instead of four 4-byte ones (which keep the address unit
    // busy four times as long: 28 accesses per 4 elements were ~40 us of it per launch) -- when the block's 4 KB of every stream
    // are 16-byte aligned and inside the segment
    const bool vec = ((base + pl.y) & 3) == 0 && (reinterpret_cast<size_t>(p + pl.y) & 15) == 0 && i0 + 3 < seg_len;
    if (vec) {
        float4* gq = reinterpret_cast<float4*>(grad + base + i0);
        float4* mq = reinterpret_cast<float4*>(m + base + i0);
        float4* vq = reinterpret_cast<float4*>(v + base + i0);
        float4* pq = reinterpret_cast<float4*>(p + i0);
        float4 g4;
        if (NT) { const adam_f4 t4 = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(gq)); g4 = make_float4(t4.x, t4.y, t4.z, t4.w); }
        else g4 = *gq;
        if (grad2) {
            const float4 h4 = *reinterpret_cast<const float4*>(grad2 + base + i0);
            g4.x += h4.x; g4.y += h4.y; g4.z += h4.z; g4.w += h4.w;
        }
        if (sg.zero_grad) *gq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sk) return;
        float4 m4, v4, p4 = *pq;
        if (NT) {
            const adam_f4 a4 = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(mq)), b4 = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(vq));
            m4 = make_float4(a4.x, a4.y, a4.z, a4.w); v4 = make_float4(b4.x, b4.y, b4.z, b4.w);
        } else { m4 = *mq; v4 = *vq; }
        update(g4.x * sg.gscale, m4.x, v4.x, p4.x, i0);
        update(g4.y * sg.gscale, m4.y, v4.y, p4.y, i0 + 1);
        update(g4.z * sg.gscale, m4.z, v4.z, p4.z, i0 + 2);
        update(g4.w * sg.gscale, m4.w, v4.w, p4.w, i0 + 3);
        if (NT) {
            __builtin_nontemporal_store(adam_f4{m4.x, m4.y, m4.z, m4.w}, reinterpret_cast<adam_f4*>(mq));
            __builtin_nontemporal_store(adam_f4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<adam_f4*>(vq));
            if (NT == 2) __builtin_nontemporal_store(adam_f4{p4.x, p4.y, p4.z, p4.w}, reinterpret_cast<adam_f4*>(pq));
            else *pq = p4;
        } else { *mq = m4; *vq = v4; *pq = p4; }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = i0 + k;
        if (i < seg_len) {
            const float g = (grad[base + i] + (grad2 ? grad2[base + i] : 0.0f)) * sg.gscale;
            if (sg.zero_grad) grad[base + i] = 0.0f;
            if (sk) continue;
            float mi = m[base + i], vi = v[base + i], pi = p[i];
            update(g, mi, vi, pi, i);
            m[base + i] = mi;
            v[base + i] = vi;
            p[i] = pi;
        }
    }
}

// nblocks plan entries over gridDim.x workgroups (at most 4096 by default: see the launch)
template <int NT>
__global__ void __launch_bounds__(256) adam_kernel(AdamSegs sg, const int2* __restrict__ plan, int nblocks, float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, const float* __restrict__ step_count,
                                                   float b1, float b2, float eps, const int* __restrict__ skip, const float* __restrict__ grad2)
{
    const bool sk = skip && skip[0] != 0;   // guarded step (see step_guard_kernel)
    if (sk && !sg.zero_grad) return;
    const float t = step_count[0];
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) adam_block<NT>(sg, plan[b], grad, m, v, t, b1, b2, eps, sk, grad2);
}

long long adam_blocks(int nseg, const long long* off)
{
    long long nb = 0;
    for (int s = 0; s < nseg; s++) nb += (off[s + 1] - off[s] + kAdamChunk - 1) / kAdamChunk;
    return nb;
}

}  // namespace

extern "C" {

int dgs_train_ops_abi_version(void) { return DGS_TRAIN_OPS_ABI_VERSION; }
const char* dgs_train_ops_last_error(void) { return g_err.c_str(); }

int dgs_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_sum, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_sum) return fail(-1, "dgs_ssim_forward: bad argument");
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))
        return fail(-1, "dgs_ssim_forward: pass all three derivative maps or none");
    static const Gauss g = make_gauss();
    dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, C);
    const SsimFwdArgs A{H, W, img1, img2, ssim_sum, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("ssim_fwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dmean, float* dL_dimg1,
                      void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dmean || !dL_dimg1)
        return fail(-1, "dgs_ssim_backward: bad argument");
    static const Gauss g = make_gauss();
    dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, C);
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, inv_n, 0.f, img1, img2, g, dm_dmu1, dm_dsigma1_sq,
                       dm_dsigma12, dL_dmean, dL_dimg1, (const float* const*)nullptr, CombineArgs{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("ssim_bwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_lbs_supported(int M, int H) { return H >= 0 && H <= kLbsHmax && M > 0 && M <= 4 * kLbsBwdThreads && lbs_bwd_lds_bytes(M, H) <= 160 * 1024; }
size_t dgs_lbs_scratch_bytes(int M, int H) { return (size_t)kLbsBlocks * (size_t)M * (size_t)(kLbsAttr + H + 2) * sizeof(float); }

static int lbs_check(int N, int M, int H)
{
    if (N < 0 || M <= 0 || H < 0 || H > kLbsHmax) return fail(-1, "dgs_lbs: bad sizes");
    return 0;
}

int dgs_lbs_forward(int N, int M, int H, const float* x, const float* feature, int feature_stride, const long long* idx,
                    const float* ntab, const float* attrs, const float* mask, float* d_xyz, float* d_rot, float* d_scale,
                    void* stream)
{
    if (int e = lbs_check(N, M, H)) return e;
    if (N == 0) return 0;
    LbsArgs a{N, M, H, feature_stride, x, feature, idx, ntab, attrs, mask, 3 + H + 2, nullptr, nullptr};
    hipLaunchKernelGGL(lbs_fwd_kernel<false>, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, d_xyz, d_rot, d_scale,
                       AsmArgs{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("lbs_fwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_lbs_backward(int N, int M, int H, const float* x, const float* feature, int feature_stride, const long long* idx,
                     const float* ntab, const float* attrs, const float* mask, const float* g_xyz, const float* g_rot,
                     const float* g_scale, float* g_feature, float* g_ntab, float* g_attrs, void* scratch, void* stream)
{
    if (int e = lbs_check(N, M, H)) return e;
    const int G = kLbsAttr + H + 2;
    const size_t lds = lbs_bwd_lds_bytes(M, H);
    if (lds > 160 * 1024 || M > 4 * kLbsBwdThreads) return fail(-2, "dgs_lbs_backward: node tables do not fit the 160 KB of LDS");
    if (!scratch) return fail(-1, "dgs_lbs_backward: scratch is NULL");
    LbsArgs a{N, M, H, feature_stride, x, feature, idx, ntab, attrs, mask, 3 + H + 2, nullptr, nullptr};
    const int chunk = (N + kLbsBlocks - 1) / kLbsBlocks;
    hipLaunchKernelGGL((lbs_bwd_kernel<false, false>), dim3(kLbsBlocks), dim3(kLbsBwdThreads), lds, (hipStream_t)stream, a, g_xyz, g_rot, g_scale,
                       g_feature, H, 0, (float*)scratch, chunk > 0 ? chunk : 1, AsmArgs{});
    hipLaunchKernelGGL(lbs_reduce_kernel, dim3((M * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, M, H,
                       g_ntab, g_attrs, kLbsBlocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("lbs_bwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_regloss_forward(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                        float lambda_normal, float lambda_dist, float* loss, void* stream)
{
    if (H <= 0 || W <= 0 || !allmap || !rays_d || !rays_o || !wvt || !loss) return fail(-1, "dgs_regloss_forward: bad argument");
    RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, nullptr, 0, nullptr};
    hipLaunchKernelGGL(regloss_fwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, loss, (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("regloss_fwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_regloss_backward(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                         float lambda_normal, float lambda_dist, const float* g, float* d_allmap, void* stream)
{
    if (H <= 0 || W <= 0 || !allmap || !rays_d || !rays_o || !wvt || !g || !d_allmap) return fail(-1, "dgs_regloss_backward: bad argument");
    RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, nullptr, 0, nullptr};
    hipLaunchKernelGGL(regloss_bwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, g, d_allmap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("regloss_bwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_regloss_backward_slot(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                              float lambda_normal, float lambda_dist, const float* g, float* d_allmap, const float* const* rays_slot,
                              int write_all, void* stream)
{
    if (H <= 0 || W <= 0 || !allmap || (!rays_d && !rays_slot) || !rays_o || !wvt || !g || !d_allmap)
        return fail(-1, "dgs_regloss_backward_slot: bad argument");
    RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, rays_slot, write_all, nullptr};
    hipLaunchKernelGGL(regloss_bwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, g, d_allmap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("regloss_bwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_deform_reduce(int M, int H, const float* node_radius_raw, const float* node_weight_raw, float* g_nodes, float* g_radius_raw,
                      float* g_weight_raw, float* g_attrs, int accumulate, void* scratch, void* stream)
{
    if (M <= 0 || H < 0 || H > kLbsHmax || !scratch || !node_radius_raw || !node_weight_raw || !g_nodes || !g_radius_raw || !g_weight_raw || !g_attrs)
        return fail(-1, "dgs_deform_reduce: bad argument");
    const int G = kLbsAttr + H + 2;
    hipLaunchKernelGGL(lbs_reduce_raw_kernel, dim3((M * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, M, H,
                       node_radius_raw, node_weight_raw, g_nodes, g_radius_raw, g_weight_raw, g_attrs, accumulate & 1, 1,
                       (accumulate & 4) ? (float*)scratch : (float*)nullptr, (accumulate & 16) ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "lbs_reduce_raw_kernel: launch failed");
}

size_t dgs_adam_plan_bytes(long long total) { return (size_t)(total / kAdamChunk + kAdamSeg + 1) * sizeof(int2); }

int dgs_adam_plan(int nseg, const long long* offsets, void* plan, void* stream)
{
    if (nseg <= 0 || nseg > kAdamSeg || !offsets || !plan) return fail(-1, "dgs_adam_plan: bad argument");
    std::vector<int2> host;
    for (int s = 0; s < nseg; s++)
        for (long long b = 0; b < offsets[s + 1] - offsets[s]; b += kAdamChunk) host.push_back(make_int2(s, (int)b));
    if (host.empty()) return 0;
    hipError_t e = hipMemcpyAsync(plan, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);  // `host` dies at return
    if (e != hipSuccess) return fail(-4, std::string("dgs_adam_plan: ") + hipGetErrorString(e));
    return 0;
}

int dgs_adam_step_pattern(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* grad, float* exp_avg, float* exp_avg_sq,
                          const float* step_count, float beta1, float beta2, float eps, const void* plan, void* stream);

int dgs_adam_step_sched(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                        const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                        float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1, float beta2,
                        float eps, const void* plan, void* stream);

int dgs_adam_step_guarded(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                          float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1,
                          float beta2, float eps, const void* plan, const int* skip, void* stream);
int dgs_adam_step_zero(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                       const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                       float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                       float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream);
int dgs_adam_step_origin(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                         const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                         const float* step_origins, float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq,
                         const float* step_count, float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream);

int dgs_adam_step(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* grad, float* exp_avg,
                  float* exp_avg_sq, const float* step_count, float beta1, float beta2, float eps, const void* plan, void* stream)
{
    return dgs_adam_step_pattern(nseg, params, offsets, lrs, nullptr, nullptr, nullptr, grad, exp_avg, exp_avg_sq, step_count, beta1,
                                 beta2, eps, plan, stream);
}

int dgs_adam_step_pattern(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* grad, float* exp_avg, float* exp_avg_sq,
                          const float* step_count, float beta1, float beta2, float eps, const void* plan, void* stream)
{
    return dgs_adam_step_sched(nseg, params, offsets, lrs, lrs2, periods, splits, nullptr, nullptr, 0.0f, 1.0f, grad, exp_avg, exp_avg_sq,
                               step_count, beta1, beta2, eps, plan, stream);
}

int dgs_adam_step_sched(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                        const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                        float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1, float beta2,
                        float eps, const void* plan, void* stream)
{
    return dgs_adam_step_guarded(nseg, params, offsets, lrs, lrs2, periods, splits, lrs_final, sched_steps, sched_t0, grad_scale, grad,
                                 exp_avg, exp_avg_sq, step_count, beta1, beta2, eps, plan, nullptr, stream);
}

int dgs_select_row(const float* table, int nrows, int row_floats, int* counter, int* override_, int stride, int offset, float* row_out, void* stream)
{
    if (!table || !counter || !override_ || !row_out || nrows <= 0 || row_floats <= 0 || stride <= 0 || offset < 0)
        return fail(-1, "dgs_select_row: bad argument");
    const mlp::SelectArgs q{table, nrows, row_floats, counter, override_, stride, offset, row_out};
    hipLaunchKernelGGL(select_row_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, q);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "select_row_kernel: launch failed");
}

int dgs_step_guard(const int* skip, float* step_count, float* status, float* host_ring, int ring_len, const float* loss, void* stream)
{
    if (!step_count || !status || (host_ring && ring_len <= 0)) return fail(-1, "dgs_step_guard: bad argument");
    hipLaunchKernelGGL(step_guard_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, skip, step_count, status, host_ring, ring_len, loss);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "step_guard_kernel: launch failed");
}

int dgs_adam_step_guarded(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                          float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1,
                          float beta2, float eps, const void* plan, const int* skip, void* stream)
{
    return dgs_adam_step_zero(nseg, params, offsets, lrs, lrs2, periods, splits, lrs_final, sched_steps, sched_t0, grad_scale,
                              const_cast<float*>(grad), 0, exp_avg, exp_avg_sq, step_count, beta1, beta2, eps, plan, skip, stream);
}

int dgs_adam_step_zero(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                       const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                       float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                       float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream)
{
    return dgs_adam_step_origin(nseg, params, offsets, lrs, lrs2, periods, splits, lrs_final, sched_steps, sched_t0, nullptr, grad_scale, grad,
                                zero_grad, exp_avg, exp_avg_sq, step_count, beta1, beta2, eps, plan, skip, stream);
}

int dgs_adam_step_origin(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                         const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                         const float* step_origins, float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq,
                         const float* step_count, float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream)
{
    return dgs_adam_step_sum2(nseg, params, offsets, lrs, lrs2, periods, splits, lrs_final, sched_steps, sched_t0, step_origins, grad_scale, grad,
                              nullptr, zero_grad, exp_avg, exp_avg_sq, step_count, beta1, beta2, eps, plan, skip, stream);
}

int dgs_adam_step_sum2(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                       const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                       const float* step_origins, float grad_scale, float* grad, const float* grad2, int zero_grad, float* exp_avg,
                       float* exp_avg_sq, const float* step_count, float beta1, float beta2, float eps, const void* plan, const int* skip,
                       void* stream)
{
    if (grad2 && zero_grad) return fail(-1, "dgs_adam_step_sum2: zero_grad clears the first buffer only; clear both yourself");
    if ((lrs_final != nullptr) != (sched_steps != nullptr)) return fail(-1, "dgs_adam_step_sched: pass lrs_final and sched_steps together");
    if (nseg <= 0 || nseg > kAdamSeg || !params || !offsets || !lrs || !grad || !exp_avg || !exp_avg_sq || !step_count || !plan)
        return fail(-1, "dgs_adam_step: bad argument");
    if ((lrs2 != nullptr) != (periods != nullptr) || (lrs2 != nullptr) != (splits != nullptr))
        return fail(-1, "dgs_adam_step_pattern: pass lrs2, periods and splits together");
    AdamSegs sg;
    for (int s = 0; s < nseg; s++) {
        sg.p[s] = params[s]; sg.off[s] = offsets[s]; sg.lr[s] = lrs[s];
        sg.lr2[s] = lrs2 ? lrs2[s] : lrs[s];
        sg.period[s] = periods ? periods[s] : 0;
        sg.split[s] = splits ? splits[s] : 0;
        if (sg.period[s] < 0 || sg.split[s] < 0) return fail(-1, "dgs_adam_step_pattern: negative period / split");
        sg.lr_final[s] = lrs_final ? lrs_final[s] : lrs[s];
        sg.sched_steps[s] = sched_steps ? sched_steps[s] : 0.0f;
        sg.t_origin[s] = step_origins ? step_origins[s] : 0.0f;
        // (a NEGATIVE origin is a parameter that arrives with steps already taken elsewhere: the deformation model's optimiser runs on
        // from the node pre-training stage, Trainer.adopt_deform_state)
        if (sg.t_origin[s] != sg.t_origin[s]) return fail(-1, "dgs_adam_step_origin: step origin is NaN");
        if (sg.sched_steps[s] > 0.0f && !(lrs[s] > 0.0f && sg.lr_final[s] > 0.0f))
            return fail(-1, "dgs_adam_step_sched: a scheduled segment needs positive initial and final rates");
    }
    sg.sched_t0 = sched_t0;
    sg.gscale = grad_scale;
    sg.zero_grad = zero_grad ? 1 : 0;
    sg.off[nseg] = offsets[nseg];
    const long long nb = adam_blocks(nseg, offsets);
    if (nb == 0) return 0;
    static const int nt = getenv("DGS_ADAM_NT") ? atoi(getenv("DGS_ADAM_NT")) : 1;   // default 1: -0.9 % per step (0.760 against 0.767 ms, three pairs); 2: noise
    // grid cap: 16 workgroups per CU (two rounds of resident workgroups) walk the plan -- the surfel update beside the node-MLP
    // backward chain: 0.754 against 0.763 ms per step (five pairs; 1024: +-0, 2048: 0.756, 6144 / 8192: +-0); DGS_ADAM_WGS=0: uncapped
    static const long long cap = getenv("DGS_ADAM_WGS") ? atoll(getenv("DGS_ADAM_WGS")) : 4096;
    const unsigned grid = (unsigned)(cap > 0 && cap < nb ? cap : nb);
    if (nt == 2)
        hipLaunchKernelGGL(adam_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sg, (const int2*)plan, (int)nb, grad, exp_avg,
                           exp_avg_sq, step_count, beta1, beta2, eps, skip, grad2);
    else if (nt == 1)
        hipLaunchKernelGGL(adam_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sg, (const int2*)plan, (int)nb, grad, exp_avg,
                           exp_avg_sq, step_count, beta1, beta2, eps, skip, grad2);
    else
        hipLaunchKernelGGL(adam_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sg, (const int2*)plan, (int)nb, grad, exp_avg,
                           exp_avg_sq, step_count, beta1, beta2, eps, skip, grad2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("adam_kernel: ") + hipGetErrorString(e));
    return 0;
}

// ---- control-node MLP (node_mlp.h) ---------------------------------------------------------------------------------
size_t dgs_mlp_packed_floats(void) { return mlp::kPackedFloats; }
size_t dgs_mlp_saved_floats(int M) { return mlp::sv_total(M); }
size_t dgs_mlp_scratch_floats(int M) { return mlp::sc_total(M); }

// params / grads: host arrays of 28 device pointers: (W, b) of T1, T2, L0..L7, local_rotation, warp, rotation, scaling
static const int kHeadRows[4] = {4, 3, 4, 2};

int dgs_mlp_forward(int M, const float* x, int x_stride, const float* t, int t_stride, const float* const* params,
                    const float* rot_bias, float* packed, float* saved, float* attrs, void* stream)
{
    return dgs_mlp_forward_select(M, x, x_stride, t, t_stride, params, rot_bias, packed, saved, attrs, nullptr, 0, 0, nullptr, nullptr, 1, 0,
                                  nullptr, stream);
}

int dgs_mlp_forward_select(int M, const float* x, int x_stride, const float* t, int t_stride, const float* const* params,
                           const float* rot_bias, float* packed, float* saved, float* attrs, const float* table, int nrows, int row_floats,
                           int* counter, int* override_, int stride, int offset, float* row_out, void* stream)
{
    if (M <= 0 || M % 64) return fail(-1, "dgs_mlp_forward: M must be a positive multiple of 64");
    if (table && (!counter || !override_ || !row_out || nrows <= 0 || row_floats <= 0 || stride <= 0 || offset < 0))
        return fail(-1, "dgs_mlp_forward_select: bad view table arguments");
    if (!x || !t || !params || !packed || !saved || !attrs) return fail(-1, "dgs_mlp_forward: null pointer");
    mlp::Weights w{};
    for (int l = 0; l < 10; l++) { w.W[l] = params[2 * l]; w.b[l] = params[2 * l + 1]; }
    int r = 0;
    for (int h = 0; h < 4; h++)
        for (int k = 0; k < kHeadRows[h]; k++, r++) { w.hw[r] = params[20 + 2 * h] + (size_t)k * mlp::kW; w.hb[r] = params[21 + 2 * h] + k; }
    for (; r < 16; r++) { w.hw[r] = w.hw[0]; w.hb[r] = w.hb[0]; }
    hipStream_t s = (hipStream_t)stream;
    int nthreads = mlp::kFwdVecs + mlp::kBwdVecs;   // one thread per float4 of the two operand arrays
    const mlp::SelectArgs sel{table, nrows, row_floats, counter, override_, stride, offset, row_out};
    hipLaunchKernelGGL(mlp::mlp_pack_kernel, dim3((nthreads + 255) / 256 + (table ? 1 : 0)), dim3(256), 0, s, w, reinterpret_cast<float4*>(packed), sel);
    mlp::FwdArgs a{};
    a.M = M; a.x = x; a.x_stride = x_stride; a.t = t; a.t_stride = t_stride;
    a.wp = reinterpret_cast<const float4*>(packed);
    a.bias = packed + mlp::kBiasOff;
    a.saved = saved; a.attrs = attrs;
    for (int i = 0; i < 4; i++) a.rot_bias[i] = rot_bias ? rot_bias[i] : 0.f;
    hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, s, a);
    if (hipGetLastError() != hipSuccess) return fail(-2, "dgs_mlp_forward: launch failed");
    return 0;
}

int dgs_mlp_backward(int M, const float* g_attrs, const float* packed, const float* saved, float* scratch, float* const* grads,
                     int accumulate, void* stream)
{
    return dgs_mlp_backward_reduce(M, const_cast<float*>(g_attrs), packed, saved, scratch, grads, accumulate, 0, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, 0, nullptr, stream);
}

int dgs_mlp_backward_reduce(int M, float* g_attrs, const float* packed, const float* saved, float* scratch, float* const* grads,
                            int accumulate, int H, const float* node_radius_raw, const float* node_weight_raw, float* g_nodes,
                            float* g_radius_raw, float* g_weight_raw, int reduce_flags, void* lbs_table, void* stream)
{
    if (M <= 0 || M % 64) return fail(-1, "dgs_mlp_backward: M must be a positive multiple of 64");
    if (!g_attrs || !packed || !saved || !scratch || !grads) return fail(-1, "dgs_mlp_backward: null pointer");
    if (lbs_table && (H < 0 || H > kLbsHmax || !node_radius_raw || !node_weight_raw || !g_nodes || !g_radius_raw || !g_weight_raw))
        return fail(-1, "dgs_mlp_backward_reduce: bad argument");
    static_assert(kLbsAttr == mlp::kHeads, "the node table's attribute columns are the MLP's outputs");
    hipStream_t s = (hipStream_t)stream;
    mlp::BwdArgs b{};
    b.M = M; b.g_attrs = g_attrs; b.saved = saved; b.scratch = scratch;
    if (lbs_table)   // flags as dgs_deform_reduce: bit 0 add to the gradients, bit 2 leave the table zeroed
        b.fold = mlp::ReduceFold{(float*)lbs_table, kLbsAttr + H + 2, H, node_radius_raw, node_weight_raw, g_nodes, g_radius_raw, g_weight_raw,
                                 g_attrs, reduce_flags & 1, (reduce_flags & 4) ? 1 : 0, (reduce_flags & 16) ? 1 : 0};
    b.wq = reinterpret_cast<const float4*>(packed) + (size_t)mlp::kFwdVecs;
    hipLaunchKernelGGL(mlp::mlp_bwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, s, b);

    mlp::WgArgs g{};
    g.M = M; g.accumulate = accumulate;
    int r = 0;
    for (int h = 0; h < 4; h++)
        for (int k = 0; k < kHeadRows[h]; k++, r++) { g.hw[r] = grads[20 + 2 * h] + (size_t)k * mlp::kW; g.hb[r] = grads[21 + 2 * h] + k; }
    for (; r < 16; r++) { g.hw[r] = g.hw[0]; g.hb[r] = g.hb[0]; }
    int nd = 0, block = 0;
    auto add = [&](const float* dz, int dzs, int out, const float* x, int xs, int in, float* dw, int dws, float* db) {
        mlp::WgDesc& d = g.d[nd++];
        d.dz = dz; d.dz_stride = dzs; d.out = out; d.x = x; d.x_stride = xs; d.in = in; d.dw = dw; d.dw_stride = dws; d.db = db;
        d.iblocks = (in + mlp::kWgTileI - 1) / mlp::kWgTileI;
        d.ntiles = ((out + mlp::kWgTileJ - 1) / mlp::kWgTileJ) * d.iblocks;
    };
    const int W = mlp::kW;
    auto Hs = [&](int l) { return saved + mlp::sv_h(M, l); };
    auto dZ = [&](int l) { return scratch + mlp::sc_dz(M, l); };
    add(g_attrs, mlp::kHeads, mlp::kHeads, Hs(7), W, W, nullptr, W, nullptr);                          // heads
    for (int l = 7; l >= 1; l--) {
        float* gw = grads[2 * (l + 2)];
        float* gb = grads[2 * (l + 2) + 1];
        if (l == 5) {
            add(dZ(5), W, W, saved + mlp::sv_inp(M), mlp::kInPad, mlp::kIn, gw, mlp::kIn + W, gb);    // [inp | .]
            add(dZ(5), W, W, Hs(4), W, W, gw + mlp::kIn, mlp::kIn + W, nullptr);                       // [. | H4]
        } else {
            add(dZ(l), W, W, Hs(l - 1), W, W, gw, W, gb);
        }
    }
    add(dZ(0), W, W, saved + mlp::sv_inp(M), mlp::kInPad, mlp::kIn, grads[4], mlp::kIn, grads[5]);   // L0
    add(scratch + mlp::sc_dt2(M), 32, mlp::kTOut, saved + mlp::sv_t1(M), W, W, grads[2], W, grads[3]);        // time net 2
    add(scratch + mlp::sc_dt1(M), W, W, saved + mlp::sv_et(M), mlp::kTPad, mlp::kTCh, grads[0], mlp::kTCh, grads[1]);  // time net 1
    g.ndesc = nd;
    block = mlp::wg_place(g);
    hipLaunchKernelGGL(mlp::mlp_wgrad_kernel, dim3(block), dim3(mlp::kWgThreads), 0, s, g);
    if (hipGetLastError() != hipSuccess) return fail(-2, "dgs_mlp_backward: launch failed");
    return 0;
}

int dgs_knn_points(int N, int M, int D, int K, const float* x, const float* nodes, long long* idx, float* dist2,
                   void* stream)
{
    if (N < 0 || M <= 0 || D < 1 || D > kKnnDpad || K < 1 || K > 4 || K > M) return fail(-1, "dgs_knn_points: bad argument");
    if (N == 0) return 0;
    if (!x || !nodes || !idx) return fail(-1, "dgs_knn_points: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return launch_knn<1>(N, M, D, x, nodes, idx, dist2, s);
    case 2: return launch_knn<2>(N, M, D, x, nodes, idx, dist2, s);
    case 3: return launch_knn<3>(N, M, D, x, nodes, idx, dist2, s);
    default: return launch_knn<4>(N, M, D, x, nodes, idx, dist2, s);
    }
}

// ---- deformation + activations in one pass (dgs_deform_*) ------------------------------------------------------------
int dgs_deform_forward(int N, int M, int H, const float* xyz, const float* feature, int feature_stride, const long long* idx,
                       const float* nodes, const float* node_radius_raw, const float* node_weight_raw, const float* attrs,
                       const float* mask, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                       float* means3D, float* scales, float* rotations, float* opacity, void* stream)
{
    if (int e = lbs_check(N, M, H)) return e;
    if (N == 0) return 0;
    if (!xyz || !feature || !idx || !nodes || !node_radius_raw || !node_weight_raw || !attrs || !scaling_raw || !rotation_raw ||
        !opacity_raw || !means3D || !scales || !rotations || !opacity)
        return fail(-1, "dgs_deform_forward: NULL pointer");
    LbsArgs a{N, M, H, feature_stride, xyz, feature, idx, nodes, attrs, mask, 3 + H, node_radius_raw, node_weight_raw};
    AsmArgs s{};
    s.scaling_raw = scaling_raw; s.rotation_raw = rotation_raw; s.opacity_raw = opacity_raw;
    s.means3D = means3D; s.scales = scales; s.rotations = rotations; s.opacity = opacity;
    hipLaunchKernelGGL(lbs_fwd_kernel<true>, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("lbs_fwd_kernel<asm>: ") + hipGetErrorString(e));
    return 0;
}

int dgs_deform_backward(int N, int M, int H, const float* xyz, const float* feature, int feature_stride, const long long* idx,
                        const float* nodes, const float* node_radius_raw, const float* node_weight_raw, const float* attrs,
                        const float* mask, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                        const float* g_means3D, const float* g_scales, const float* g_rotations, const float* g_opacity,
                        float* g_xyz, float* g_scaling_raw, float* g_rotation_raw, float* g_opacity_raw, float* g_feature,
                        float* g_nodes, float* g_radius_raw, float* g_weight_raw, float* g_attrs, int accumulate, void* scratch,
                        void* stream)
{
    if (int e = lbs_check(N, M, H)) return e;
    const int G = kLbsAttr + H + 2;
    const size_t lds = lbs_bwd_lds_bytes(M, H);
    if (lds > 160 * 1024 || M > 4 * kLbsBwdThreads) return fail(-2, "dgs_deform_backward: node tables do not fit the 160 KB of LDS");
    if (!scratch || !g_means3D || !g_scales || !g_rotations || !g_opacity || !g_xyz || !g_scaling_raw || !g_rotation_raw ||
        !g_opacity_raw || !g_feature || !g_nodes || !g_radius_raw || !g_weight_raw || !g_attrs)
        return fail(-1, "dgs_deform_backward: NULL pointer");
    LbsArgs a{N, M, H, feature_stride, xyz, feature, idx, nodes, attrs, mask, 3 + H, node_radius_raw, node_weight_raw};
    AsmArgs s{};
    s.scaling_raw = scaling_raw; s.rotation_raw = rotation_raw; s.opacity_raw = opacity_raw;
    s.g_means3D = g_means3D; s.g_scales = g_scales; s.g_rotations = g_rotations; s.g_opacity = g_opacity;
    s.g_xyz = g_xyz; s.g_scaling_raw = g_scaling_raw; s.g_rotation_raw = g_rotation_raw; s.g_opacity_raw = g_opacity_raw;
    if (accumulate & 2) {
        // coherent variant (surfels stored by nearest node): one zeroed [M][G] table, wave-level sums, global atomics
        // accumulate bit 2 (value 4): `scratch` is a persistent table that is zero on entry and must be zero on exit
        const bool persistent = (accumulate & 4) != 0;
        const bool fixed = (accumulate & 16) != 0;   // bit 4: 64-bit fixed-point table, integer atomics (order-free sums)
        if (!persistent) {
            const hipError_t me = hipMemsetAsync(scratch, 0, (size_t)M * G * (fixed ? sizeof(unsigned long long) : sizeof(float)), (hipStream_t)stream);
            if (me != hipSuccess) return fail(-4, std::string("dgs_deform_backward: ") + hipGetErrorString(me));
        }
        auto kern = fixed ? (H == 8 ? lbs_bwd_kernel<true, true, 8, true> : lbs_bwd_kernel<true, true, 0, true>)
                          : (H == 8 ? lbs_bwd_kernel<true, true, 8> : lbs_bwd_kernel<true, true, 0>);   // the trainer's hyper_dim, specialised
        hipLaunchKernelGGL(kern, dim3((N + kCohThreads - 1) / kCohThreads), dim3(kCohThreads), 0, (hipStream_t)stream, a,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, g_feature, feature_stride, accumulate & 1,
                           (float*)scratch, kCohThreads, s);
        if (!(accumulate & 8))      // bit 3: the caller reduces the table later (dgs_deform_reduce), e.g. on another stream
            hipLaunchKernelGGL(lbs_reduce_raw_kernel, dim3((M * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, M, H,
                               node_radius_raw, node_weight_raw, g_nodes, g_radius_raw, g_weight_raw, g_attrs, accumulate & 1, 1,
                               persistent ? (float*)scratch : (float*)nullptr, fixed ? 1 : 0);
    } else {
        if (accumulate & 16) return fail(-1, "dgs_deform_backward: the fixed-point table (bit 4) needs the coherent variant (bit 1)");
        if (accumulate & 8) return fail(-1, "dgs_deform_backward: the deferred reduce (bit 3) needs the coherent variant (bit 1)");
        const int chunk = (N + kLbsBlocks - 1) / kLbsBlocks;
        hipLaunchKernelGGL((lbs_bwd_kernel<true, false>), dim3(kLbsBlocks), dim3(kLbsBwdThreads), lds, (hipStream_t)stream, a, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, g_feature, feature_stride, accumulate & 1, (float*)scratch,
                           chunk > 0 ? chunk : 1, s);
        hipLaunchKernelGGL(lbs_reduce_raw_kernel, dim3((M * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, M, H,
                           node_radius_raw, node_weight_raw, g_nodes, g_radius_raw, g_weight_raw, g_attrs, accumulate & 1, kLbsBlocks,
                           (float*)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("lbs_bwd_kernel<asm>: ") + hipGetErrorString(e));
    return 0;
}

// ---- photometric loss: (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM) (train_gui.py:292-296) ---------------------
size_t dgs_photo_blocks(int C, int H, int W) { return (size_t)((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH) * (size_t)C; }
size_t dgs_regloss_blocks(int H, int W) { return (size_t)((W + 15) / 16) * ((H + 15) / 16); }

int dgs_photo_forward(int C, int H, int W, const float* img, const float* gt, float* partials, float* dm_dmu1, float* dm_dsigma1_sq,
                      float* dm_dsigma12, const float* const* gt_slot, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !partials || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12)
        return fail(-1, "dgs_photo_forward: bad argument");
    static const Gauss g = make_gauss();
    dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, C);
    const SsimFwdArgs A{H, W, img, gt, nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, nullptr, partials, gt_slot};
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("ssim_fwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_regloss_forward_partials_z(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                                   float lambda_normal, float lambda_dist, float* partials, const float* const* rays_slot,
                                   float* zero_plane, void* stream);

int dgs_regloss_forward_partials(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                                 float lambda_normal, float lambda_dist, float* partials, const float* const* rays_slot, void* stream)
{
    return dgs_regloss_forward_partials_z(H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, partials, rays_slot, nullptr, stream);
}

int dgs_regloss_forward_partials_z(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                                   float lambda_normal, float lambda_dist, float* partials, const float* const* rays_slot,
                                   float* zero_plane, void* stream)
{
    if (H <= 0 || W <= 0 || !allmap || (!rays_d && !rays_slot) || !rays_o || !wvt || !partials)
        return fail(-1, "dgs_regloss_forward_partials: bad argument");
    RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, rays_slot, 0, zero_plane};
    hipLaunchKernelGGL(regloss_fwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, (float*)nullptr,
                       partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("regloss_fwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

size_t dgs_regloss_fused_blocks(int H, int W) { return (size_t)((W + kRW - 1) / kRW) * ((H + kRH - 1) / kRH); }

int dgs_regloss_fused(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt, float lambda_normal,
                      float lambda_dist, float* partials, float* d_allmap, const float* const* rays_slot, void* stream)
{
    if (H <= 0 || W <= 0 || !allmap || (!rays_d && !rays_slot) || !rays_o || !wvt || !partials || !d_allmap)
        return fail(-1, "dgs_regloss_fused: bad argument");
    if ((long long)H * W * 8 >= (1ll << 32)) return fail(-2, "dgs_regloss_fused: image too large for 32-bit offsets");
    RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, rays_slot, 1, nullptr};
    hipLaunchKernelGGL(regloss_fused_kernel, dim3((W + kRW - 1) / kRW, (H + kRH - 1) / kRH), dim3(256), 0, (hipStream_t)stream, a, partials,
                       d_allmap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("regloss_fused_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_loss_forward_merged(int C, int H, int W, const float* img, const float* gt, float* photo_partials, float* dm_dmu1,
                            float* dm_dsigma1_sq, float* dm_dsigma12, const float* const* gt_slot, const float* allmap, const float* rays_d,
                            const float* rays_o, const float* wvt, float lambda_normal, float lambda_dist, float* reg_partials,
                            float* d_allmap, const float* const* rays_slot, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !photo_partials || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !allmap ||
        (!rays_d && !rays_slot) || !rays_o || !wvt || !reg_partials || !d_allmap)
        return fail(-1, "dgs_loss_forward_merged: bad argument");
    if ((long long)H * W * 8 >= (1ll << 32)) return fail(-2, "dgs_loss_forward_merged: image too large for 32-bit offsets");
    static const Gauss g = make_gauss();
    const int sgx = (W + kTW - 1) / kTW, sgy = (H + kTH - 1) / kTH, rgx = (W + kRW - 1) / kRW, rgy = (H + kRH - 1) / kRH;
    const long long total = (long long)sgx * sgy * C + (long long)rgx * rgy;
    if (total >= (1ll << 31)) return fail(-2, "dgs_loss_forward_merged: grid too large");
    const SsimFwdArgs A{H, W, img, gt, nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, nullptr, photo_partials, gt_slot};
    const RegArgs a{H, W, allmap, rays_d, rays_o, wvt, lambda_normal, lambda_dist, rays_slot, 1, nullptr};
    hipLaunchKernelGGL(loss_fwd_merged_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, A, g, sgx, sgy, C, a, reg_partials,
                       d_allmap, rgx, rgy);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("loss_fwd_merged_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_photo_backward_combine_guard(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                                     const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg,
                                     const float* const* gt_slot, const float* photo_partials, long long nphoto, const float* reg_partials,
                                     long long nreg, float* loss_out, const int* guard_skip, float* guard_step_count, float* guard_status,
                                     float* guard_ring, int guard_ring_len, void* stream)
{
    if (guard_step_count && (!loss_out || !guard_status || (guard_ring && guard_ring_len <= 0)))
        return fail(-1, "dgs_photo_backward_combine_guard: the guard needs loss_out, status and a ring length");
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !g_loss || !dL_dimg)
        return fail(-1, "dgs_photo_backward: bad argument");
    if (loss_out && (!photo_partials || !reg_partials || nphoto < 0 || nreg < 0)) return fail(-1, "dgs_photo_backward_combine: bad argument");
    static const Gauss g = make_gauss();
    dim3 grid((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, C);
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    CombineArgs c{photo_partials, (int)nphoto, reg_partials, (int)nreg, inv_n, lambda_dssim, loss_out,
                  guard_skip, guard_step_count, guard_status, guard_ring, guard_ring_len};
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, -lambda_dssim * inv_n,
                       (1.0f - lambda_dssim) * inv_n, img, gt, g, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, g_loss, dL_dimg, gt_slot, c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("ssim_bwd_kernel: ") + hipGetErrorString(e));
    return 0;
}

int dgs_photo_backward_combine(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                               const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg, const float* const* gt_slot,
                               const float* photo_partials, long long nphoto, const float* reg_partials, long long nreg, float* loss_out,
                               void* stream)
{
    return dgs_photo_backward_combine_guard(C, H, W, img, gt, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, lambda_dssim, g_loss, dL_dimg, gt_slot,
                                            photo_partials, nphoto, reg_partials, nreg, loss_out, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

int dgs_photo_backward(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                       const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg, const float* const* gt_slot,
                       void* stream)
{
    return dgs_photo_backward_combine(C, H, W, img, gt, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, lambda_dssim, g_loss, dL_dimg, gt_slot, nullptr, 0,
                                      nullptr, 0, nullptr, stream);
}

int dgs_loss_combine(const float* photo_partials, long long nphoto, const float* reg_partials, long long nreg, long long n,
                     float lambda_dssim, float* out, void* stream)
{
    if (!photo_partials || !reg_partials || !out || n <= 0 || nphoto < 0 || nreg < 0) return fail(-1, "dgs_loss_combine: bad argument");
    CombineArgs c{photo_partials, (int)nphoto, reg_partials, (int)nreg, 1.0f / (float)n, lambda_dssim, out, nullptr, nullptr, nullptr, nullptr, 0};
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "loss_combine_kernel: launch failed");
}

// ---- densification statistics ----------------------------------------------------------------------------------------
int dgs_densify_view(int P, const int* radii, const float* g_means2D, float* grad_norm, float* visible, int* radii_vis, void* stream)
{
    if (P < 0 || (P > 0 && (!radii || !g_means2D || !grad_norm || !visible || !radii_vis))) return fail(-1, "dgs_densify_view: bad argument");
    if (P == 0) return 0;
    hipLaunchKernelGGL(densify_view_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, radii, g_means2D, grad_norm,
                       visible, radii_vis);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "densify_view_kernel: launch failed");
}

int dgs_densify_accumulate_guarded(int P, const float* grad_norm, const float* visible, const int* radii_vis, float* accum, float* denom,
                                   int* max_radii, const int* skip, void* stream)
{
    if (P < 0 || (P > 0 && (!grad_norm || !visible || !radii_vis || !accum || !denom || !max_radii)))
        return fail(-1, "dgs_densify_accumulate: bad argument");
    if (P == 0) return 0;
    hipLaunchKernelGGL(densify_accum_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, grad_norm, visible, radii_vis,
                       accum, denom, max_radii, skip);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "densify_accum_kernel: launch failed");
}

int dgs_densify_accumulate(int P, const float* grad_norm, const float* visible, const int* radii_vis, float* accum, float* denom,
                           int* max_radii, void* stream)
{
    return dgs_densify_accumulate_guarded(P, grad_norm, visible, radii_vis, accum, denom, max_radii, nullptr, stream);
}

int dgs_knn_points2(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                    long long* idx, float* dist2, void* stream)
{
    const int D = D1 + D2;
    if (N < 0 || M <= 0 || D1 < 1 || D2 < 1 || D > kKnnDpad || K < 1 || K > 4 || K > M) return fail(-1, "dgs_knn_points2: bad argument");
    if (N == 0) return 0;
    if (!x1 || !x2 || !nodes || !idx) return fail(-1, "dgs_knn_points2: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
    case 1: return launch_knn<1>(N, M, D, x1, nodes, idx, dist2, s, x2, D1, x2_stride);
    case 2: return launch_knn<2>(N, M, D, x1, nodes, idx, dist2, s, x2, D1, x2_stride);
    case 3: return launch_knn<3>(N, M, D, x1, nodes, idx, dist2, s, x2, D1, x2_stride);
    default: return launch_knn<4>(N, M, D, x1, nodes, idx, dist2, s, x2, D1, x2_stride);
    }
}

int dgs_knn_refine_mode(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                        long long* idx, int mode, void* stream)
{
    const int D = D1 + D2;
    if (mode != 0 && mode != 1) return fail(-1, "dgs_knn_refine: mode must be 0 (3-D culling) or 1 (matrix cores)");
    if (N < 0 || M <= 0 || D1 < 3 || D2 < 0 || D > kKnnDpad || K < 1 || K > 4 || K > M) return fail(-1, "dgs_knn_refine: bad argument");
    if (M > 2048) return fail(-2, "dgs_knn_refine: more than 2048 nodes (use dgs_knn_points)");
    if (N == 0) return 0;
    if (!x1 || (D2 > 0 && !x2) || !nodes || !idx) return fail(-1, "dgs_knn_refine: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const float* xb = D2 > 0 ? x2 : nullptr;
    switch (K) {
    case 1: return launch_knn_refine<1>(N, M, D, x1, nodes, idx, s, xb, D1, x2_stride, mode == 1);
    case 2: return launch_knn_refine<2>(N, M, D, x1, nodes, idx, s, xb, D1, x2_stride, mode == 1);
    case 3: return launch_knn_refine<3>(N, M, D, x1, nodes, idx, s, xb, D1, x2_stride, mode == 1);
    default: return launch_knn_refine<4>(N, M, D, x1, nodes, idx, s, xb, D1, x2_stride, mode == 1);
    }
}

int dgs_knn_refine(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                   long long* idx, void* stream)
{
    return dgs_knn_refine_mode(N, M, D1, D2, K, x1, x2, x2_stride, nodes, idx, 0, stream);
}

}  // extern "C"
