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

#include "../../include/dgs_train_ops.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

constexpr int kT = 16;          // output tile edge
constexpr int kR = 5;           // window radius (11 taps)
constexpr int kIn = kT + 2 * kR;  // 26

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

__global__ void __launch_bounds__(256) ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                       Gauss g, float* __restrict__ ssim_sum, float* __restrict__ dm_dmu1,
                                                       float* __restrict__ dm_ds11, float* __restrict__ dm_ds12)
{
    __shared__ float s_a[kIn][kIn + 1], s_b[kIn][kIn + 1];
    __shared__ float s_h[5][kIn][kT + 1];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < kIn * kIn; i += 256) {
        const int r = i / kIn, c = i - r * kIn;
        const int y = y0 + r - kR, x = x0 + c - kR;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        s_a[r][c] = in ? img1[plane + (size_t)y * W + x] : 0.f;
        s_b[r][c] = in ? img2[plane + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < kIn * kT; i += 256) {  // horizontal pass: 26 rows x 16 columns
        const int r = i / kT, c = i - r * kT;
        float m1 = 0.f, m2 = 0.f, q11 = 0.f, q22 = 0.f, q12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = s_a[r][c + k], b = s_b[r][c + k], w = g.w[k];
            m1 += w * a; m2 += w * b; q11 += w * a * a; q22 += w * b * b; q12 += w * a * b;
        }
        s_h[0][r][c] = m1; s_h[1][r][c] = m2; s_h[2][r][c] = q11; s_h[3][r][c] = q22; s_h[4][r][c] = q12;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        mu1 += w * s_h[0][ly + k][lx]; mu2 += w * s_h[1][ly + k][lx];
        s11 += w * s_h[2][ly + k][lx]; s22 += w * s_h[3][ly + k][lx]; s12 += w * s_h[4][ly + k][lx];
    }
    const int x = x0 + lx, y = y0 + ly;
    float val = 0.f;
    if (x < W && y < H) {
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sg1 = s11 - mu1_sq, sg2 = s22 - mu2_sq, sg12 = s12 - mu12;
        const float A = 2.f * mu12 + kC1, B = 2.f * sg12 + kC2, Cc = mu1_sq + mu2_sq + kC1, D = sg1 + sg2 + kC2;
        const float inv_cd = 1.0f / (Cc * D);
        val = A * B * inv_cd;
        if (dm_dmu1) {
            // map = A B / (Cc D) with sigma1^2 = s11 - mu1^2, sigma12 = s12 - mu1 mu2 (loss_utils.py:59-71)
            const size_t o = plane + (size_t)y * W + x;
            dm_dmu1[o] = (2.f * mu2 * B - 2.f * mu2 * A) * inv_cd - val * (2.f * mu1 / Cc - 2.f * mu1 / D);
            dm_ds11[o] = -val / D;
            dm_ds12[o] = 2.f * A * inv_cd;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(int H, int W, float inv_n, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, Gauss g, const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_ds11, const float* __restrict__ dm_ds12,
                                                       const float* __restrict__ dL_dmean, float* __restrict__ dL_dimg1)
{
    __shared__ float s_in[3][kIn][kIn + 1];
    __shared__ float s_h[3][kIn][kT + 1];
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = tid; i < kIn * kIn; i += 256) {
        const int r = i / kIn, c = i - r * kIn;
        const int y = y0 + r - kR, x = x0 + c - kR;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const size_t o = plane + (size_t)y * W + x;
        s_in[0][r][c] = in ? dm_dmu1[o] : 0.f;
        s_in[1][r][c] = in ? dm_ds11[o] : 0.f;
        s_in[2][r][c] = in ? dm_ds12[o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < kIn * kT; i += 256) {
        const int r = i / kT, c = i - r * kT;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = g.w[k];
            a += w * s_in[0][r][c + k]; b += w * s_in[1][r][c + k]; d += w * s_in[2][r][c + k];
        }
        s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = d;
    }
    __syncthreads();
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        a += w * s_h[0][ly + k][lx]; b += w * s_h[1][ly + k][lx]; d += w * s_h[2][ly + k][lx];
    }
    const int x = x0 + lx, y = y0 + ly;
    if (x < W && y < H) {
        const size_t o = plane + (size_t)y * W + x;
        // the zero-padded symmetric window is its own adjoint
        dL_dimg1[o] = (a + 2.f * img1[o] * b + img2[o] * d) * (inv_n * dL_dmean[0]);
    }
}

// ---- KNN ------------------------------------------------------------------------------------------------------
constexpr int kKnnChunk = 1024;  // nodes staged per pass
constexpr int kKnnDpad = 16;

// Q = number of float4 per (zero-padded) node row: D <= 4*Q.  Compile-time so that the distance loop fully
// unrolls and the wave-uniform node reads become ds_read_b128 broadcasts.
template <int K, int Q>
__global__ void __launch_bounds__(256) knn_kernel(int N, int M, int D, const float* __restrict__ x, const float* __restrict__ nodes,
                                                  long long* __restrict__ idx, float* __restrict__ dist2)
{
    __shared__ float4 s_nodes[kKnnChunk * Q];
    const int p = blockIdx.x * 256 + threadIdx.x;
    float xv[4 * Q];
#pragma unroll
    for (int d = 0; d < 4 * Q; d++) xv[d] = (p < N && d < D) ? x[(size_t)p * D + d] : 0.f;
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = INFINITY; bi[k] = 0; }
    for (int base = 0; base < M; base += kKnnChunk) {
        const int cnt = (M - base) < kKnnChunk ? (M - base) : kKnnChunk;
        __syncthreads();
        float* s_flat = reinterpret_cast<float*>(s_nodes);
        for (int i = threadIdx.x; i < cnt * 4 * Q; i += 256) {
            const int r = i / (4 * Q), d = i - r * (4 * Q);
            s_flat[i] = d < D ? nodes[(size_t)(base + r) * D + d] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; j++) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const float4 nd = s_nodes[j * Q + q];  // wave-uniform address: LDS broadcast
                float t;
                t = xv[4 * q + 0] - nd.x; acc += t * t;
                t = xv[4 * q + 1] - nd.y; acc += t * t;
                t = xv[4 * q + 2] - nd.z; acc += t * t;
                t = xv[4 * q + 3] - nd.w; acc += t * t;
            }
            // insertion into the sorted K best; strict < keeps the lower index on ties
            if (acc < bd[K - 1]) {
                bd[K - 1] = acc; bi[K - 1] = base + j;
#pragma unroll
                for (int k = K - 1; k > 0; k--) {
                    if (bd[k] < bd[k - 1]) {
                        const float td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                        const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
                    }
                }
            }
        }
    }
    if (p < N) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            idx[(size_t)p * K + k] = bi[k];
            if (dist2) dist2[(size_t)p * K + k] = bd[k];
        }
    }
}

template <int K, int Q>
int launch_knn_q(int N, int M, int D, const float* x, const float* nodes, long long* idx, float* dist2, hipStream_t s)
{
    hipLaunchKernelGGL((knn_kernel<K, Q>), dim3((N + 255) / 256), dim3(256), 0, s, N, M, D, x, nodes, idx, dist2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("knn_kernel: ") + hipGetErrorString(e));
    return 0;
}

template <int K>
int launch_knn(int N, int M, int D, const float* x, const float* nodes, long long* idx, float* dist2, hipStream_t s)
{
    switch ((D + 3) / 4) {
    case 1: return launch_knn_q<K, 1>(N, M, D, x, nodes, idx, dist2, s);
    case 2: return launch_knn_q<K, 2>(N, M, D, x, nodes, idx, dist2, s);
    case 3: return launch_knn_q<K, 3>(N, M, D, x, nodes, idx, dist2, s);
    default: return launch_knn_q<K, 4>(N, M, D, x, nodes, idx, dist2, s);
    }
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
    dim3 grid((W + kT - 1) / kT, (H + kT - 1) / kT, C);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, img1, img2, g, ssim_sum, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12);
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
    dim3 grid((W + kT - 1) / kT, (H + kT - 1) / kT, C);
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, inv_n, img1, img2, g, dm_dmu1, dm_dsigma1_sq,
                       dm_dsigma12, dL_dmean, dL_dimg1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-4, std::string("ssim_bwd_kernel: ") + hipGetErrorString(e));
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

}  // extern "C"
