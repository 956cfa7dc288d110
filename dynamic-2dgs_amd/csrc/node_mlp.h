// node_mlp.h -- the control-node deformation MLP of the train step as four gfx950 kernels (included by train_ops.hip).
//
// Replaces, for the 1024 control nodes, DeformNetwork.forward + its autograd (utils/time_utils.py:311-453 of the
// reference: positional encodings, a 13->256->30 time net, 8 x 256 ReLU layers with a skip concat after layer 4, four
// linear heads) which in PyTorch is ~150 launches of 3-10 us on 1024 rows per step:
//
//   mlp_pack_kernel   re-lays the 0.52 M weights into MFMA fragment order (once per step; the weights change every
//                     Adam step), one copy for Z = X W^T and one for dX = dZ W.
//   mlp_fwd_kernel    one workgroup (16 waves) = 16 nodes through ALL layers; activations stay in LDS, weights stream from L2 as
//                     coalesced 1-KB fragments, v_mfma_f32_16x16x4_f32 (exact fp32).  Post-ReLU activations are saved.
//   mlp_bwd_kernel    the same tiling backwards: dZ_l = (dZ_{l+1} W_{l+1}) * [H_l > 0], every dZ_l saved.
//   mlp_wgrad_kernel  all weight/bias gradients in ONE launch: dW_l = dZ_l^T X_l as 64x32 tiles over a descriptor
//                     table, reduction over the 1024 nodes inside the workgroup (no atomics, deterministic).
//
// Fragment maps of v_mfma_f32_16x16x4_f32 (lane l): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[i = 4*(l>>4) + reg][j = l&15].  The reduction index inside one 16-wide K chunk is permuted (lane group q, step s
// <-> k = 4q + s) identically for A and B, so both operands are read as one 16-byte vector per lane and chunk.
#pragma once
#include <hip/hip_runtime.h>

namespace mlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 256;        // hidden width
constexpr int kIn = 93;        // 63 (posenc xyz, 10 bands) + 30 (time net)
constexpr int kInPad = 96;
constexpr int kXCh = 63;
constexpr int kTOut = 30;
constexpr int kTCh = 13;       // posenc t, 6 bands
constexpr int kTPad = 16;
constexpr int kHeads = 13;     // local_rotation 4 | d_xyz 3 | d_rotation 4 | d_scaling 2
constexpr int kRows = 16;      // nodes per workgroup
constexpr int kThreads = 1024; // 16 waves, one 16-column tile each per 256-wide layer (4 waves/SIMD hide the L2 latency of the weights)
constexpr int kNL = 11;        // layer ids: 0 = T1, 1 = T2, 2..9 = L0..L7, 10 = heads

// layer meta ------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int l_out(int l) { return l == 1 ? kTOut : (l == 10 ? kHeads : kW); }
__host__ __device__ constexpr int l_in(int l) { return l == 0 ? kTCh : (l == 2 ? kIn : (l == 7 ? kIn + kW : kW)); }
__host__ __device__ constexpr int l_inpad(int l) { return l == 0 ? kTPad : (l == 2 ? kInPad : (l == 7 ? kInPad + kW : kW)); }
__host__ __device__ constexpr int l_outpad(int l) { return l == 1 ? 32 : (l == 10 ? 16 : kW); }
// forward packing: tiles over out, chunks over in;  backward (dgrad) packing: tiles over in, chunks over out
__host__ __device__ constexpr int fwd_tiles(int l) { return l_outpad(l) / 16; }
__host__ __device__ constexpr int fwd_kc(int l) { return l_inpad(l) / 16; }
__host__ __device__ constexpr int bwd_tiles(int l) { return l == 0 ? 0 : l_inpad(l) / 16; }   // T1 needs no input gradient
__host__ __device__ constexpr int bwd_kc(int l) { return l_outpad(l) / 16; }
__host__ __device__ constexpr int fwd_off(int l)
{
    int f = 0;
    for (int i = 0; i < l; i++) f += fwd_tiles(i) * fwd_kc(i);
    return f;
}
__host__ __device__ constexpr int bwd_off(int l)
{
    int f = 0;
    for (int i = 0; i < l; i++) f += bwd_tiles(i) * bwd_kc(i);
    return f;
}
constexpr int kFwdChunks = fwd_off(kNL);   // 16-column x 16-k fragments (64 float4 each)
constexpr int kBwdChunks = bwd_off(kNL);
constexpr size_t kBiasOff = (size_t)(kFwdChunks + kBwdChunks) * 256;   // then kNL x 256 biases (zero padded)
constexpr size_t kPackedFloats = kBiasOff + (size_t)kNL * kW;

// padded input coordinate -> column of the weight matrix, or -1 (zero padding)
__device__ __forceinline__ int in_col(int l, int p)
{
    if (l == 7) return p < kInPad ? (p < kIn ? p : -1) : p - kInPad + kIn;
    return p < l_in(l) ? p : -1;
}

struct Weights {               // device pointers
    const float* W[10];        // T1, T2, L0..L7 (row-major [out][in])
    const float* b[10];
    const float* hw[16];       // head rows: 256 floats each (13 used)
    const float* hb[16];       // head bias elements
};
struct Grads {
    float* W[10];
    float* b[10];
    float* hw[16];
    float* hb[16];
};

// saved-activation buffer layout (floats)
__host__ __device__ inline size_t sv_inp(int M) { return 0; }                                   // [M][96]
__host__ __device__ inline size_t sv_et(int M) { return (size_t)M * kInPad; }                   // [M][16]
__host__ __device__ inline size_t sv_t1(int M) { return sv_et(M) + (size_t)M * kTPad; }         // [M][256]
__host__ __device__ inline size_t sv_h(int M, int l) { return sv_t1(M) + (size_t)M * kW * (1 + l); }  // L0..L7 outputs [M][256]
__host__ __device__ inline size_t sv_total(int M) { return sv_h(M, 8); }
// backward scratch layout (floats)
__host__ __device__ inline size_t sc_dz(int M, int l) { return (size_t)M * kW * l; }            // dZ of L0..L7
__host__ __device__ inline size_t sc_dt1(int M) { return (size_t)M * kW * 8; }
__host__ __device__ inline size_t sc_dt2(int M) { return sc_dt1(M) + (size_t)M * kW; }          // [M][32]
__host__ __device__ inline size_t sc_total(int M) { return sc_dt2(M) + (size_t)M * 32; }

// ---- packing ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ const float* w_row(const Weights& w, int l, int n)
{
    return l == 10 ? w.hw[n] : w.W[l] + (size_t)n * l_in(l);
}

__global__ void __launch_bounds__(256) mlp_pack_kernel(Weights w, float4* __restrict__ packed)
{
    int g = blockIdx.x * 256 + threadIdx.x;
    int frag = g >> 6, lane = g & 63;
    if (g < kNL * kW) {
        int l = g >> 8, c = g & 255;
        reinterpret_cast<float*>(packed)[kBiasOff + g] = c < l_out(l) ? (l == 10 ? w.hb[c][0] : w.b[l][c]) : 0.f;
    }
    if (frag >= kFwdChunks + kBwdChunks) return;
    bool bwd = frag >= kFwdChunks;
    int f = bwd ? frag - kFwdChunks : frag;
    int l = 0;
#pragma unroll
    for (int i = 1; i < kNL; i++)
        if (f >= (bwd ? bwd_off(i) : fwd_off(i))) l = i;
    int local = f - (bwd ? bwd_off(l) : fwd_off(l));
    int kcn = bwd ? bwd_kc(l) : fwd_kc(l);
    int t = local / kcn, kc = local - t * kcn;
    float v[4];
    if (!bwd) {
        int n = t * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            int col = in_col(l, kc * 16 + 4 * (lane >> 4) + s);
            v[s] = (n < l_out(l) && col >= 0) ? w_row(w, l, n)[col] : 0.f;
        }
    } else {
        int col = in_col(l, t * 16 + (lane & 15));
#pragma unroll
        for (int s = 0; s < 4; s++) {
            int j = kc * 16 + 4 * (lane >> 4) + s;
            v[s] = (j < l_out(l) && col >= 0) ? w_row(w, l, j)[col] : 0.f;
        }
    }
    packed[g] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- shared GEMM core -------------------------------------------------------------------------------------------
// acc[u] += act[16 x (16*kc_count)] * fragment stream of NT consecutive tiles.  `wp` already points at
// [first tile][first chunk][lane]; consecutive tiles are `tile_stride` float4 apart.
template <int NT>
__device__ __forceinline__ void mma_run(const float* __restrict__ act, int stride, int kc_count, const float4* __restrict__ wp,
                                        int tile_stride, f32x4 (&acc)[NT], int lane)
{
    const float* arow = act + (lane & 15) * stride + 4 * (lane >> 4);
#pragma unroll 4
    for (int kc = 0; kc < kc_count; kc++) {
        float4 a = *reinterpret_cast<const float4*>(arow + kc * 16);
        float4 b[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) b[u] = wp[(size_t)u * tile_stride + kc * 64];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[u].x, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[u].y, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[u].z, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[u].w, acc[u], 0, 0, 0);
        }
    }
}

constexpr int kSIn = 100, kSH = 260, kST = 20, kSG = 36;   // LDS row strides (floats; = 4 mod 32: conflict-free 16-B reads)

struct FwdArgs {
    int M;
    const float* x; int x_stride;      // node positions (first 3 columns used)
    const float* t; int t_stride;      // time per node (stride 0 = broadcast)
    const float4* wp;                  // packed forward fragments
    const float* bias;                 // packed biases [kNL][256]
    float* saved;                      // sv_* layout
    float* attrs;                      // [M][13]
    float rot_bias[4];                 // added to the local-rotation head
};

__device__ __forceinline__ float band(float v, int q)   // q = 2 * band + is_cos
{
    float a = v * (float)(1 << (q >> 1));
    return (q & 1) ? cosf(a) : sinf(a);
}

// one 256-wide hidden layer: one tile per wave, ReLU, result to LDS and to the saved activations
__device__ __forceinline__ void hidden_store(const f32x4 (&acc)[1], int wave, int lane, float* __restrict__ sdst,
                                             float* __restrict__ gdst /* [M][256] + row0*256 */)
{
    int col = wave * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int row = 4 * (lane >> 4) + r;
        float v = fmaxf(acc[0][r], 0.f);
        sdst[row * kSH + col] = v;
        gdst[row * kW + col] = v;
    }
}

__global__ void __launch_bounds__(kThreads) mlp_fwd_kernel(FwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float sIn[kRows * kSIn];
    __shared__ __attribute__((aligned(16))) float sA[kRows * kSH];
    __shared__ __attribute__((aligned(16))) float sB[kRows * kSH];
    __shared__ __attribute__((aligned(16))) float sT[kRows * kST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * kRows, M = a.M;

    for (int e = tid; e < kRows * kInPad; e += kThreads) {
        int r = e / kInPad, c = e - r * kInPad;
        float v = 0.f;
        if (c < kXCh) {
            const float* xr = a.x + (size_t)(row0 + r) * a.x_stride;
            v = c < 3 ? xr[c] : band(xr[(c - 3) % 3], (c - 3) / 3);
        }
        sIn[r * kSIn + c] = v;
    }
    for (int e = tid; e < kRows * kTPad; e += kThreads) {
        int r = e >> 4, c = e & 15;
        float tv = a.t[(size_t)(row0 + r) * a.t_stride];
        float v = c == 0 ? tv : (c < kTCh ? band(tv, c - 1) : 0.f);
        sT[r * kST + c] = v;
        a.saved[sv_et(M) + (size_t)(row0 + r) * kTPad + c] = v;
    }
    __syncthreads();

    f32x4 acc[1];
    auto init_bias = [&](int l, int tile) {
        float bv = a.bias[l * kW + tile * 16 + (lane & 15)];
        acc[0] = f32x4{bv, bv, bv, bv};
    };
    // time net layer 1: [16 x 16] -> 256, ReLU
    init_bias(0, wave);
    mma_run<1>(sT, kST, 1, a.wp + ((size_t)(fwd_off(0) + wave * fwd_kc(0)) * 64 + lane), 0, acc, lane);
    hidden_store(acc, wave, lane, sA, a.saved + sv_t1(M) + (size_t)row0 * kW);
    __syncthreads();
    // time net layer 2: 256 -> 30 (two tiles, waves 0 and 1), no activation; lands in columns 63..92 of the MLP input
    if (wave < 2) {
        f32x4 c1[1];
        int col = wave * 16 + (lane & 15);
        float bv = a.bias[1 * kW + col];
        c1[0] = f32x4{bv, bv, bv, bv};
        mma_run<1>(sA, kSH, fwd_kc(1), a.wp + ((size_t)(fwd_off(1) + wave * fwd_kc(1)) * 64 + lane), 0, c1, lane);
        if (col < kTOut)
#pragma unroll
            for (int r = 0; r < 4; r++) sIn[(4 * (lane >> 4) + r) * kSIn + kXCh + col] = c1[0][r];
    }
    __syncthreads();
    for (int e = tid; e < kRows * kInPad; e += kThreads) {
        int r = e / kInPad, c = e - r * kInPad;
        a.saved[sv_inp(M) + (size_t)(row0 + r) * kInPad + c] = sIn[r * kSIn + c];
    }
    // L0: 96 -> 256
    init_bias(2, wave);
    mma_run<1>(sIn, kSIn, fwd_kc(2), a.wp + ((size_t)(fwd_off(2) + wave * fwd_kc(2)) * 64 + lane), 0, acc, lane);
    hidden_store(acc, wave, lane, sB, a.saved + sv_h(M, 0) + (size_t)row0 * kW);
    __syncthreads();
    float* cur = sB;
    float* nxt = sA;
#pragma unroll 1
    for (int li = 1; li < 8; li++) {
        const int l = li + 2;
        init_bias(l, wave);
        if (li == 5) {   // input = [inp | H4]
            const float4* wp = a.wp + ((size_t)(fwd_off(7) + wave * fwd_kc(7)) * 64 + lane);
            mma_run<1>(sIn, kSIn, kInPad / 16, wp, 0, acc, lane);
            mma_run<1>(cur, kSH, kW / 16, wp + (kInPad / 16) * 64, 0, acc, lane);
        } else {
            mma_run<1>(cur, kSH, kW / 16, a.wp + ((size_t)(fwd_off(l) + wave * (kW / 16)) * 64 + lane), 0, acc, lane);
        }
        hidden_store(acc, wave, lane, nxt, a.saved + sv_h(M, li) + (size_t)row0 * kW);
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    // heads: 256 -> 13 (one tile, wave 0)
    if (wave == 0) {
        f32x4 c1[1];
        int col = lane & 15;
        float bv = a.bias[10 * kW + col];
        if (col < 4) bv += a.rot_bias[col];
        c1[0] = f32x4{bv, bv, bv, bv};
        mma_run<1>(cur, kSH, kW / 16, a.wp + ((size_t)fwd_off(10) * 64 + lane), 0, c1, lane);
        if (col < kHeads)
#pragma unroll
            for (int r = 0; r < 4; r++) a.attrs[(size_t)(row0 + 4 * (lane >> 4) + r) * kHeads + col] = c1[0][r];
    }
}

// ---- backward chain ---------------------------------------------------------------------------------------------
struct BwdArgs {
    int M;
    const float* g_attrs;      // [M][13]
    const float4* wq;          // packed dgrad fragments (= packed + kFwdChunks * 64)
    const float* saved;
    float* scratch;            // sc_* layout
};

// dH tile -> mask with the saved post-ReLU activation -> dZ to LDS and scratch
__device__ __forceinline__ void dz_store(const f32x4 (&acc)[1], int tile, int lane, const float* __restrict__ hsaved /* + row0*256 */,
                                         float* __restrict__ sdst, float* __restrict__ gdst /* + row0*256 */)
{
    int col = tile * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int row = 4 * (lane >> 4) + r;
        float v = hsaved[row * kW + col] > 0.f ? acc[0][r] : 0.f;
        sdst[row * kSH + col] = v;
        gdst[row * kW + col] = v;
    }
}

__global__ void __launch_bounds__(kThreads) mlp_bwd_kernel(BwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float sA[kRows * kSH];
    __shared__ __attribute__((aligned(16))) float sB[kRows * kSH];
    __shared__ __attribute__((aligned(16))) float sDin[kRows * kSIn];   // gradient of the MLP input, padded columns 48..95
    __shared__ __attribute__((aligned(16))) float sG[kRows * kSG];      // g_attrs (16 wide), later dT2 (32 wide)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * kRows, M = a.M;

    for (int e = tid; e < kRows * 16; e += kThreads) {
        int r = e >> 4, c = e & 15;
        sG[r * kSG + c] = c < kHeads ? a.g_attrs[(size_t)(row0 + r) * kHeads + c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[1];
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // heads: dH7 = g_attrs[16 x 16] * Wh
    acc[0] = zero;
    mma_run<1>(sG, kSG, 1, a.wq + ((size_t)(bwd_off(10) + wave * bwd_kc(10)) * 64 + lane), 0, acc, lane);
    dz_store(acc, wave, lane, a.saved + sv_h(M, 7) + (size_t)row0 * kW, sA, a.scratch + sc_dz(M, 7) + (size_t)row0 * kW);
    __syncthreads();
    float* cur = sA;
    float* nxt = sB;
#pragma unroll 1
    for (int li = 7; li >= 1; li--) {   // dZ_li (cur) -> dZ_{li-1} (nxt)
        const int l = li + 2;
        acc[0] = zero;
        if (li == 5) {   // tiles 0..5 belong to the MLP input (only 3..5 carry time-net gradient), 6..21 to H4
            const float4* wq = a.wq + ((size_t)bwd_off(7) * 64 + lane);
            mma_run<1>(cur, kSH, kW / 16, wq + (size_t)(6 + wave) * bwd_kc(7) * 64, 0, acc, lane);
            if (wave < 3) {
                f32x4 c1[1] = {zero};
                mma_run<1>(cur, kSH, kW / 16, wq + (size_t)(3 + wave) * bwd_kc(7) * 64, 0, c1, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) sDin[(4 * (lane >> 4) + r) * kSIn + (3 + wave) * 16 + (lane & 15)] = c1[0][r];
            }
        } else {
            mma_run<1>(cur, kSH, kW / 16, a.wq + ((size_t)(bwd_off(l) + wave * (kW / 16)) * 64 + lane), 0, acc, lane);
        }
        dz_store(acc, wave, lane, a.saved + sv_h(M, li - 1) + (size_t)row0 * kW, nxt,
                 a.scratch + sc_dz(M, li - 1) + (size_t)row0 * kW);
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    // L0: only the time-net columns (padded 48..95) of the input gradient are needed
    if (wave < 3) {
        f32x4 c1[1] = {zero};
        mma_run<1>(cur, kSH, kW / 16, a.wq + ((size_t)(bwd_off(2) + (3 + wave) * bwd_kc(2)) * 64 + lane), 0, c1, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) sDin[(4 * (lane >> 4) + r) * kSIn + (3 + wave) * 16 + (lane & 15)] += c1[0][r];
    }
    __syncthreads();
    // dT2 = input-gradient columns 63..92 -> [16 x 32]
    for (int e = tid; e < kRows * 32; e += kThreads) {
        int r = e >> 5, c = e & 31;
        float v = c < kTOut ? sDin[r * kSIn + kXCh + c] : 0.f;
        sG[r * kSG + c] = v;
        a.scratch[sc_dt2(M) + (size_t)(row0 + r) * 32 + c] = v;
    }
    __syncthreads();
    // dT1 = (dT2 * Wt2) * [T1 > 0]
    acc[0] = zero;
    mma_run<1>(sG, kSG, bwd_kc(1), a.wq + ((size_t)(bwd_off(1) + wave * bwd_kc(1)) * 64 + lane), 0, acc, lane);
    dz_store(acc, wave, lane, a.saved + sv_t1(M) + (size_t)row0 * kW, nxt, a.scratch + sc_dt1(M) + (size_t)row0 * kW);
}

// ---- weight gradients -------------------------------------------------------------------------------------------
struct WgDesc {
    const float* dz; int dz_stride; int out;
    const float* x; int x_stride; int in;     // x already offset to its first column
    float* dw; int dw_stride;                 // dw already offset to its first column; nullptr = per-row pointers (heads)
    float* db;                                // nullptr = no bias gradient from this descriptor
    int block0;                               // first workgroup of this descriptor
    int iblocks;                              // ceil(in / 32)
};
constexpr int kWgDescs = 12;
struct WgArgs {
    int M, accumulate, ndesc;
    WgDesc d[kWgDescs];
    float* hw[16];
    float* hb[16];
};

constexpr int kWgThreads = 1024;   // 16 waves: 4 output-feature tiles x 4 quarters of the node rows (latency hiding), LDS reduce

__global__ void __launch_bounds__(kWgThreads) mlp_wgrad_kernel(WgArgs a)
{
    __shared__ float sRed[3][4][64][9];   // partial sums of row quarters 1..3: 8 accumulators + bias per lane
    int di = 0;
#pragma unroll
    for (int i = 1; i < kWgDescs; i++)
        if (i < a.ndesc && (int)blockIdx.x >= a.d[i].block0) di = i;
    const WgDesc& d = a.d[di];
    const int local = blockIdx.x - d.block0;
    const int jb = local / d.iblocks, ib = local - jb * d.iblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jt = wave & 3, part = wave >> 2;
    const int jl = lane & 15, mq = lane >> 4;
    const int j = jb * 64 + jt * 16 + jl;            // this lane's A row (output feature)
    const int i0 = ib * 32 + jl;                     // this lane's B column (input feature) of tile 0; tile 1 = +16
    const bool jv = j < d.out, iv0 = i0 < d.in, iv1 = i0 + 16 < d.in;
    const int rows = a.M >> 2;                       // rows per quarter (M is a multiple of 64)
    const float* pz = d.dz + (size_t)(part * rows + 4 * mq) * d.dz_stride + (jv ? j : 0);
    const float* px = d.x + (size_t)(part * rows + 4 * mq) * d.x_stride + (iv0 ? i0 : 0);
    const int x1 = iv1 ? 16 : 0;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float bsum = 0.f;
#pragma unroll 2
    for (int m0 = 0; m0 < rows; m0 += 16) {
        float av[4], b0[4], b1[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            av[s] = pz[(size_t)(m0 + s) * d.dz_stride];
            b0[s] = px[(size_t)(m0 + s) * d.x_stride];
            b1[s] = px[(size_t)(m0 + s) * d.x_stride + x1];
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            float as = jv ? av[s] : 0.f;
            bsum += as;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(as, iv0 ? b0[s] : 0.f, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(as, iv1 ? b1[s] : 0.f, acc1, 0, 0, 0);
        }
    }
    if (part > 0) {
        float* r = sRed[part - 1][jt][lane];
#pragma unroll
        for (int c = 0; c < 4; c++) { r[c] = acc0[c]; r[4 + c] = acc1[c]; }
        r[8] = bsum;
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const float* r = sRed[p][jt][lane];
#pragma unroll
        for (int c = 0; c < 4; c++) { acc0[c] += r[c]; acc1[c] += r[4 + c]; }
        bsum += r[8];
    }
    // D[row = 4*mq + r -> output feature][col = jl -> input feature]
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int jo = jb * 64 + jt * 16 + 4 * mq + r;
        if (jo >= d.out) continue;
        float* row = d.dw ? d.dw + (size_t)jo * d.dw_stride : a.hw[jo];
        if (iv0) row[i0] = a.accumulate ? row[i0] + acc0[r] : acc0[r];
        if (iv1) row[i0 + 16] = a.accumulate ? row[i0 + 16] + acc1[r] : acc1[r];
    }
    if (ib == 0 && (d.db || !d.dw)) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (mq == 0 && jv) {
            float* p = d.dw ? d.db + j : a.hb[j];
            *p = a.accumulate ? *p + bsum : bsum;
        }
    }
}

}  // namespace mlp
