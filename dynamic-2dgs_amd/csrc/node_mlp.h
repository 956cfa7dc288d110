// node_mlp.h -- the control-node deformation MLP of the train step as four gfx950 kernels (included by train_ops.hip).
//
// Replaces, for the 1024 control nodes, DeformNetwork.forward + its autograd (utils/time_utils.py:311-453 of the
// reference: positional encodings, a 13->256->30 time net, 8 x 256 ReLU layers with a skip concat after layer 4, four
// linear heads) which in PyTorch is ~150 launches of 3-10 us on 1024 rows per step:
//
//   mlp_pack_kernel   re-lays the 0.52 M weights into the operand order of the two chains (once per step; the weights
//                     change every Adam step): [k/4][column] float4 for Z = X W^T and for dX = dZ W.
//   mlp_fwd_kernel    one workgroup (16 waves) = 8 nodes through ALL layers; activations stay in LDS, weights stream from L2
//                     as coalesced 1-KB rows.  Post-ReLU activations are saved.
//   mlp_bwd_kernel    the same tiling backwards: dZ_l = (dZ_{l+1} W_{l+1}) * [H_l > 0], every dZ_l saved.
//   mlp_wgrad_kernel  all weight/bias gradients in ONE launch: dW_l = dZ_l^T X_l as 64x32 tiles over a descriptor
//                     table, reduction over the 1024 nodes inside the workgroup (no atomics, deterministic).
//
// Why 8 nodes per workgroup and v_mfma_f32_4x4x1: a chain through all layers needs every weight (2.2 MB) in every
// workgroup, so a workgroup is bounded by its CU's 64 B/clk L2 port (~17 us) and by fp32 MFMA issue (256 flop/clk/CU
// whatever the tile shape).  With the 16-row tiles of v_mfma_f32_16x16x4 the 1024 nodes are 64 workgroups on a quarter of
// the CUs and MFMA issue is the bound (27 us + latencies: measured 54 us); v_mfma_f32_4x4x1_16b computes 16 independent
// 4x4 outer products per instruction, which with the A-block broadcast (cbsz = 4, abid = g) is D[4 rows of group g][64
// columns] += X[4g..4g+3][k] * W[k][64 columns]: 8 nodes per workgroup fill the instruction, 128 workgroups halve the
// MFMA time per CU, and both operand reads are vectors (A: one ds_read_b128 of 4 k for the 8 rows, B: one 16-byte load
// per lane = 4 k of its column).  Lane maps (tools/micro/mfma4x4_probe.hip checks them on the device):
// A[blk][i] lane 4 blk + i, B[blk][j] lane 4 blk + j, D[blk][i][j] VGPR i of lane 4 blk + j.
//
// A 256-wide layer is split over the 16 waves as 4 column groups of 64 x 4 quarters of K; the four partial sums of an
// output meet in LDS, where bias + ReLU (or the ReLU mask) are applied by all 1024 threads.  Narrow layers (30, 13 or 32
// outputs) use one column group x 16 slices of K.
#pragma once
#include <hip/hip_runtime.h>

#ifndef MLP_TRACE_POINT
#define MLP_TRACE_POINT() do { } while (0)   // tools/micro/mlp_chain_bench.hip stamps the device clock here
#endif

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
constexpr int kRows = 8;       // nodes per workgroup: two 4-row groups of the 4x4x1 MFMA
constexpr int kThreads = 1024; // 16 waves
constexpr int kNL = 11;        // layer ids: 0 = T1, 1 = T2, 2..9 = L0..L7, 10 = heads
constexpr int kXOff = 32;      // the MLP input in LDS is [time 30 | 2 zeros | posenc(xyz) 63 | 1 zero]: the time-net gradient is one column group

// layer meta (weights as PyTorch stores them, [out][in]) -------------------------------------------------------------
__host__ __device__ constexpr int l_out(int l) { return l == 1 ? kTOut : (l == 10 ? kHeads : kW); }
__host__ __device__ constexpr int l_in(int l) { return l == 0 ? kTCh : (l == 2 ? kIn : (l == 7 ? kIn + kW : kW)); }
// forward Z = X W^T: K = padded inputs, C = padded outputs;  backward dX = dZ W: K = padded outputs, C = the input columns needed
__host__ __device__ constexpr int f_K(int l) { return l == 0 ? kTPad : (l == 2 ? kInPad : (l == 7 ? kW + kInPad : kW)); }
__host__ __device__ constexpr int f_C(int l) { return (l == 1 || l == 10) ? 64 : kW; }
__host__ __device__ constexpr int b_K(int l) { return l == 1 ? 32 : (l == 10 ? 16 : kW); }
__host__ __device__ constexpr int b_C(int l) { return l == 0 ? 0 : (l == 2 ? 64 : (l == 7 ? kW + 64 : kW)); }   // L0: time columns only; L5: H4 + time
__host__ __device__ constexpr int f_off(int l)   // in float4
{
    int f = 0;
    for (int i = 0; i < l; i++) f += f_K(i) / 4 * f_C(i);
    return f;
}
__host__ __device__ constexpr int b_off(int l)
{
    int f = 0;
    for (int i = 0; i < l; i++) f += b_K(i) / 4 * b_C(i);
    return f;
}
constexpr int kFwdVecs = f_off(kNL);
constexpr int kBwdVecs = b_off(kNL);
constexpr size_t kBiasOff = (size_t)(kFwdVecs + kBwdVecs) * 4;   // then kNL x 256 biases (zero padded)
constexpr size_t kPackedFloats = kBiasOff + (size_t)kNL * kW;

// coordinate of a layer's input as the kernels hold it in LDS -> column of the weight matrix, or -1 (zero padding)
__device__ __forceinline__ int inp_col(int p)
{
    return p < kTOut ? kXCh + p : ((p < kXOff || p >= kXOff + kXCh) ? -1 : p - kXOff);
}
__device__ __forceinline__ int in_col(int l, int p)
{
    if (l == 2) return inp_col(p);
    if (l == 7) return p < kW ? kIn + p : inp_col(p - kW);   // LDS row = [H4 | input]; the reference concatenates [input, H4]
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
__host__ __device__ inline size_t sv_inp(int M) { return 0; }                                   // [M][96]: posenc(xyz) 63 | time 30 | 0
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

// The view of a replayed step (dgs_select_row): row_out <- table[v] with v = override[0] if it is >= 0 (then reset to -1), else
// (counter[0] * stride + offset) mod nrows; counter[0] += 1.  One workgroup.
struct SelectArgs {
    const float* table; int nrows, row_floats; int* counter; int* override_; int stride, offset; float* row_out;
};
__device__ __forceinline__ void select_row_body(const SelectArgs& q)
{
    const int ov = q.override_[0], c = q.counter[0];
    const int v = ov >= 0 ? (ov % q.nrows) : (int)(((long long)c * q.stride + q.offset) % q.nrows);
    for (int i = threadIdx.x; i < q.row_floats; i += blockDim.x) q.row_out[i] = q.table[(size_t)v * q.row_floats + i];
    __syncthreads();
    if (threadIdx.x == 0) { q.counter[0] = c + 1; q.override_[0] = -1; }
}

// one thread per float4 of the two operand arrays: forward [l][k/4][c] = W_l[c][col(4 (k/4) + s)], backward
// [l][j/4][c] = W_l[4 (j/4) + s][col(c)].  Rider (sel.table != null): one more workgroup at the end of the grid picks the view of
// this step -- the node MLP behind this kernel is the first consumer of the view's time, and as a node of its own in front of the
// step the selection cost its 4 us + the 5-6 us fork behind it.
__global__ void __launch_bounds__(256) mlp_pack_kernel(Weights w, float4* __restrict__ packed, SelectArgs sel)
{
    if (sel.table && blockIdx.x == gridDim.x - 1) { select_row_body(sel); return; }
    int g = blockIdx.x * 256 + threadIdx.x;
    if (g < kNL * kW) {
        int l = g >> 8, c = g & 255;
        reinterpret_cast<float*>(packed)[kBiasOff + g] = c < l_out(l) ? (l == 10 ? w.hb[c][0] : w.b[l][c]) : 0.f;
    }
    if (g >= kFwdVecs + kBwdVecs) return;
    bool bwd = g >= kFwdVecs;
    int f = bwd ? g - kFwdVecs : g;
    int l = 0;
#pragma unroll
    for (int i = 1; i < kNL; i++)
        if (f >= (bwd ? b_off(i) : f_off(i))) l = i;
    int local = f - (bwd ? b_off(l) : f_off(l));
    int C = bwd ? b_C(l) : f_C(l);
    int kk = local / C, c = local - kk * C;
    float v[4];
    if (!bwd) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            int col = in_col(l, 4 * kk + s);
            v[s] = (c < l_out(l) && col >= 0) ? w_row(w, l, c)[col] : 0.f;
        }
    } else {
        int col = in_col(l, c);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            int j = 4 * kk + s;
            v[s] = (j < l_out(l) && col >= 0) ? w_row(w, l, j)[col] : 0.f;
        }
    }
    packed[g] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- shared GEMM core -------------------------------------------------------------------------------------------
// A wave owns 64 output columns (lane = column) and a slice of K.  Its operand rows ([k/4][C] float4) arrive in CHUNKS of up to
// 8 rows through two register buffers: while one chunk multiplies, the next one -- of this layer or of the following one, the
// weights do not depend on the activations -- is in flight, across the barriers of the layer.  (All 16 rows of a layer in one
// request leave the load pipe idle while they multiply and the matrix pipe idle while the next 16 arrive: 3.3 us per layer
// against 1.9 us for 256 KB at the CU's 64 B/clk.)
constexpr int kChunk = 8;

// `p` is wave-uniform (scalar base registers: one address VGPR for all rows instead of a 64-bit pair per row)
template <int N>
__device__ __forceinline__ void frag_load(float4 (&b)[kChunk], const float4* __restrict__ p, int C, int lane)
{
#pragma unroll
    for (int kk = 0; kk < N; kk++) b[kk] = p[kk * C + lane];
}

// acc[g][v] += X[4 g + v][k] W[k][column] over the chunk's 4 N values of k; `arow` = this lane's activation row (lane & 7) at
// the chunk's first k
template <int N>
__device__ __forceinline__ void frag_mma(const float4 (&b)[kChunk], const float* __restrict__ arow, f32x4 (&acc)[2])
{
    // The two row groups alternate on the matrix pipe (a dependent 4x4x1 needs two wait states, the other group fills them).
    // The empty asm statements pin that order: the intrinsics are pure, and left alone the compiler runs one group's whole chain
    // first, keeps every activation vector alive for the second (spills; a scratch reload then waits for every weight load in
    // flight) and pays the wait states.  "memory" on the last one keeps the LDS read of step kk + 1 inside step kk.
#define MLP_PIN(mem) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : mem)
    float4 a = *reinterpret_cast<const float4*>(arow);
#pragma unroll
    for (int kk = 0; kk < N; kk++) {
        float4 an = a;
        if (kk + 1 < N) an = *reinterpret_cast<const float4*>(arow + 4 * (kk + 1));
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b[kk].x, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b[kk].x, acc[1], 4, 1, 0);
        MLP_PIN();
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b[kk].y, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b[kk].y, acc[1], 4, 1, 0);
        MLP_PIN();
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b[kk].z, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b[kk].z, acc[1], 4, 1, 0);
        MLP_PIN();
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b[kk].w, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b[kk].w, acc[1], 4, 1, 0);
        MLP_PIN("memory");
        a = an;
    }
#undef MLP_PIN
}

// partial sums of a wave -> LDS [wave][row = 4 g + v][lane]
__device__ __forceinline__ void part_store(float* __restrict__ sPart, int wave, int lane, const f32x4 (&acc)[2])
{
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int v = 0; v < 4; v++) sPart[(wave * 8 + 4 * g + v) * 64 + lane] = acc[g][v];
}
// wide layer (wave = 4 kq + cg): output (row, column c) = the four K quarters of column group c >> 6
__device__ __forceinline__ float part_sum_wide(const float* __restrict__ sPart, int row, int c)
{
    const float* p = sPart + ((c >> 6) * 8 + row) * 64 + (c & 63);
    return (p[0] + p[4 * 8 * 64]) + (p[8 * 8 * 64] + p[12 * 8 * 64]);
}
// narrow layer (wave = K slice): output (row, column ln < 64) = the 16 slices
__device__ __forceinline__ float part_sum_narrow(const float* __restrict__ sPart, int row, int ln)
{
    const float* p = sPart + row * 64 + ln;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) s += p[w * 8 * 64];
    return s;
}

constexpr int kSA = 356, kST = 20, kSG = 20, kSD = 36;   // LDS row strides (floats; = 4 or 20 mod 32: the 8 rows of a 16-B read hit 8 bank groups)
constexpr int kPartFloats = 16 * 8 * 64;

template <int L> struct FOff { static constexpr int v = f_off(L); };
template <int L> struct BOff { static constexpr int v = b_off(L); };

struct FwdArgs {
    int M;
    const float* x; int x_stride;      // node positions (first 3 columns used)
    const float* t; int t_stride;      // time per node (stride 0 = broadcast)
    const float4* wp;                  // packed forward operand
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

__global__ void __launch_bounds__(kThreads) mlp_fwd_kernel(FwdArgs a)
{
    // activation rows [hidden 256 | MLP input 96]: the skip layer reads one contiguous K = 352
    __shared__ __attribute__((aligned(16))) float sA[kRows * kSA];
    __shared__ __attribute__((aligned(16))) float sB[kRows * kSA];
    __shared__ __attribute__((aligned(16))) float sT[kRows * kST];
    __shared__ float sPart[kPartFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kq = wave >> 2, r8 = lane & 7;
    const int row0 = blockIdx.x * kRows, M = a.M;
    const int fc = tid & 255, fv = tid >> 8;                 // finalize: column and row-in-group of this thread
    // operand rows of this wave (wave-uniform pointers): a wide layer l is [K/4][256] and the wave owns rows kq * K/16 .. of column
    // group cg; a narrow one is [64][64] and the wave owns rows 4 wave ..
#define FW(l, row) (a.wp + (FOff<l>::v + (row) * kW + cg * 64))
#define FN(l) (a.wp + (FOff<l>::v + wave * 4 * 64))
    float4 q0[kChunk], q1[kChunk];
    frag_load<1>(q0, FW(0, kq), kW, lane);                   // T1
    frag_load<4>(q1, FN(1), 64, lane);                       // T2
    float bias = a.bias[0 * kW + fc];

    for (int e = tid; e < kRows * kInPad; e += kThreads) {
        int r = e / kInPad, p = e - r * kInPad;
        if (p < kTOut) continue;                             // time-net outputs: written by T2 below
        int q = p - kXOff;
        float v = 0.f;
        if (q >= 0 && q < kXCh) {
            const float* xr = a.x + (size_t)(row0 + r) * a.x_stride;
            v = q < 3 ? xr[q] : band(xr[(q - 3) % 3], (q - 3) / 3);
        }
        sA[r * kSA + kW + p] = v;
        sB[r * kSA + kW + p] = v;
        int sc = (q >= 0 && q < kXCh) ? q : (p < kXOff ? kIn + (p - kTOut) : kInPad - 1);   // the 3 zero columns 93..95
        a.saved[sv_inp(M) + (size_t)(row0 + r) * kInPad + sc] = v;
    }
    for (int e = tid; e < kRows * kTPad; e += kThreads) {
        int r = e >> 4, c = e & 15;
        float tv = a.t[(size_t)(row0 + r) * a.t_stride];
        float v = c == 0 ? tv : (c < kTCh ? band(tv, c - 1) : 0.f);
        sT[r * kST + c] = v;
        a.saved[sv_et(M) + (size_t)(row0 + r) * kTPad + c] = v;
    }
    MLP_TRACE_POINT();
    __syncthreads();
    MLP_TRACE_POINT();

    f32x4 acc[2];
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // partial sums -> bias + ReLU -> LDS activation rows and the saved activations; then the next layer's bias
    auto finish_hidden = [&](float* __restrict__ sdst, float* __restrict__ gdst /* [M][256] + row0 * 256 */, int next_bias) {
        part_store(sPart, wave, lane, acc);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 2; g++) {
            int row = 4 * g + fv;
            float v = fmaxf(part_sum_wide(sPart, row, fc) + bias, 0.f);
            sdst[row * kSA + fc] = v;
            gdst[row * kW + fc] = v;
        }
        bias = a.bias[next_bias];
        __syncthreads();
        MLP_TRACE_POINT();
        acc[0] = zero; acc[1] = zero;
    };
    const float* rA = sA + r8 * kSA;
    const float* rB = sB + r8 * kSA;

    // time net layer 1: [8 x 16] -> 256, ReLU
    acc[0] = zero; acc[1] = zero;
    frag_mma<1>(q0, sT + r8 * kST + kq * 4, acc);
    frag_load<6>(q0, FW(2, kq * 6), kW, lane);               // L0
    finish_hidden(sA, a.saved + sv_t1(M) + (size_t)row0 * kW, 1 * kW + (lane < kTOut ? lane : 0));

    // time net layer 2: 256 -> 30, no activation; lands in the first columns of the MLP input
    frag_mma<4>(q1, rA + wave * 16, acc);
    frag_load<8>(q1, FW(3, kq * 16), kW, lane);              // L1, first half
    part_store(sPart, wave, lane, acc);
    __syncthreads();
    if (tid < 512) {
        int row = tid >> 6;
        float v = lane < kTOut ? part_sum_narrow(sPart, row, lane) + bias : 0.f;
        if (lane < kXOff) {
            sA[row * kSA + kW + lane] = v;
            sB[row * kSA + kW + lane] = v;
        }
        if (lane < kTOut) a.saved[sv_inp(M) + (size_t)(row0 + row) * kInPad + kXCh + lane] = v;
    }
    bias = a.bias[2 * kW + fc];
    __syncthreads();
    MLP_TRACE_POINT();
    acc[0] = zero; acc[1] = zero;

    // L0: 96 -> 256
    frag_mma<6>(q0, rA + kW + kq * 24, acc);
    frag_load<8>(q0, FW(3, kq * 16 + 8), kW, lane);          // L1, second half
    finish_hidden(sB, a.saved + sv_h(M, 0) + (size_t)row0 * kW, 3 * kW + fc);

    // L1 .. L4, L6, L7: 256 -> 256.  `src` = this lane's row of the input, weights of layer id `l`; the two requests of the step
    // are the next two chunks of the stream
#define HIDDEN(src, dst, li, l, LOADA, LOADB)                                          \
    frag_mma<8>(q1, (src) + kq * 64, acc);                                             \
    LOADA;                                                                             \
    frag_mma<8>(q0, (src) + kq * 64 + 32, acc);                                        \
    LOADB;                                                                             \
    finish_hidden(dst, a.saved + sv_h(M, li) + (size_t)row0 * kW, ((l) + 1) * kW + fc)
    HIDDEN(rB, sA, 1, 3, frag_load<8>(q1, FW(4, kq * 16), kW, lane), frag_load<8>(q0, FW(4, kq * 16 + 8), kW, lane));
    HIDDEN(rA, sB, 2, 4, frag_load<8>(q1, FW(5, kq * 16), kW, lane), frag_load<8>(q0, FW(5, kq * 16 + 8), kW, lane));
    HIDDEN(rB, sA, 3, 5, frag_load<8>(q1, FW(6, kq * 16), kW, lane), frag_load<8>(q0, FW(6, kq * 16 + 8), kW, lane));
    // L4; then L5 = [H4 | MLP input] -> 256: operand rows 0..63 belong to H4, 64..87 to the input, each split over the 4 kq
    HIDDEN(rA, sB, 4, 6, frag_load<8>(q1, FW(7, kq * 16), kW, lane), frag_load<8>(q0, FW(7, kq * 16 + 8), kW, lane));
    frag_mma<8>(q1, rB + kq * 64, acc);
    frag_load<6>(q1, FW(7, 64 + kq * 6), kW, lane);
    frag_mma<8>(q0, rB + kq * 64 + 32, acc);
    frag_load<8>(q0, FW(8, kq * 16), kW, lane);              // L6, first half
    frag_mma<6>(q1, rB + kW + kq * 24, acc);
    frag_load<8>(q1, FW(8, kq * 16 + 8), kW, lane);
    finish_hidden(sA, a.saved + sv_h(M, 5) + (size_t)row0 * kW, 8 * kW + fc);
    // L6 (the buffers of the two halves are swapped from here on)
    frag_mma<8>(q0, rA + kq * 64, acc);
    frag_load<8>(q0, FW(9, kq * 16), kW, lane);
    frag_mma<8>(q1, rA + kq * 64 + 32, acc);
    frag_load<8>(q1, FW(9, kq * 16 + 8), kW, lane);
    finish_hidden(sB, a.saved + sv_h(M, 6) + (size_t)row0 * kW, 9 * kW + fc);
    // L7
    frag_mma<8>(q0, rB + kq * 64, acc);
    frag_load<4>(q0, FN(10), 64, lane);                      // heads
    frag_mma<8>(q1, rB + kq * 64 + 32, acc);
    finish_hidden(sA, a.saved + sv_h(M, 7) + (size_t)row0 * kW, 10 * kW + (lane < kHeads ? lane : 0));
#undef HIDDEN
    // heads: 256 -> 13
    frag_mma<4>(q0, rA + wave * 16, acc);
    part_store(sPart, wave, lane, acc);
    __syncthreads();
    if (tid < 512 && lane < kHeads) {
        int row = tid >> 6;
        float v = part_sum_narrow(sPart, row, lane) + bias;
        if (lane < 4) v += a.rot_bias[lane];
        a.attrs[(size_t)(row0 + row) * kHeads + lane] = v;
    }
    MLP_TRACE_POINT();
#undef FW
#undef FN
}

// ---- backward chain ---------------------------------------------------------------------------------------------
// Optional: the reduction of the skinning backward's node table (dgs_deform_reduce, lbs_reduce_raw_kernel) folded into the head of
// this kernel.  The table row of a node is [13 attribute gradients | H hyper-coordinate gradients | radius | weight]; a workgroup
// needs the attribute gradients of its own 8 nodes only, so it reads them from the table, finishes the other columns
// (exp / sigmoid chain rules of the raw radius and weight), leaves the row zeroed for the next backward and stores the
// attribute gradients for the weight-gradient kernel -- one launch (6 us in the step's tail, on its critical path) less.
struct ReduceFold {
    float* table;              // [M][G], or null: g_attrs is read as given
    int G, H;                  // G = 13 + H + 2
    const float* rad_raw; const float* w_raw;
    float* g_nodes;            // [M][3 + H]
    float* g_rad_raw; float* g_w_raw;
    float* g_attrs_out;        // [M][13]
    int accumulate, clear;
    int fixed;                 // the table holds 64-bit fixed-point sums (units of 2^-44: dgs_deform_backward accumulate bit 4)
};

struct BwdArgs {
    int M;
    const float* g_attrs;      // [M][13]
    const float4* wq;          // packed dgrad operand (= packed + kFwdVecs)
    const float* saved;
    float* scratch;            // sc_* layout
    ReduceFold fold;
};

__global__ void __launch_bounds__(kThreads) mlp_bwd_kernel(BwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float sA[kRows * kSA];
    __shared__ __attribute__((aligned(16))) float sB[kRows * kSA];
    __shared__ __attribute__((aligned(16))) float sG[kRows * kSG];      // g_attrs (16 wide)
    __shared__ __attribute__((aligned(16))) float sD[kRows * kSD];      // gradient of the time-net output (32 wide)
    __shared__ float sPart[kPartFloats];
    __shared__ float sPartN[kPartFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kq = wave >> 2, r8 = lane & 7;
    const int row0 = blockIdx.x * kRows, M = a.M;
    const int fc = tid & 255, fv = tid >> 8;
    // dgrad operand of layer l: [K/4][C] with C = b_C(l); wide: rows kq * K/16 .. of column group cg; narrow: rows 4 wave .. of one group
#define BW(l, row) (a.wq + (BOff<l>::v + (row) * b_C(l) + cg * 64))
#define BN(l, col0) (a.wq + (BOff<l>::v + wave * 4 * b_C(l) + (col0)))
    float4 q0[kChunk], q1[kChunk];
    frag_load<1>(q0, BW(10, kq), kW, lane);                  // heads
    frag_load<8>(q1, BW(9, kq * 16), kW, lane);              // L7, first half
    // the ReLU masks are the saved post-activation values of this thread's two outputs (rows fv and 4 + fv, column fc)
    const float* hs = a.saved + (size_t)(row0 + fv) * kW + fc;
    float h0 = hs[sv_h(M, 7)], h1 = hs[sv_h(M, 7) + 4 * kW];
    if (a.fold.table) {
        const ReduceFold& f = a.fold;
        const int T = 3 + f.H;
        for (int e = tid; e < kRows * f.G; e += kThreads) {
            const int r = e / f.G, c = e - r * f.G, node = row0 + r;
            float acc;
            if (f.fixed) {
                unsigned long long* t64 = reinterpret_cast<unsigned long long*>(f.table) + (size_t)node * f.G + c;
                acc = (float)((double)(long long)*t64 * (1.0 / 17592186044416.0));
                if (f.clear) *t64 = 0ull;
            } else {
                acc = f.table[(size_t)node * f.G + c];
                if (f.clear) f.table[(size_t)node * f.G + c] = 0.f;
            }
            if (c < kHeads) {
                sG[r * kSG + c] = acc;
                f.g_attrs_out[(size_t)node * kHeads + c] = acc;
            } else if (c < kHeads + f.H) {
                float* d = f.g_nodes + (size_t)node * T + 3 + (c - kHeads);
                *d = f.accumulate ? *d + acc : acc;
            } else if (c == kHeads + f.H) {
                const float v = acc * expf(f.rad_raw[node]);
                f.g_rad_raw[node] = f.accumulate ? f.g_rad_raw[node] + v : v;
            } else {
                const float w = 1.0f / (1.0f + expf(-f.w_raw[node]));
                const float v = acc * w * (1.0f - w);
                f.g_w_raw[node] = f.accumulate ? f.g_w_raw[node] + v : v;
            }
            if (c < 3 && !f.accumulate) f.g_nodes[(size_t)node * T + c] = 0.f;   // node positions are detached in the reference
        }
        for (int e = tid; e < kRows * (16 - kHeads); e += kThreads) sG[(e / (16 - kHeads)) * kSG + kHeads + e % (16 - kHeads)] = 0.f;
    } else {
        for (int e = tid; e < kRows * 16; e += kThreads) {
            int r = e >> 4, c = e & 15;
            sG[r * kSG + c] = c < kHeads ? a.g_attrs[(size_t)(row0 + r) * kHeads + c] : 0.f;
        }
    }
    __syncthreads();

    f32x4 acc[2];
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // partial sums -> ReLU mask -> dZ to LDS (when the chain goes on) and to the scratch; then the mask of the next stage
    auto finish_dz = [&](float* __restrict__ sdst, float* __restrict__ gdst /* + row0 * 256 */, size_t next_mask) {
        part_store(sPart, wave, lane, acc);
        __syncthreads();
        float v0 = h0 > 0.f ? part_sum_wide(sPart, fv, fc) : 0.f;
        float v1 = h1 > 0.f ? part_sum_wide(sPart, 4 + fv, fc) : 0.f;
        if (sdst) {
            sdst[fv * kSA + fc] = v0;
            sdst[(4 + fv) * kSA + fc] = v1;
        }
        gdst[fv * kW + fc] = v0;
        gdst[(4 + fv) * kW + fc] = v1;
        h0 = hs[next_mask]; h1 = hs[next_mask + 4 * kW];
        __syncthreads();
        acc[0] = zero; acc[1] = zero;
    };
    const float* rA = sA + r8 * kSA;
    const float* rB = sB + r8 * kSA;

    // heads: dZ7 = (g_attrs[8 x 16] * Wh) * [H7 > 0]
    acc[0] = zero; acc[1] = zero;
    frag_mma<1>(q0, sG + r8 * kSG + kq * 4, acc);
    frag_load<8>(q0, BW(9, kq * 16 + 8), kW, lane);
    finish_dz(sA, a.scratch + sc_dz(M, 7) + (size_t)row0 * kW, sv_h(M, 6));

    // dZ_{li-1} = (dZ_li * W_li) * [H_{li-1} > 0], W_li = layer id li + 2
#define DGRAD(src, dst, li, LOADA, LOADB, QA, QB)                                      \
    frag_mma<8>(QA, (src) + kq * 64, acc);                                             \
    LOADA;                                                                             \
    frag_mma<8>(QB, (src) + kq * 64 + 32, acc);                                        \
    LOADB;                                                                             \
    finish_dz(dst, a.scratch + sc_dz(M, (li) - 1) + (size_t)row0 * kW, (li) >= 2 ? sv_h(M, (li) - 2) : sv_t1(M))
    DGRAD(rA, sB, 7, frag_load<8>(q1, BW(8, kq * 16), kW, lane), frag_load<8>(q0, BW(8, kq * 16 + 8), kW, lane), q1, q0);
    DGRAD(rB, sA, 6, frag_load<8>(q1, BW(7, kq * 16), b_C(7), lane), frag_load<8>(q0, BW(7, kq * 16 + 8), b_C(7), lane), q1, q0);
    // L5: dZ4 from the H4 columns, and the time-net columns of the MLP input (column group 4 of the operand, K in 16 slices)
    {
        frag_mma<8>(q1, rA + kq * 64, acc);
        frag_load<4>(q1, BN(7, kW), b_C(7), lane);
        frag_mma<8>(q0, rA + kq * 64 + 32, acc);
        frag_load<8>(q0, BW(6, kq * 16), kW, lane);          // L4, first half
        f32x4 accn[2] = {zero, zero};
        frag_mma<4>(q1, rA + wave * 16, accn);
        frag_load<8>(q1, BW(6, kq * 16 + 8), kW, lane);
        part_store(sPartN, wave, lane, accn);
        finish_dz(sB, a.scratch + sc_dz(M, 4) + (size_t)row0 * kW, sv_h(M, 3));
        // (sPartN is complete after the first barrier of finish_dz; sD is read two stages later)
        if (tid < 512 && lane < 32) sD[(tid >> 6) * kSD + lane] = part_sum_narrow(sPartN, tid >> 6, lane);
    }
    DGRAD(rB, sA, 4, frag_load<8>(q0, BW(5, kq * 16), kW, lane), frag_load<8>(q1, BW(5, kq * 16 + 8), kW, lane), q0, q1);
    DGRAD(rA, sB, 3, frag_load<8>(q0, BW(4, kq * 16), kW, lane), frag_load<8>(q1, BW(4, kq * 16 + 8), kW, lane), q0, q1);
    DGRAD(rB, sA, 2, frag_load<8>(q0, BW(3, kq * 16), kW, lane), frag_load<8>(q1, BW(3, kq * 16 + 8), kW, lane), q0, q1);
    DGRAD(rA, sB, 1, frag_load<4>(q0, BN(2, 0), 64, lane), frag_load<2>(q1, BW(1, kq * 2), kW, lane), q0, q1);
#undef DGRAD
    // L0: only the time-net columns of the input gradient are needed (the node positions get no gradient here)
    frag_mma<4>(q0, rB + wave * 16, acc);
    part_store(sPartN, wave, lane, acc);
    __syncthreads();
    if (tid < 512 && lane < 32) {   // dT2 [8 x 32] (columns 30, 31: zero operand columns)
        int row = tid >> 6;
        float v = sD[row * kSD + lane] + part_sum_narrow(sPartN, row, lane);
        sD[row * kSD + lane] = v;
        a.scratch[sc_dt2(M) + (size_t)(row0 + row) * 32 + lane] = v;
    }
    __syncthreads();
    // dT1 = (dT2 * Wt2) * [T1 > 0]
    acc[0] = zero; acc[1] = zero;
    frag_mma<2>(q1, sD + r8 * kSD + kq * 8, acc);
    finish_dz(nullptr, a.scratch + sc_dt1(M) + (size_t)row0 * kW, sv_t1(M));
#undef BW
#undef BN
}

// ---- weight gradients -------------------------------------------------------------------------------------------
// dW[j][i] = sum_m dZ[m][j] X[m][i] is a sum of M outer products, which is what v_mfma_f32_4x4x1 computes: with the A-block
// broadcast (cbsz = 4, abid = g) one instruction is dZ[m][4g..4g+3] (x) X[m][64 columns], and both operands are ROWS of the
// row-major activations -- lane l holds dZ[m][j0 + l] and X[m][i0 + l]: coalesced loads straight from global memory, no
// transposition, no LDS staging.  A workgroup owns a 16 x 64 tile of one layer's dW (4 accumulators of 4 registers); its 8 waves
// split the M nodes and meet in LDS (fixed order: deterministic, no atomics).  ~540 workgroups of 512 threads, up to four per
// CU: the loop is bound by the latency of the activation rows (written by other XCDs: they come from the memory side), and
// waves in flight are what hides it.
// (Round 1 used 16x16x4 tiles whose operands are columns of the activations: twelve 4-byte loads with 64-byte segments per
// 8 MFMAs, 44 us; the chip-wide MFMA time of these products is 8 us.)
struct WgDesc {
    const float* dz; int dz_stride; int out;
    const float* x; int x_stride; int in;     // x already offset to its first column
    float* dw; int dw_stride;                 // dw already offset to its first column; nullptr = per-row pointers (heads)
    float* db;                                // nullptr = no bias gradient from this descriptor
    int iblocks;                              // ceil(in / 64)
    int ntiles;                               // ceil(out / 16) * iblocks
    int xcd, first;                           // the descriptor's tiles are slots first .. first + ntiles - 1 of XCD `xcd` (wg_place)
};
constexpr int kWgDescs = 12;
struct WgArgs {
    int M, accumulate, ndesc;
    WgDesc d[kWgDescs];
    float* hw[16];
    float* hb[16];
};

// XCD placement.  Workgroups go to the 8 XCDs round-robin (blockIdx & 7), each XCD has its own 4 MB L2, and the activations of
// all layers (19 MB) do not fit one: with tiles handed out in descriptor order every XCD streamed every layer over the fabric
// (168 MB per launch at ~6.7 TB/s = the 25 us the kernel took whatever its inner loop looked like).  A layer's dZ and X are 2 MB
// and its 64 tiles reuse them 16x / 4x: all tiles of a descriptor go to ONE XCD (slot = blockIdx >> 3), descriptors are spread
// over the XCDs longest-first.  Returns the grid size.
inline int wg_place(WgArgs& a)
{
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool done[kWgDescs] = {};
    for (int n = 0; n < a.ndesc; n++) {
        int best = -1;
        for (int i = 0; i < a.ndesc; i++)
            if (!done[i] && (best < 0 || a.d[i].ntiles > a.d[best].ntiles)) best = i;
        int x = 0;
        for (int k = 1; k < 8; k++)
            if (load[k] < load[x]) x = k;
        done[best] = true;
        a.d[best].xcd = x;
        a.d[best].first = load[x];
        load[x] += a.d[best].ntiles;
    }
    int most = 0;
    for (int k = 0; k < 8; k++) most = load[k] > most ? load[k] : most;
    return 8 * most;
}

constexpr int kWgWaves = 4;       // slices of the nodes
constexpr int kWgThreads = 64 * kWgWaves;
constexpr int kWgTileJ = 16, kWgTileI = 64;
constexpr int kWgG = kWgTileJ / 4;   // accumulators (4 rows of dW each) per wave
constexpr int kWgPhase = 32;         // rows per register set

// Three things bound this kernel before its matrix instructions do (measured one by one, tools/micro/mlp_chain_bench.hip):
//  * a 4-byte-per-lane load occupies the CU's address unit as long as a 16-byte one: 256 B per 16 clocks = a quarter of the
//    64 B/clk the L1 can take, and with 830 KB of operands per CU that alone was 25 us.  So every lane loads 16 bytes -- four
//    ROWS of the tile per instruction (lane l: row l / 16, columns 4 (l % 16) ..) -- and the wave turns them into the
//    lane = column layout of the MFMA operands through 1.3 KB of its own LDS (no barrier: nobody else touches it);
//  * the rows were written by other XCDs and come from the memory side (~2 us under load): two register sets of 32 rows, one
//    multiplies (128 MFMAs) while the 16 loads of the other are in flight;
//  * around a loop's back edge the compiler waits for EVERY load in flight (s_waitcnt vmcnt(0) at the loop header), which
//    turns a ring of small batches into one exposed latency per trip; with two sets the only wait is "all of the other set",
//    which is what vmcnt(0) means at that point anyway.
__global__ void __launch_bounds__(kWgThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) mlp_wgrad_kernel(WgArgs a)
{
    __shared__ float sRed[kWgWaves][kWgG][4][64];   // [wave][row group g][v][lane]: 16 KB
    __shared__ float sBias[kWgWaves][kWgTileJ];
    __shared__ __attribute__((aligned(16))) float sX[kWgWaves][2][4 * kWgTileI];   // per wave: 4 rows x 64 columns, double-buffered
    __shared__ __attribute__((aligned(16))) float sZ[kWgWaves][2][16 * kWgTileJ];              // per wave: 16 rows x 16 columns, double-buffered
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int di = -1;
#pragma unroll
    for (int i = 0; i < kWgDescs; i++)
        if (i < a.ndesc && a.d[i].xcd == xcd && slot >= a.d[i].first && slot < a.d[i].first + a.d[i].ntiles) di = i;
    if (di < 0) return;                              // this XCD has fewer tiles than the fullest one
    const WgDesc& d = a.d[di];
    const int local = slot - d.first;
    const int jb = local / d.iblocks, ib = local - jb * d.iblocks;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = jb * kWgTileJ + lane, i = ib * kWgTileI + lane;
    const bool jv = lane < kWgTileJ && j < d.out, iv = i < d.in;
    const int rows = a.M / kWgWaves;                 // nodes per wave (M is a multiple of 64: rows of 16)
    // buffer loads: resource = the matrix (reads past its end return 0: a 16-byte load of the last columns of a 13-wide row runs
    // into the next row, and past the buffer on the last one), scalar offset = the row, one 32-bit lane offset for the whole kernel
    const int zs = d.dz_stride * 4, xs = d.x_stride * 4;   // row strides in bytes
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dz), 0, a.M * zs, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, a.M * xs, 0x00020000);
    const int zb = wave * rows * zs, xb = wave * rows * xs;
    // lane -> (row, 4 columns) of one load: X 4 rows (row l / 16, columns 4 (l % 16)); dZ 16 rows (row l / 4, columns 4 (l % 4))
    const unsigned xoff = (unsigned)(lane >> 4) * xs + (unsigned)(ib * kWgTileI + 4 * (lane & 15)) * 4u;
    const unsigned zoff = (unsigned)(lane >> 2) * zs + (unsigned)(jb * kWgTileJ + 4 * (lane & 3)) * 4u;
    f32x4 acc[kWgG];
#pragma unroll
    for (int g = 0; g < kWgG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    constexpr int kGroups = kWgPhase / 4;
    float4 xq[2][kGroups], zq[2][kWgPhase / 16];   // (a third set, two phases in flight, spills at 168 registers: 40 us)
    auto request = [&](int s, int m0) {   // groups past the end: the last group again (never used)
#pragma unroll
        for (int g = 0; g < kGroups; g++) {
            const int m = m0 + 4 * g < rows ? m0 + 4 * g : rows - 4;
            xq[s][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, xb + m * xs, 0));
        }
#pragma unroll
        for (int h = 0; h < kWgPhase / 16; h++) {
            const int m = m0 + 16 * h < rows ? m0 + 16 * h : rows - 16;
            zq[s][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rz, zoff, zb + m * zs, 0));
        }
    };
    float* const wx = &sX[wave][0][(lane >> 4) * kWgTileI + 4 * (lane & 15)];
    float* const wz = &sZ[wave][0][(lane >> 2) * kWgTileJ + 4 * (lane & 3)];
    const float* const rxp = &sX[wave][0][lane];
    const float* const rzp = &sZ[wave][0][lane & 15];
    // one set of 32 rows = 8 groups of 4: group g + 2 is written to LDS and group g + 1 read back while group g multiplies (a
    // wave's LDS operations complete in order; without the overlap the write -> read -> multiply chain of a group is ~450 clocks)
    auto put = [&](int s, int g) {
        *reinterpret_cast<float4*>(wx + (g & 1) * 4 * kWgTileI) = xq[s][g];
        if ((g & 3) == 0) *reinterpret_cast<float4*>(wz + ((g >> 2) & 1) * 16 * kWgTileJ) = zq[s][g >> 2];
    };
    auto get = [&](int g, float (&av)[4], float (&bvv)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            av[r] = rzp[((g >> 2) & 1) * 16 * kWgTileJ + (4 * (g & 3) + r) * kWgTileJ];
            bvv[r] = rxp[(g & 1) * 4 * kWgTileI + r * kWgTileI];
        }
    };
    auto multiply = [&](int s, int m0) {
        float av[2][4], bvv[2][4];
        put(s, 0);
        put(s, 1);
        get(0, av[0], bvv[0]);
#pragma unroll
        for (int g = 0; g < kGroups; g++) {
            if (g + 1 < kGroups) get(g + 1, av[(g + 1) & 1], bvv[(g + 1) & 1]);
            const bool jvs = jv && m0 + 4 * g < rows;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float av_ = jvs ? av[g & 1][r] : 0.f, bv_ = iv ? bvv[g & 1][r] : 0.f;
                bsum += av_;
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av_, bv_, acc[0], 4, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av_, bv_, acc[1], 4, 1, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av_, bv_, acc[2], 4, 2, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av_, bv_, acc[3], 4, 3, 0);
                static_assert(kWgG == 4, "one MFMA per row group");
            }
            if (g + 2 < kGroups) put(s, g + 2);
        }
    };
    request(0, 0);
#pragma unroll 1
    for (int m0 = 0; m0 < rows; m0 += 2 * kWgPhase) {
        request(1, m0 + kWgPhase);
        multiply(0, m0);
        request(0, m0 + 2 * kWgPhase);
        multiply(1, m0 + kWgPhase);
    }
#pragma unroll
    for (int g = 0; g < kWgG; g++)
#pragma unroll
        for (int v = 0; v < 4; v++) sRed[wave][g][v][lane] = acc[g][v];
    if (lane < kWgTileJ) sBias[wave][lane] = bsum;
    __syncthreads();
    // thread -> (row group g = wave, column lane): 4 outputs dW[16 jb + 4 g + v][64 ib + lane]
    if (iv) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int jo = jb * kWgTileJ + 4 * wave + v;
            if (jo >= d.out) continue;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kWgWaves; w++) sum += sRed[w][wave][v][lane];
            float* row = d.dw ? d.dw + (size_t)jo * d.dw_stride : a.hw[jo];
            row[i] = a.accumulate ? row[i] + sum : sum;
        }
    }
    if (ib == 0 && (d.db || !d.dw) && threadIdx.x < kWgTileJ) {
        const int jo = jb * kWgTileJ + threadIdx.x;
        if (jo < d.out) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kWgWaves; w++) sum += sBias[w][threadIdx.x];
            float* p = d.dw ? d.db + jo : a.hb[jo];
            *p = a.accumulate ? *p + sum : sum;
        }
    }
}

}  // namespace mlp
