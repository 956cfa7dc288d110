// ab/reduce_variants.h -- reductions 0-2 of the backward blend's 16 partials (butterfly on the LDS crossbar, matrix pipe, hybrid); variant 3 (DPP) is wave_reduce.h's wave_reduce16_dpp.  Select with -DDGS_AB_BUILD -DDGS_BWD_REDUCE=n.
// A/B material: measured, parity-green when it was measured, NOT part of the product library.  Included by kernels_blend.h only under
// -DDGS_AB_BUILD (tools/ab_variants.sh); the numbers that retired it are in the comments below and in DESIGN.md section 4.
// (no include guard / namespace of its own: textually included inside namespace dgs)

// DGS_BWD_REDUCE selects the implementation (A/B builds; the default is the measured best):
//   0  halving butterfly on the LDS crossbar: 17 ds_bpermute + 30 v_cndmask + 17 v_add (round-1 kernel),
//   1  matrix pipe: 16 x v_mfma_f32_16x16x4_f32 (exact fp32) with the partial as the A operand and a one-hot column
//      selector as B:  D[i][n] += sum_k v_n[lane 16 k + i]  -- one instruction folds the four 16-lane rows of value n into
//      column n of ONE 16x16 accumulator, so after the 16 instructions lane (q, n) holds four row sums of value n; three
//      adds and two cross-row exchanges finish.  The blend kernels issue no other MFMA, the pipe is otherwise idle,
//   2  hybrid: v_permlane32_swap folds the two wave halves first (values n and n + 8 share a register), then 8 MFMAs,
//   3  VALU only: permlane swaps across rows, bank-masked DPP adds inside a row (wave_reduce.h),
//   4  transposition through the wave's own LDS (wave_reduce.h: 16 ds_write_addtid_b32 + 4 ds_read_b128 + 17 VALU); DGS_RED_PHASES = 2
//      does it in two rounds of 8 values through half the LDS.
// Measured at 200k / 800x800 (blend bwd, ms): 0: 0.338, 1: 0.548 (the matrix pipe -- 16 x 32 cycles per visit -- becomes the
// bottleneck), 2: 0.455, 3: 0.315 (round 2; 0.304 on the round-4 kernel), 4 (default since round 4): 0.267 with two rounds, 48 staged
// entries per chunk and 5 workgroups per CU (30 KB of LDS); one round through 4 KB: 0.304 at 3 workgroups per CU (64 entries per chunk),
// 0.277 at 4 (48), 0.272 at 5 (32); two rounds at 4 workgroups per CU (64): 0.281 -- the occupancy decides, the chunk size does not
// (variant 3 with 48 entries: 0.304).  Where the 0.267 go (DGS_DIAG_BWD): atomics 0.009, reduction 0.056 (0.104 with variant 3),
// gradient arithmetic 0.10, alpha evaluation + loop + staging 0.104.
// On return lane l holds the wave total of v[l & 15] (variants 1, 2) / v[l >> 2] (variants 0, 3, 4 with one round); reduce16_slot() tells which.
__device__ __forceinline__ float wave_reduce16_butterfly(float (&v)[16], int lane)
{
    float a8[8], a4[4], a2[2], a1;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float keep = h5 ? v[i + 8] : v[i];
        const float send = h5 ? v[i] : v[i + 8];
        a8[i] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = h4 ? a8[i + 4] : a8[i];
        const float send = h4 ? a8[i] : a8[i + 4];
        a4[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float keep = h3 ? a4[i + 2] : a4[i];
        const float send = h3 ? a4[i] : a4[i + 2];
        a2[i] = keep + __shfl_xor(send, 8, 64);
    }
    {
        const float keep = h2 ? a2[1] : a2[0];
        const float send = h2 ? a2[0] : a2[1];
        a1 = keep + __shfl_xor(send, 4, 64);
    }
    a1 += __shfl_xor(a1, 2, 64);
    a1 += __shfl_xor(a1, 1, 64);
    return a1;
}

__device__ __forceinline__ float mfma_rows_finish(const f32x4_t& d0, const f32x4_t& d1)
{
    float p = ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    return p;
}

__device__ __forceinline__ float wave_reduce16_mfma(float (&v)[16], int lane)
{
    // two accumulators (even / odd columns) so that consecutive MFMAs do not wait for each other's result
    f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const int col = lane & 15;
#pragma unroll
    for (int n = 0; n < 16; n += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[n], col == n ? 1.f : 0.f, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[n + 1], col == n + 1 ? 1.f : 0.f, d1, 0, 0, 0);
    }
    return mfma_rows_finish(d0, d1);
}

__device__ __forceinline__ float wave_reduce16_hybrid(float (&v)[16], int lane)
{
    f32x4_t d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const int col = (lane & 15) - ((lane & 32) >> 2);   // lanes 32..63 carry value n + 8 in register n
#pragma unroll
    for (int n = 0; n < 8; n += 2) {
        float h[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            // after the swap x = (lower half of v[n], lower half of v[n+8]), y = (upper half of v[n], upper half of v[n+8])
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[n + u]), __float_as_uint(v[n + u + 8]), false, false);
            h[u] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h[0], col == n ? 1.f : 0.f, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h[1], col == n + 1 ? 1.f : 0.f, d1, 0, 0, 0);
    }
    return mfma_rows_finish(d0, d1);
}


struct BwdRedCtx {};

__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane, const BwdRedCtx&)
{
#if DGS_BWD_REDUCE == 3
    return wave_reduce16_dpp(v);   // wave_reduce.h: permlane swaps + bank-masked DPP adds, no LDS
#elif DGS_BWD_REDUCE == 0
    return wave_reduce16_butterfly(v, lane);
#elif DGS_BWD_REDUCE == 1
    return wave_reduce16_mfma(v, lane);
#else
    return wave_reduce16_hybrid(v, lane);
#endif
}

// which of the 16 values lane `lane` holds after wave_reduce16, or -1 if the lane holds a duplicate
__device__ __forceinline__ int reduce16_slot(int lane)
{
#if DGS_BWD_REDUCE == 0 || DGS_BWD_REDUCE == 3
    return (lane & 3) == 0 ? (lane >> 2) : -1;
#else
    return lane < 16 ? lane : -1;
#endif
}
