// wave_reduce.h -- sum of 16 per-lane values over the 64 lanes of a gfx950 wave without touching LDS.
//
// v_permlane32_swap / v_permlane16_swap fold the wave halves and the row pairs (after them row r holds value i + 4 r in
// register i), then DPP adds with bank-masked writes fold a 16-lane row: row_mirror (lane l + lane 15-l -> lanes 0..7 keep
// registers 0,1, lanes 8..15 registers 2,3), row_half_mirror, two quad permutes.  Any pairing works for a sum; the mirrors
// are the ones DPP offers across 8 and 4 lanes.  On return every lane of quad k (lanes 4k .. 4k+3) holds the wave total of
// v[k].  36 VALU instructions, no ds_bpermute (measured on MI355X against the halving butterfly on the LDS crossbar --
// 17 ds_bpermute + 30 v_cndmask + 17 v_add -- inside the backward blend kernel: 0.338 -> 0.315 ms).
#pragma once
#include <hip/hip_runtime.h>

namespace dgs {

__device__ __forceinline__ float wave_reduce16_dpp(float (&v)[16])
{
    float h[8], g[4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
        h[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[i]), __float_as_uint(h[i + 4]), false, false);
        g[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    float f0, f1, e;
    // s_nop: a DPP source written by the previous VALU instruction needs two wait states (the assembler does not insert them)
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %4, %4 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(f0), "=&v"(f1), "=&v"(e)
        : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
    return e;
}

// Eight values: on return the eight lanes 8k .. 8k+7 hold the wave total of v[k].
__device__ __forceinline__ float wave_reduce8_dpp(float (&v)[8])
{
    float h[4], g[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        h[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // lanes 0..31: value i, lanes 32..63: value i + 4
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[i]), __float_as_uint(h[i + 2]), false, false);
        g[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // row r: value i + 2 r
    }
    float f, e;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"       // lanes 0..7 of a row: value 2 r
        "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xc\n\t"       // lanes 8..15:         value 2 r + 1
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(f), "=&v"(e)
        : "v"(g[0]), "v"(g[1]));
    return e;
}

}  // namespace dgs
