// wave_reduce.h -- sum of 16 per-lane values over the 64 lanes of a gfx950 wave: in registers (permlane swaps + DPP; used by the
// skinning backward of train_ops.hip) and through the wave's own LDS (the backward blend).
// (The row-wise sums of the row-per-block backward are in ab/blend_bwd_rows.h.)
#pragma once
#include <hip/hip_runtime.h>

namespace dgs {

// ---- in registers --------------------------------------------------------------------------------------------------------
// v_permlane32_swap / v_permlane16_swap fold the wave halves and the row pairs (after them row r holds value i + 4 r in
// register i), then DPP adds with bank-masked writes fold a 16-lane row: row_mirror (lane l + lane 15-l -> lanes 0..7 keep
// registers 0,1, lanes 8..15 registers 2,3), row_half_mirror, two quad permutes.  Any pairing works for a sum; the mirrors
// are the ones DPP offers across 8 and 4 lanes.  On return every lane of quad k (lanes 4k .. 4k+3) holds the wave total of
// v[k].  36 VALU instructions, no ds_bpermute (measured on MI355X against the halving butterfly on the LDS crossbar --
// 17 ds_bpermute + 30 v_cndmask + 17 v_add -- inside the backward blend kernel: 0.338 -> 0.315 ms).
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

// ---- the sum through LDS ---------------------------------------------------------------------------------------------
// The DPP / permlane version above is 36 VALU instructions of the expensive classes (v_permlane*_swap 8.5 cycles, DPP adds
// 4.4-4.8) = ~170 of a backward visit's ~500 issue cycles, while the LDS pipe of the blend kernels is 22 % busy
// (SQ_LDS_IDX_ACTIVE, profiles/r03_pmc_metric.json).  Here the wave TRANSPOSES the 64 x 16 partials through its own 4 KB of
// LDS: every lane stores value k into row k with ds_write_addtid_b32 (address = M0 + offset + 4 * lane: no address VGPR,
// 2 LDS cycles per row), then lane l = 4 k + p reads a quarter of row k -- four ds_read_b128 -- and adds its 16 numbers (15 plain
// v_add_f32), and two quad DPP adds join the four quarters: 17 VALU instructions.  The quarter a lane reads in round i is
// rotated by (lane >> 3) & 3 so that the 16 lanes the LDS serves per cycle (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...)
// hit 16 different 16-byte columns of the 256-byte bank row: conflict free.  PH = 2 does it in two rounds of 8 values through
// 2 KB (lane l = 8 k + p reads an eighth of row k: 2 reads, 7 adds, 3 DPP adds per round).
// On return every lane of quad k holds the wave total of v[k] (PH = 1; same contract as wave_reduce16_dpp), or, for PH = 2,
// lanes 8 k .. 8 k + 3 hold v[k] and lanes 8 k + 4 .. 8 k + 7 hold v[k + 8] (reduce16_slot in kernels_blend.h).
typedef float red_f32x4 __attribute__((ext_vector_type(4)));

template <int PH>
struct RedLds {
    uint32_t m0;                  // LDS byte address of this wave's buffer (wave-uniform)
    const red_f32x4* rd[4 / PH];  // this lane's read addresses, one per round
    __device__ __forceinline__ void init(float* buf /* this wave's [16 / PH][64] floats */, int lane)
    {
        m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)buf);
        if (PH == 1) {
            const int k = lane >> 2, p = lane & 3, s = (lane >> 3) & 3;
#pragma unroll
            for (int i = 0; i < 4 / PH; i++) rd[i] = (const red_f32x4*)(buf + k * 64) + 4 * ((i + s) & 3) + p;
        } else {
            const int k = lane >> 3, p = lane & 7, s = (lane >> 4) & 1;
#pragma unroll
            for (int i = 0; i < 4 / PH; i++) rd[i] = (const red_f32x4*)(buf + k * 64) + 2 * p + ((i + s) & 1);
        }
    }
};

__device__ __forceinline__ float red_sum4(const red_f32x4& x) { return (x.x + x.y) + (x.z + x.w); }

__device__ __forceinline__ void red_store8(uint32_t m0, float a, float b, float c, float d, float e, float f, float g, float h)
{
    asm volatile(
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"                                   // SALU write of M0 -> LDS add-TID instruction: one wait state
        "ds_write_addtid_b32 %0 offset:0\n\t"
        "ds_write_addtid_b32 %1 offset:256\n\t"
        "ds_write_addtid_b32 %2 offset:512\n\t"
        "ds_write_addtid_b32 %3 offset:768\n\t"
        "ds_write_addtid_b32 %4 offset:1024\n\t"
        "ds_write_addtid_b32 %5 offset:1280\n\t"
        "ds_write_addtid_b32 %6 offset:1536\n\t"
        "ds_write_addtid_b32 %7 offset:1792"
        :
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h), "s"(m0)
        : "memory");
}

__device__ __forceinline__ void red_store8_hi(uint32_t m0, float a, float b, float c, float d, float e, float f, float g, float h)   // rows 8..15
{
    asm volatile(
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "ds_write_addtid_b32 %0 offset:2048\n\t"
        "ds_write_addtid_b32 %1 offset:2304\n\t"
        "ds_write_addtid_b32 %2 offset:2560\n\t"
        "ds_write_addtid_b32 %3 offset:2816\n\t"
        "ds_write_addtid_b32 %4 offset:3072\n\t"
        "ds_write_addtid_b32 %5 offset:3328\n\t"
        "ds_write_addtid_b32 %6 offset:3584\n\t"
        "ds_write_addtid_b32 %7 offset:3840"
        :
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h), "s"(m0)
        : "memory");
}

__device__ __forceinline__ float wave_reduce16_lds(float (&v)[16], const RedLds<1>& r)
{
    // (M0 is written and consumed inside each asm block; nothing else in these kernels uses it)
    red_store8(r.m0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    red_store8_hi(r.m0, v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
    const red_f32x4 x0 = *r.rd[0], x1 = *r.rd[1], x2 = *r.rd[2], x3 = *r.rd[3];
    float t = (red_sum4(x0) + red_sum4(x1)) + (red_sum4(x2) + red_sum4(x3));
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    return t;
}

__device__ __forceinline__ float red_row8(float t)   // sum over the 8 lanes 8 j .. 8 j + 7
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    return t;
}

__device__ __forceinline__ float wave_reduce16_lds(float (&v)[16], const RedLds<2>& r, int lane)
{
    red_store8(r.m0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const red_f32x4 x0 = *r.rd[0], x1 = *r.rd[1];
    red_store8(r.m0, v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);   // (LDS operations of a wave execute in order: the reads above see round 1)
    const red_f32x4 y0 = *r.rd[0], y1 = *r.rd[1];
    // lane 8 k + p holds an eighth of value k (round 1) and of value k + 8 (round 2).  The two 8-lane sums share their three exchange
    // steps (round 5; were three DPP adds each + one select): the first step crosses the halves of the group -- lanes p < 4 keep their
    // round-1 number and receive the round-1 number of lane 7 - p, lanes p >= 4 the same for round 2 -- and the two quad steps then add
    // like with like.  Two selects + three DPP adds; lanes 8 k .. 8 k + 3 end with the total of v[k], lanes 8 k + 4 .. 8 k + 7 with v[k + 8].
    const float lo = red_sum4(x0) + red_sum4(x1), hi = red_sum4(y0) + red_sum4(y1);
    const bool upper = (lane & 4) != 0;
    const float send = upper ? lo : hi;
    float keep = upper ? hi : lo;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(keep)
        : "v"(send));
    return keep;
}

}  // namespace dgs
