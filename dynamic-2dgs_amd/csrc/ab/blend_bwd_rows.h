// ab/blend_bwd_rows.h -- the row-per-block backward blend (round 4): 2.6 x slower than blend_bwd_kernel on the L2 float-atomic rate.  Select with -DDGS_AB_BUILD -DDGS_BWD_ROWS=1.
// A/B material: measured, parity-green when it was measured, NOT part of the product library.  Included by kernels_blend.h only under
// -DDGS_AB_BUILD (tools/ab_variants.sh); the numbers that retired it are in the comments below and in DESIGN.md section 4.
// (no include guard / namespace of its own: textually included inside namespace dgs)

// ---- backward blend, one list per 16-lane row (round 4) -------------------------------------------------------------------
// In blend_bwd_kernel above all 64 lanes of a wave visit the same entry, and 30 of them blend it on the 200k / 800x800 scene
// (tools/blend_stats.py).  Here every DPP row of the wave -- one 4x4 pixel block of the quadrant (surfel_math.h lane_pixel) --
// walks the list of the entries that can reach ITS block: while staging, a lane tests its entry against the four blocks
// (blocks_hit_linear: the record's pixel box, then a tangent-plane bound of the footprint's conic; also what decides whether the
// entry is staged at all), the entry's planes are stored once per wave, compacted as before, and each block it reaches gets the
// slot number appended to its row's byte list (rank among the ballot of that block).  One wave-instruction of the visit loop then
// serves four (entry, block) pairs: lane l reads the planes of the slot its row is at (four addresses per ds_read_b128), rows that
// have run out of entries read the null slot (opacity 0: fails the alpha test, contributes exact zeros).  A row's list also stops
// at ITS pixels' last contributor instead of the wave's.  The 16 partials are summed per row (wave_reduce.h rows_reduce16) and all
// 64 lanes issue one atomic each: row r' = (l >> 1) & 3 of value (l >> 3) + 8 (l & 1), to the surfel of row r's entry.
// Per-pixel arithmetic and entry order are those of blend_bwd_kernel; the sums reach the accumulator rows in 16 instead of 4
// pieces per (tile, entry).  Iterations per wave: 0.87 of the visits of the kernel above on the 200k / 800x800 scene (tools/blend_stats.py:
// a splat that reaches a quadrant reaches 2.9 of its 4 blocks, and the four rows of a wave wait for the longest list of the chunk).
//
// MEASURED AND NOT THE DEFAULT (round 4; parity suite green; same lease, 200k / 800x800, ms per launch):
//     blend_bwd_kernel, DPP reduction (round 3)      0.304
//     blend_bwd_kernel, LDS reduction (the default)  0.269
//     this kernel                                    0.704     without its atomics 0.253, without reduction + atomics 0.189
// The float atomics, free in the kernel above (0.271 -> 0.262 without them), are what this design cannot afford: a visit there ends
// in 16 lanes adding to ONE 64-byte accumulator row -- the wave-wide sum has already merged the up to four blocks an entry reaches in
// the quadrant -- while an iteration here ends in 64 lanes adding to up to four rows, and whenever the rows of the wave sit on the same
// entry (large splats: most of the time) four lanes of one instruction hit the same address.  3.5 M (entry, block) pairs x 16 lanes
// instead of 1.2 M visits x 16: the L2's atomic units are the bound (0.45 ms).  Merging equal targets across the four rows before
// the atomic costs ~14 DPP-class instructions per iteration (two exchange steps of value + target) -- as much as the 13 % fewer
// iterations save, and the upper bound without any atomic is only 6 % under the default.  Kept as the committed A/B
// (-DDGS_BWD_ROWS=1, tools/ab_variants.sh); lane_pixel keeps the block layout, which costs the default nothing.
// ---- row-wise sums for the row-per-block backward (kernels_blend.h blend_bwd_rows_kernel) ----------------------------------
// Every 16-lane row r of the wave holds the 16 partials of ITS OWN list entry; wanted: for every row the 16 sums over its 16
// lanes.  Same transposition as above, in two rounds of 8 values through 2 KB: all lanes store value k into row k
// (ds_write_addtid_b32), then lane l -- value k = l >> 3, source row r' = (l >> 1) & 3, half h = l & 1 -- reads the 8 numbers of
// (k, r', h) with two ds_read_b128 (rotated by (l >> 4) & 1: conflict free, see above), adds them (7 v_add_f32) and joins the two
// halves with one quad DPP add.  After the two rounds lane l keeps the total of value (l >> 3) + 8 (l & 1) of row (l >> 1) & 3:
// 64 results, 64 lanes, one global atomic each -- no lane carries a duplicate.  Row 8 of the buffer transports one 32-bit word
// per lane (the surfel id of the row's entry) to the lanes that finish that row; row 9 a second one (deterministic variant).
struct RedRows {
    uint32_t m0;                 // LDS byte address of this wave's buffer
    const red_f32x4* rd[2];      // this lane's two read addresses
    const uint32_t* meta;        // word of source row (lane >> 1) & 3 in row 8 (row 9: + 64)
    __device__ __forceinline__ void init(float* buf /* [10][64] */, int lane)
    {
        m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)buf);
        const int k = lane >> 3, r = (lane >> 1) & 3, h = lane & 1, s = (lane >> 4) & 1;
#pragma unroll
        for (int i = 0; i < 2; i++) rd[i] = (const red_f32x4*)(buf + k * 64) + 4 * r + 2 * h + ((i + s) & 1);
        meta = (const uint32_t*)(buf + 8 * 64) + 16 * r;
    }
};

__device__ __forceinline__ void red_store_word(uint32_t w, int row /* 8 or 9 */)   // M0 as left by red_store8
{
    if (row == 8) asm volatile("ds_write_addtid_b32 %0 offset:2048" : : "v"(w) : "memory");
    else asm volatile("ds_write_addtid_b32 %0 offset:2304" : : "v"(w) : "memory");
}

__device__ __forceinline__ float red_pair(float t)   // t + the neighbouring lane's t (lanes 2 j, 2 j + 1)
{
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(t));
    return t;
}

// on return: the total of value (lane >> 3) + 8 (lane & 1) over the 16 lanes of row (lane >> 1) & 3, and in w8 (w9) the word that
// row's lanes passed as word8 (word9; only transported when TWO)
template <bool TWO>
__device__ __forceinline__ float rows_reduce16(float (&v)[16], uint32_t word8, uint32_t word9, const RedRows& r, int lane, uint32_t& w8, uint32_t& w9)
{
    red_store8(r.m0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    red_store_word(word8, 8);
    if (TWO) red_store_word(word9, 9);
    const red_f32x4 x0 = *r.rd[0], x1 = *r.rd[1];
    w8 = r.meta[0];
    w9 = TWO ? r.meta[64] : 0u;
    red_store8(r.m0, v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);   // (LDS operations of a wave execute in order: the reads above see round 1)
    const red_f32x4 y0 = *r.rd[0], y1 = *r.rd[1];
    const float lo = red_pair(red_sum4(x0) + red_sum4(x1));
    const float hi = red_pair(red_sum4(y0) + red_sum4(y1));
    return (lane & 1) ? hi : lo;
}

// sum over the 16 lanes of a row, on every lane of the row (rare 2-D filter branch of the backward)
__device__ __forceinline__ float row_sum16(float t)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    return t;
}

// maximum over the 16 lanes of a row, on every lane of the row
__device__ __forceinline__ int row_max16(int t)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    return t;
}


static_assert(kChunkB <= 60, "a row's byte list holds 64 slots and is read two ahead");
constexpr int kNullSlot = kChunkB;   // the slot behind the staged ones: the entry that contributes nothing
template <bool DET>
struct BwdRowStage {
    f32x4 a[3][kChunkB + 1];
    f32x4 tw[kChunkB + 1];     // (Tw.x Tw.y Tw.z opacity)
    f32x4 tuv[kChunkB + 1];    // (Tu.x Tu.y Tv.x Tv.y)
    f32x4 q3[kChunkB + 1];     // (n.x n.y n.z r)
    f32x4 q4[kChunkB + 1];     // (g b, 0-based list index as bits, byte offset of the surfel's accumulator row; null slot: index INT_MAX, offset ~0)
    uint32_t idx[4][16];       // row r: the slots of its block's entries in visit order, one byte each
    float red[DET ? 10 : 9][64];   // rows_reduce16: 8 values per round + one row of per-lane words (two in the deterministic variant)
};

template <bool DET>
__global__ void __launch_bounds__(kTilePix, DGS_BWD_MINWAVES) blend_bwd_rows_kernel(BlendBwdArgs a)
{
    __shared__ BwdRowStage<DET> s_stage[4];

    const int ntiles = a.tiles_x * a.tiles_y;
    int tile = tile_for_block(blockIdx.x, a.tiles_x, a.tiles_y, a.mode);
    if (a.mode < 3 && tile >= ntiles) return;
    if (a.mode >= 3) tile = (int)a.order[tile];
    if (tile >= ntiles) return;   // mode 4: empty slot
    if (a.tile_last[tile] == 0u) return;   // no pixel of the tile blended anything
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_pixel(tid, lx_, ly_);
    const int px = tx * kTileX + lx_, py = ty * kTileY + ly_;
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
    const float tpx = (float)(tx * kTileX), tpy = (float)(ty * kTileY);
    const float X0 = tpx + 8.0f, Y0 = tpy + 8.0f;
    const float qx = tpx + (float)(8 * (wave & 1)), qy = tpy + (float)(8 * (wave >> 1));
    const float qus0 = kSqrt2 * ((wave & 1) ? 0.5f : -7.5f), qvs0 = kSqrt2 * ((wave & 2) ? 0.5f : -7.5f);
    const float us = kSqrt2 * ((float)lx_ - 7.5f), vs = kSqrt2 * ((float)ly_ - 7.5f);
    const uint2 range = a.ranges[tile];
    BwdRowStage<DET>& S = s_stage[wave];

    const size_t plane = (size_t)ntiles * kTilePix;
    const size_t slot_px = (size_t)tile * kTilePix + tid;
    PixBwdA st;
    {
        float gpix[3] = {0.f, 0.f, 0.f}, goth[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (inside) {
            const size_t HW = (size_t)a.H * a.W;
            const size_t pix = (size_t)py * a.W + px;
#pragma unroll
            for (int c = 0; c < 3; c++) gpix[c] = a.dL_dpix[c * HW + pix];
#pragma unroll
            for (int c = 0; c < 8; c++) goth[c] = a.dL_dothers[c * HW + pix];
        }
        const int last = inside ? (int)a.n_contrib[slot_px] : 0;
        const int medc = inside ? (int)a.n_contrib[plane + slot_px] : 0;
        pixbwd_init_affine(st, inside ? a.final_T[slot_px] : 0.f, a.final_T[plane + slot_px], a.final_T[2 * plane + slot_px], last, medc, gpix,
                           goth, a.bg);
    }
    // a row's list ends at its own pixels' last contributor; the wave walks the tile's list back to front from the largest of the four
    const int row_last = row_max16(st.last_contributor);
    const int rl0 = __builtin_amdgcn_readlane(row_last, 0), rl1 = __builtin_amdgcn_readlane(row_last, 16);
    const int rl2 = __builtin_amdgcn_readlane(row_last, 32), rl3 = __builtin_amdgcn_readlane(row_last, 48);
    const int wave_last = max(max(rl0, rl1), max(rl2, rl3));
    RedRows rc;
    rc.init(&S.red[0][0], lane);
    // the null slot: planes of an entry that fails the alpha test for every pixel and whose other constants are finite
    if (lane < 7) {
        f32x4* planes[7] = {&S.a[0][kNullSlot], &S.a[1][kNullSlot], &S.a[2][kNullSlot], &S.tw[kNullSlot], &S.tuv[kNullSlot], &S.q3[kNullSlot], &S.q4[kNullSlot]};
        f32x4 z = mk4(0.f, 0.f, 0.f, 0.f);
        if (lane == 6) z = mk4(0.f, 0.f, __int_as_float(0x7fffffff), __uint_as_float(0xffffffffu));
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (lane == k) *planes[k] = z;
    }
    const uint8_t* my_list = (const uint8_t*)&S.idx[row][0];
    const int kk = (lane >> 3) + 8 * (lane & 1);   // the value this lane finishes (rows_reduce16)

    const bool stager = lane < kChunkB;
    uint32_t id_next = stager && wave_last - 1 - lane >= 0 ? a.point_list[range.x + (uint32_t)(wave_last - 1 - lane)] : 0u;
    for (int top = wave_last - 1; top >= 0; top -= kChunkB) {
        const int e_mine = top - lane;
        const uint32_t id = id_next;   // (lanes beyond the front of the list hold id 0: a valid record, masked out below)
        const float4* src = a.rec + (size_t)id * kRecQuads;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3s = src[3], q4s = src[4], bx = src[5];
        id_next = stager && e_mine - kChunkB >= 0 ? a.point_list[range.x + (uint32_t)(e_mine - kChunkB)] : 0u;
        const TileAffine ta = tile_affine(as_quad(q0), as_quad(q1), as_quad(q2), X0, Y0);
        const uint32_t bm = (stager & (e_mine >= 0)) ? blocks_hit_linear(ta, qus0, qvs0, as_quad(bx), qx, qy) : 0u;
        const bool h0 = (bm & 1u) && e_mine < rl0, h1 = (bm & 2u) && e_mine < rl1, h2 = (bm & 4u) && e_mine < rl2, h3 = (bm & 8u) && e_mine < rl3;
        const bool hit = h0 | h1 | h2 | h3;
        const unsigned long long m = ballot64(hit);
        if (m == 0ull) continue;
        const unsigned long long m0 = ballot64(h0), m1 = ballot64(h1), m2 = ballot64(h2), m3 = ballot64(h3);
        ((uint32_t*)&S.idx[0][0])[lane] = 0x01010101u * (uint32_t)kNullSlot;   // every list: null slots behind its entries
        if (hit) {
            const int slot = lane_rank(m);
            S.a[0][slot] = mk4(ta.a0.x, ta.a0.y, ta.a0.z, ta.a0.w);
            S.a[1][slot] = mk4(ta.a1.x, ta.a1.y, ta.a1.z, ta.a1.w);
            S.a[2][slot] = mk4(ta.a2.x, ta.a2.y, ta.a2.z, ta.a2.w);
            S.tw[slot] = mk4(q1.z, q1.w, q2.x, q2.w);
            S.tuv[slot] = mk4(q0.x, q0.y, q0.w, q1.x);
            S.q3[slot] = mk4(q3s);
            S.q4[slot] = mk4(q4s.x, q4s.y, __int_as_float(e_mine), __uint_as_float(id * (uint32_t)(kAccFloats * 4)));
            uint8_t* lists = (uint8_t*)&S.idx[0][0];
            if (h0) lists[lane_rank(m0)] = (uint8_t)slot;
            if (h1) lists[64 + lane_rank(m1)] = (uint8_t)slot;
            if (h2) lists[128 + lane_rank(m2)] = (uint8_t)slot;
            if (h3) lists[192 + lane_rank(m3)] = (uint8_t)slot;
        }
        __builtin_amdgcn_wave_barrier();   // the slice is private to this wave: its LDS writes above are ordered before its reads below
        const int n01 = max(__builtin_popcountll(m0), __builtin_popcountll(m1)), n23 = max(__builtin_popcountll(m2), __builtin_popcountll(m3));
        const int niter = __builtin_amdgcn_readfirstlane(max(n01, n23));   // (ballot popcounts: uniform, the loop counter belongs on the scalar unit)
        int sl = my_list[0];
        int sl_next = my_list[1];
        f32x4 a0 = S.a[0][sl], a1 = S.a[1][sl], a2 = S.a[2][sl];
        f32x4 tw = S.tw[sl], tuv = S.tuv[sl], q3 = S.q3[sl], q4 = S.q4[sl];
        for (int i = 0; i < niter; i++) {
            AlphaEval ev;
            bool ok = alpha_affine(us, vs, as_quad(a0), as_quad(a1), as_quad(a2), ev);
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(ev.a), "+v"(ev.alpha) : : "memory");   // see blend_fwd_kernel: one register set, loads pinned behind their last use
#endif
            sl = sl_next;
            a0 = S.a[0][sl]; a1 = S.a[1][sl]; a2 = S.a[2][sl];
            DGS_PIN4(a0); DGS_PIN4(a1); DGS_PIN4(a2);
            const int e = __float_as_int(q4.z);   // 0-based list index of the row's entry == the reference's `contributor`
            ok = ok & (e < st.last_contributor);
            if (ballot64(ok) != 0ull) {
                bool use3d;
                const float depth = alpha_depth(ev, tw.x, tw.y, tw.z, use3d);
                ok = ok & (depth >= kNear);
                float out[16], out2d[2];
                pixbwd_step_affine(st, ev, ok, use3d, depth, e, pfx, pfy, tw.x, tw.y, as_quad(tuv), tw.w, as_quad(q3),
                                   Quad{q4.x, q4.y, 0.f, 0.f}, out, out2d);
                uint32_t rid, re;
#if DGS_DIAG_BWD >= 2
                float tot = 0.f; rid = __float_as_uint(q4.w); re = 0;
                for (int k = 0; k < 16; k++) asm volatile("" : : "v"(out[k]));
#else
                const float tot = rows_reduce16<DET>(out, __float_as_uint(q4.w), __float_as_uint(q4.z), rc, lane, rid, re);
#endif
#if DGS_DIAG_BWD >= 1
                asm volatile("" : : "v"(tot), "v"(rid));
                if (false) {
#else
                if (rid != 0xffffffffu && tot != 0.0f) {
#endif   // (a row without an entry, or one none of whose pixels blended it, adds nothing)
                    if (DET) a.det_part[(((size_t)(range.x + re) * 4 + wave) * 4 + ((lane >> 1) & 3)) * kAccFloats + kk] = tot;
                    else atomicAdd((float*)((char*)a.acc + (rid + 4u * (uint32_t)kk)), tot);   // rid = byte offset of the surfel's accumulator row
                }
                if (ballot64(ok && !use3d) != 0ull) {  // rare 2-D filter branch (backward.cu:436-443)
                    const float mx = row_sum16(out2d[0]);
                    const float my = row_sum16(out2d[1]);
                    if ((lane & 15) == 0 && (mx != 0.0f || my != 0.0f)) {
                        float* dst = DET ? a.det_part + (((size_t)(range.x + (uint32_t)e) * 4 + wave) * 4 + row) * kAccFloats
                                         : (float*)((char*)a.acc + __float_as_uint(q4.w));
                        if (DET) { dst[kAccMean2D] = mx; dst[kAccMean2D + 1] = my; }
                        else { atomicAdd(dst + kAccMean2D, mx); atomicAdd(dst + kAccMean2D + 1, my); }
                    }
                }
            }
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(st.T) : : "memory");
#endif
            tw = S.tw[sl]; tuv = S.tuv[sl]; q3 = S.q3[sl]; q4 = S.q4[sl];
            DGS_PIN4(tw); DGS_PIN4(tuv); DGS_PIN4(q3); DGS_PIN4(q4);
            sl_next = my_list[i + 2];
            asm volatile("" : "+v"(sl_next));   // requested here, a visit before the address is formed from it
        }
        __builtin_amdgcn_wave_barrier();
    }
}

