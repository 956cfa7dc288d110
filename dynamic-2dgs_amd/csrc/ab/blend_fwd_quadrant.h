// ab/blend_fwd_quadrant.h -- the round-3 forward blend: one list per WAVE (8x8 quadrant); superseded by blend_fwd_rows_kernel (0.126 -> 0.120 ms, bit-identical results).  Select with -DDGS_AB_BUILD -DDGS_FWD_ROWS=0.
// A/B material: measured, parity-green when it was measured, NOT part of the product library.  Included by kernels_blend.h only under
// -DDGS_AB_BUILD (tools/ab_variants.sh); the numbers that retired it are in the comments below and in DESIGN.md section 4.
// (no include guard / namespace of its own: textually included inside namespace dgs)

struct FwdStage {            // one wave's staging slice: the chunk's visited entries, compacted (+1: the visit loop reads one slot ahead)
    f32x4 a[3][kChunk + 1];  // alpha part of the entry's affine image (tile_affine)
    f32x4 tw[kChunk + 1];    // (Tw.x Tw.y Tw.z, 1-based list position as bits)
    f32x4 q3[kChunk + 1];    // (n.x n.y n.z r)
    f32x4 q4[kChunk + 1];    // (g b - -): a 16-byte plane like the others, so one address register serves all six
};

__global__ void __launch_bounds__(kTilePix, DGS_FWD_MINWAVES) blend_fwd_kernel(BlendFwdArgs a)
{
    __shared__ FwdStage s_stage[4];
    __shared__ uint32_t s_max[4];

    const int ntiles = a.tiles_x * a.tiles_y;
    int tile = tile_for_block(blockIdx.x, a.tiles_x, a.tiles_y, a.mode);
    if (a.mode < 3 && tile >= ntiles) return;
    if (a.mode >= 3) tile = (int)a.order[tile];
    if (tile >= ntiles) return;   // mode 4: empty slot
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_pixel(tid, lx_, ly_);
    const int px = tx * kTileX + lx_, py = ty * kTileY + ly_;
    const bool inside = px < a.W && py < a.H;
    const float tpx = (float)(tx * kTileX), tpy = (float)(ty * kTileY);
    const float X0 = tpx + 8.0f, Y0 = tpy + 8.0f;
    // this wave's quadrant: first pixel, and its span in the scaled tile-relative coordinates of the affine form
    const float qx = tpx + (float)(8 * (wave & 1)), qy = tpy + (float)(8 * (wave >> 1));
    const float qus0 = kSqrt2 * ((wave & 1) ? 0.5f : -7.5f), qvs0 = kSqrt2 * ((wave & 2) ? 0.5f : -7.5f);
    // sqrt2 x (pixel - tile centre); NaN = this pixel takes no further entry (outside the image, or saturated)
    float us = inside ? kSqrt2 * ((float)lx_ - 7.5f) : __builtin_nanf("");
    const float vs = kSqrt2 * ((float)ly_ - 7.5f);

    const uint2 range = a.ranges[tile];
    const uint32_t len = range.y - range.x;
    FwdStage& S = s_stage[wave];

    PixFwd st;
    pixfwd_init(st);

    // Visit, in list order, the nhit entries the wave has staged.  Every instruction of this loop is issued once per (wave, entry)
    // visit, scalar ones included (the CU's scalar unit issues ~1 instruction per cycle for all four SIMDs: 25 scalar
    // instructions per visit -- bit-scan of a visit mask, saturation ballots, early-out tests -- cost as much issue time as the
    // arithmetic).  Hence: the staged entries are COMPACTED (a counted loop over consecutive slots), a pixel that saturates
    // (forward.cu:402-406: it blends neither this entry nor any later one) is poisoned with one select, and whether the wave
    // still has live pixels is tested per chunk, not per visit.
    auto visit = [&](auto track_median, int nhit) {
        f32x4 a0 = S.a[0][0], a1 = S.a[1][0], a2 = S.a[2][0];
        f32x4 tw = S.tw[0], q3 = S.q3[0], q4 = S.q4[0];
        for (int i = 0; i < nhit; i++) {
            AlphaEval e;
            const bool pass = alpha_affine(us, vs, as_quad(a0), as_quad(a1), as_quad(a2), e);
            // Software pipeline over the visited entries with ONE register set: the next entry's alpha part is requested as soon
            // as this one's has been consumed, its Tw / normal / colour at the end of the visit -- each INTO THE SAME registers,
            // a good hundred cycles before it is needed.  The empty asm statements pin the order: left alone the compiler hoists
            // the loads above the evaluation, needs a second register set and pays twelve moves per visit to rotate it.
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(e.a), "+v"(e.alpha) : : "memory");
#endif
            a0 = S.a[0][i + 1]; a1 = S.a[1][i + 1]; a2 = S.a[2][i + 1];
            bool use3d;
            const float depth = alpha_depth(e, tw.x, tw.y, tw.z, use3d);
            float w, test_T;
            pixfwd_weight(st, e.alpha, w, test_T);
            const bool ok = pass & (depth >= kNear);      // forward.cu:388 (float 0.2f: same set as (double)depth < 0.2)
            const bool blend = ok & !(test_T < kTmin);
            if (blend) {
                st.contributor = __float_as_uint(tw.w);   // 1-based list position (forward.cu:356)
                pixfwd_accumulate<decltype(track_median)::value>(st, w, test_T, depth, as_quad(q3), Quad{q4.x, q4.y, 0.f, 0.f});
            }
            us = (ok ^ blend) ? __builtin_nanf("") : us;   // passed but saturated: the pixel is finished
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(st.T), "+v"(us) : : "memory");
#endif
            tw = S.tw[i + 1]; q3 = S.q3[i + 1]; q4 = S.q4[i + 1];
        }
    };

    uint32_t id_next = lane < len ? a.point_list[range.x + lane] : 0u;
    unsigned long long alive = ballot64(inside);   // lanes that still take entries (wave-uniform)
    for (uint32_t base = 0; base < len && alive != 0ull; base += kChunk) {
        const uint32_t e_mine = base + (uint32_t)lane;
        const uint32_t id = id_next;   // (lanes beyond the end of the list hold id 0: a valid record, masked out below)
        // Straight-line staging: all six quads of the record are requested at once and every lane runs the whole test (a
        // conditional ladder -- list end, box, footprint -- makes the compiler sink each load behind the test before it:
        // five dependent trips to memory per chunk).  Entries that can touch the quadrant are compacted: slot = rank among them.
        const float4* src = a.rec + (size_t)id * kRecQuads;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4], bx = src[5];
        // the id of the lane's next entry travels while this chunk is visited (the record loads of the next step then start at once)
        id_next = e_mine + kChunk < len ? a.point_list[range.x + e_mine + kChunk] : 0u;
        const TileAffine ta = tile_affine(as_quad(q0), as_quad(q1), as_quad(q2), X0, Y0);
        const bool hit = (e_mine < len) & block_box_hit(bx, qx, qy) & block_hit_affine(ta, qus0, qus0 + 7.0f * kSqrt2, qvs0, qvs0 + 7.0f * kSqrt2);
        const unsigned long long m = ballot64(hit);
        if (m == 0ull) continue;
        if (hit) {
            const int slot = lane_rank(m);
            S.a[0][slot] = mk4(ta.a0.x, ta.a0.y, ta.a0.z, ta.a0.w);
            S.a[1][slot] = mk4(ta.a1.x, ta.a1.y, ta.a1.z, ta.a1.w);
            S.a[2][slot] = mk4(ta.a2.x, ta.a2.y, ta.a2.z, ta.a2.w);
            S.tw[slot] = mk4(q1.z, q1.w, q2.x, __uint_as_float(e_mine + 1u));
            S.q3[slot] = mk4(q3);
            S.q4[slot] = mk4(q4);
        }
        __builtin_amdgcn_wave_barrier();   // the slice is private to this wave: its LDS writes above are ordered before its reads below
        // median bookkeeping (forward.cu:421-425) only while some pixel of the wave still has T > 0.5
        if (ballot64(st.T > 0.5f && us == us) != 0ull) visit(std::true_type{}, __builtin_popcountll(m));
        else visit(std::false_type{}, __builtin_popcountll(m));
        __builtin_amdgcn_wave_barrier();
        alive = ballot64(us == us);   // wave-level early out (forward.cu:334-336 votes per block)
    }

    // per-tile maximum of the last contributor: the backward starts there instead of walking the
    // whole list (backward.cu:276-279 skips those entries one by one)
    uint32_t m = inside ? st.last : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    if (tid == 0) {
        uint32_t mm = s_max[0];
        mm = s_max[1] > mm ? s_max[1] : mm;
        mm = s_max[2] > mm ? s_max[2] : mm;
        mm = s_max[3] > mm ? s_max[3] : mm;
        a.tile_last[tile] = mm;
    }

    const size_t plane = (size_t)ntiles * kTilePix;
    const size_t slot = (size_t)tile * kTilePix + tid;
    a.final_T[slot] = st.T;
    a.final_T[plane + slot] = st.dist1;
    a.final_T[2 * plane + slot] = st.dist2;
    a.n_contrib[slot] = st.last;
    a.n_contrib[plane + slot] = st.med_c;
    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)py * a.W + px;
        a.out_color[pix] = st.C[0] + st.T * a.bg[0];
        a.out_color[HW + pix] = st.C[1] + st.T * a.bg[1];
        a.out_color[2 * HW + pix] = st.C[2] + st.T * a.bg[2];
        a.out_others[pix] = st.D;                 // DEPTH_OFFSET 0   (auxiliary.h:25-30)
        a.out_others[HW + pix] = 1.f - st.T;      // ALPHA_OFFSET 1
        a.out_others[2 * HW + pix] = st.N[0];     // NORMAL_OFFSET 2..4
        a.out_others[3 * HW + pix] = st.N[1];
        a.out_others[4 * HW + pix] = st.N[2];
        a.out_others[5 * HW + pix] = st.med_d;    // MIDDEPTH_OFFSET 5
        a.out_others[6 * HW + pix] = st.distortion;  // DISTORTION_OFFSET 6
        a.out_others[7 * HW + pix] = st.med_w;    // MEDIAN_WEIGHT_OFFSET 7
    }
}

