// kernels_blend.h -- per-tile forward and backward alpha blend for gfx950.
//
// One 16x16 screen tile per 256-thread workgroup = 4 wave64; wave w owns the 8x8 pixel quadrant (w & 1, w >> 1) of the
// tile (surfel_math.h lane_pixel), so all 64 lanes of a wave consume the same staged surfel at the same time.
//
// Since round 3 the four waves of a tile run AUTONOMOUSLY: each walks the tile's depth-sorted list in chunks of 64 entries, one
// entry per lane.  The lane gathers the entry's packed 96-B record, turns it into the entry's AFFINE image for this tile
// (surfel_math.h tile_affine: p = A + dxs B + dys C around the projected centre -- 6 FMAs per pixel instead of 12 operations),
// tests it against the wave's own quadrant (exact pixel box of the record, then the exact footprint: block_hit_affine) and, if it
// can touch the quadrant, stages it in the WAVE'S OWN slice of LDS: three float4 planes for the alpha test, then Tw, normal and
// colour planes that are only read when some pixel blends the entry.  One ballot gives the chunk's 64-bit visit mask; the inner
// loop walks its set bits and reads the planes with wave-uniform addresses (LDS broadcast, conflict free).  No workgroup barrier
// inside the list loop: with the shared 256-entry batches of rounds 1-2 a wave spent 30-40 % of its cycles parked at the two
// barriers per batch waiting for its three siblings (SQ_WAIT_ANY, profiles/r02_pmc_blend_metric.json), and a tile ran as long
// as its slowest quadrant TWICE over (once per barrier).  The price: a record is fetched and its affine image computed by up to
// four waves instead of one (staging is ~100 VALU instructions per 64 entries against ~1300 for their visits).
// Replaces renderCUDA of forward.cu:265-463 and backward.cu:143-449.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "surfel_math.h"
#include "wave_reduce.h"

namespace dgs {

#ifndef DGS_FWD_MINWAVES
#define DGS_FWD_MINWAVES 5   // (the long-tile path's extra state must not cost the short path its fifth wave per SIMD)
#endif
#ifndef DGS_PIN_PREFETCH
#define DGS_PIN_PREFETCH 1
#endif
// ---- A/B switches ------------------------------------------------------------------------------------------------------------
// The product library is built with NONE of these set: the superseded kernels and reductions they select live in csrc/ab/ and are
// only reachable with -DDGS_AB_BUILD (tools/ab_variants.sh builds such twins next to the product, tools/ab_check.sh runs the parity
// file and the timing on each).  DGS_DIAG_BWD modes produce WRONG results by design (they remove work to price it).
#ifndef DGS_FWD_ROWS
#define DGS_FWD_ROWS 1       // 0: ab/blend_fwd_quadrant.h (one list per wave)
#endif
#ifndef DGS_BWD_ROWS
#define DGS_BWD_ROWS 0       // 1: ab/blend_bwd_rows.h (one list per 16-lane row: 2.6 x slower on the L2's float-atomic rate)
#endif
#ifndef DGS_BWD_REDUCE
#define DGS_BWD_REDUCE 4     // 0-3: ab/reduce_variants.h (3 = the DPP sum of wave_reduce.h), 4: transposition through the wave's own LDS (wave_reduce.h)
#endif
#ifndef DGS_DIAG_BWD
#define DGS_DIAG_BWD 0       // 1 no atomics, 2 no reduction either, 3 evaluation + loop only
#endif
#if !defined(DGS_AB_BUILD) && (DGS_FWD_ROWS != 1 || DGS_BWD_ROWS != 0 || DGS_BWD_REDUCE != 4 || DGS_DIAG_BWD != 0)
#error "A/B variants and diagnostic modes are not part of the product library: build them with -DDGS_AB_BUILD (tools/ab_variants.sh)"
#endif
constexpr int kChunk = 64;   // list entries staged per wave and step: one per lane

__device__ __forceinline__ Quad as_quad(const float4& v) { return Quad{v.x, v.y, v.z, v.w}; }

// Staged planes are read as native 4-vectors and pinned as such (an empty asm that takes the whole vector): the loop-carried
// float4 of a software-pipelined read is otherwise split by the compiler into four ds_read_b32 with an address register each.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Quad as_quad(const f32x4& v) { return Quad{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ f32x4 mk4(float x, float y, float z, float w) { return f32x4{x, y, z, w}; }
__device__ __forceinline__ f32x4 mk4(const float4& v) { return f32x4{v.x, v.y, v.z, v.w}; }
#define DGS_PIN4(v) asm volatile("" : "+v"(v))

// __ballot() takes an int: the bool -> int -> "!= 0" round trip is not always folded back (a v_cndmask 0/1 + v_cmp_ne per use in the
// blend loops when the predicate is an AND of lane masks); the builtin takes the i1 as it is
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ unsigned long long wave_uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

struct LongThr { uint32_t fwd, bwd; };   // long-tile path: thresholds of this launch (list length / traversed length), in the image buffer

struct BlendFwdArgs {
    const uint2* ranges;         // [T]
    const uint32_t* order;       // [T] dispatch order (mode 3) or null
    const uint32_t* point_list;  // [R] sorted surfel ids
    const float4* rec;           // [P*5]
    int W, H, tiles_x, tiles_y, mode;
    const float* bg;             // [3] device
    float* final_T;              // [3][T*256] tile-major: T, dist1, dist2
    uint32_t* n_contrib;         // [2][T*256] tile-major: last contributor, median contributor
    uint32_t* tile_last;         // [T] max over the tile of `last contributor` (zeroed by scan_tiles_kernel, written with atomicMax)
    const LongThr* long_thr;     // thresholds of the long-tile path, or null (path off: grid = blend_grid_size)
    float* out_color;            // [3,H,W]
    float* out_others;           // [8,H,W]
    // accumulator rider: the first `clear_blocks` workgroups (a multiple of 8: the XCD of every tile workgroup stays what it was) zero the
    // backward's per-surfel accumulator rows -- 80 B per surfel of stores that ride under this VALU-bound kernel instead of costing the
    // backward a 10-us launch of their own -- and *clean_flag says so to prep_bwd_kernel (1 = rows are zero; the backward blend writes 0)
    float4* clear;
    uint32_t clear_n4;
    int clear_blocks;
    uint32_t* clean_flag;
};

// Tile order.  The dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md) and every XCD has a private
// 4 MiB L2, so which tiles share an XCD decides how often a surfel record is re-fetched.  Modes (A/B-able at
// run time through dgs_set_option(DGS_OPT_TILE_ORDER, m)):
//   0  plain row-major blockIdx (neighbouring tiles land on 8 different L2s),
//   1  contiguous: XCD x gets tiles [x*T/8, (x+1)*T/8)  (best locality, but whole image bands -- empty sky vs
//      dense centre -- go to one XCD: load imbalance),
//   2  row-interleaved: XCD x gets tile rows r with r % 8 == x, walked left to right (horizontal neighbours
//      share an L2, every XCD samples the whole image height).
//   4  XCD-local groups, longest first inside an XCD: the image is cut into 4 x 4-tile groups (a surfel's tiles mostly fall
//      into one), the groups are dealt to the 8 XCDs heaviest-first in snake order (balanced sums), and XCD x walks ITS
//      tiles in descending length: order[8 k + x] = its k-th tile (tile_order_body; empty slots hold `ntiles`).
__device__ __forceinline__ int tile_for_block(int bid, int tiles_x, int tiles_y, int mode)
{
    constexpr int kXcd = 8;
    const int ntiles = tiles_x * tiles_y;
    if (mode == 0 || mode >= 3) return bid;  // modes 3, 4 index the sorted order[] with it
    const int xcd = bid % kXcd, k = bid / kXcd;
    if (mode == 1) {
        const int per = (ntiles + kXcd - 1) / kXcd;
        return xcd * per + k;  // may be >= ntiles for the padded tail; caller checks
    }
    // mode 2: the k-th tile of this XCD is in its (k / tiles_x)-th row
    const int row = (k / tiles_x) * kXcd + xcd;
    if (row >= tiles_y) return ntiles;
    return row * tiles_x + (k % tiles_x);
}

// Mode 3 (default): longest-processing-time-first.  Tile work varies by >10x (empty corners vs the dense centre)
// and a workgroup owns its tile to the end, so with ~7 resident workgroups per CU the kernel time is set by
// whichever SIMD drew the heaviest tiles last.  Dispatching tiles in descending list length (counting sort below)
// puts the long tiles first and lets the short ones fill the tail.
constexpr int kOrderBins = 1024;
constexpr int kOrderGroup = 4;      // mode 4: tiles per group edge
constexpr int kOrderMaxGroups = 1024;
// mode 4: slots of the order array = 8 x (most tiles one XCD can get): every XCD gets at most ceil(G / 8) + 1 groups of 16 tiles
__host__ __device__ inline int order_groups(int tiles_x, int tiles_y)
{
    return ((tiles_x + kOrderGroup - 1) / kOrderGroup) * ((tiles_y + kOrderGroup - 1) / kOrderGroup);
}
__host__ __device__ inline int order_slots(int tiles_x, int tiles_y)
{
    return 8 * kOrderGroup * kOrderGroup * ((order_groups(tiles_x, tiles_y) + 7) / 8 + 1);
}

// body shared by tile_order_kernel and scan_tiles_kernel (which orders the forward's tiles right after it has scanned their
// counts: one launch less in the step); all 1024 threads of the workgroup must call it
__device__ __forceinline__ void tile_order_body(const uint2* ranges, const uint32_t* weights, int ntiles, uint32_t* order, uint32_t* s_hist /*[kOrderBins]*/,
                                                uint32_t* s_wsum /*[16]*/)
{
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    // bin 0 = heaviest.  weight = list length (ranges) or traversed length (weights), 4 entries per bin, saturating.
    // Both passes take four tiles per thread and trip with all four loads in flight (this body is the serial tail of the launch it
    // rides in -- one workgroup, the rest of the device idle: a load -> LDS atomic loop was one memory round trip per 1024 tiles
    // and pass, five at 2500 tiles)
    auto weight4 = [&](int t0, uint32_t (&w)[4]) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const int t = t0 + 1024 * l;
            if (weights) w[l] = t < ntiles ? weights[t] : 0u;
            else { const uint2 r = t < ntiles ? ranges[t] : make_uint2(0u, 0u); w[l] = r.y - r.x; }
        }
    };
    for (int t0 = tid; t0 < ntiles; t0 += 4096) {
        uint32_t w[4];
        weight4(t0, w);
#pragma unroll
        for (int l = 0; l < 4; l++)
            if (t0 + 1024 * l < ntiles) atomicAdd(&s_hist[kOrderBins - 1 - min(w[l] >> 2, (uint32_t)(kOrderBins - 1))], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins (one per thread)
    const uint32_t v = s_hist[tid];
    uint32_t inc = v;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(inc, d, 64);
        if (lane >= d) inc += n;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    __syncthreads();
    s_hist[tid] = base + inc - v;
    __syncthreads();
    for (int t0 = tid; t0 < ntiles; t0 += 4096) {
        uint32_t w[4], pos[4];
        weight4(t0, w);
#pragma unroll
        for (int l = 0; l < 4; l++)
            pos[l] = t0 + 1024 * l < ntiles ? atomicAdd(&s_hist[kOrderBins - 1 - min(w[l] >> 2, (uint32_t)(kOrderBins - 1))], 1u) : 0u;
#pragma unroll
        for (int l = 0; l < 4; l++)
            if (t0 + 1024 * l < ntiles) order[pos[l]] = (uint32_t)(t0 + 1024 * l);
    }
}

// exclusive scan of s[0 .. 1024) in place, one element per thread of the 1024-thread workgroup
__device__ __forceinline__ void scan1024(uint32_t* s, uint32_t* s_wsum /*[16]*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t v = s[tid];
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(inc, d, 64);
        if (lane >= d) inc += n;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    __syncthreads();
    s[tid] = base + inc - v;
    __syncthreads();
}

// Mode 4 (see tile_for_block).  s_hist: [8][kOrderBins]; s_gw, s_gx: [kOrderMaxGroups].  All 1024 threads must call it.
// group_xcd [kOrderMaxGroups] (global): the group -> XCD map.  The forward (deal = true) deals the groups by list length and
// stores the map; the backward reuses it (its weights, the traversed lengths, follow the list lengths, and the records a
// group's tiles share should meet the L2 that already holds them): no second ranking.
__device__ __forceinline__ void tile_order_xcd_body(const uint2* ranges, const uint32_t* weights, int tiles_x, int tiles_y, uint32_t* order,
                                                    uint32_t* group_xcd, bool deal, uint32_t* s_hist, uint32_t* s_gw, uint32_t* s_gx,
                                                    uint32_t* s_wsum)
{
    const int tid = threadIdx.x;
    const int ntiles = tiles_x * tiles_y, gtx = (tiles_x + kOrderGroup - 1) / kOrderGroup, ngroups = order_groups(tiles_x, tiles_y);
    const int slots = order_slots(tiles_x, tiles_y);
    // a thread's tiles are tid, tid + 1024, ..: weight and group of the first kKeep stay in registers (800 x 800: 3 per thread)
    constexpr int kKeep = 4;
    uint32_t wk[kKeep], gk[kKeep];
    auto weight = [&](int t) { return weights ? weights[t] : (ranges[t].y - ranges[t].x); };
    auto group = [&](int t) { return (uint32_t)(((t / tiles_x) / kOrderGroup) * gtx + (t % tiles_x) / kOrderGroup); };
#pragma unroll
    for (int i = 0; i < kKeep; i++) {
        const int t = tid + 1024 * i;
        wk[i] = t < ntiles ? weight(t) : 0u;
        gk[i] = t < ntiles ? group(t) : 0u;
    }
    auto wt = [&](int i, int t) { return i < kKeep ? wk[i] : weight(t); };
    auto gr = [&](int i, int t) { return i < kKeep ? gk[i] : group(t); };
    s_gw[tid] = 0;
    for (int i = tid; i < 8 * kOrderBins; i += 1024) s_hist[i] = 0;
    for (int i = tid; i < slots; i += 1024) order[i] = (uint32_t)ntiles;   // empty slot
    if (!deal && tid < ngroups) s_gx[tid] = group_xcd[tid] & 7u;   // (& 7: a forward under another tile order left no map -- any map is valid)
    __syncthreads();
    if (deal) {
        for (int i = 0, t = tid; t < ntiles; i++, t += 1024) atomicAdd(&s_gw[gr(i, t)], wt(i, t));
        __syncthreads();
        // rank of every group by weight (descending; 64 entries per bin, ties in any order) -> its XCD in snake order
        const uint32_t gwt = tid < ngroups ? s_gw[tid] : 0u;
        const uint32_t gbin = kOrderBins - 1 - min(gwt >> 6, (uint32_t)(kOrderBins - 1));
        if (tid < ngroups) atomicAdd(&s_hist[gbin], 1u);
        __syncthreads();
        scan1024(s_hist, s_wsum);
        if (tid < ngroups) {
            const uint32_t r = atomicAdd(&s_hist[gbin], 1u);
            const uint32_t x = (r >> 3) & 1u ? 7u - (r & 7u) : (r & 7u);
            s_gx[tid] = x;
            group_xcd[tid] = x;
        }
        __syncthreads();
        s_hist[tid] = 0;
        __syncthreads();
    }
    // per XCD: counting sort of its tiles by weight, heaviest first
    for (int i = 0, t = tid; t < ntiles; i++, t += 1024) {
        const uint32_t bin = kOrderBins - 1 - min(wt(i, t) >> 2, (uint32_t)(kOrderBins - 1));
        atomicAdd(&s_hist[s_gx[gr(i, t)] * kOrderBins + bin], 1u);
    }
    __syncthreads();
    {   // the 8 exclusive scans at once: thread <-> bin, one shuffle network carrying 8 values
        const int lane = tid & 63, wave = tid >> 6;
        uint32_t v[8], inc[8];
#pragma unroll
        for (int x = 0; x < 8; x++) { v[x] = s_hist[x * kOrderBins + tid]; inc[x] = v[x]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
#pragma unroll
            for (int x = 0; x < 8; x++) {
                const uint32_t n = __shfl_up(inc[x], d, 64);
                if (lane >= d) inc[x] += n;
            }
        if (lane == 63)
#pragma unroll
            for (int x = 0; x < 8; x++) s_gw[x * 16 + wave] = inc[x];   // the group weights are no longer needed
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 8; x++) {
            uint32_t base = 0;
            for (int w = 0; w < wave; w++) base += s_gw[x * 16 + w];
            s_hist[x * kOrderBins + tid] = base + inc[x] - v[x];
        }
        __syncthreads();
    }
    for (int i = 0, t = tid; t < ntiles; i++, t += 1024) {
        const uint32_t x = s_gx[gr(i, t)];
        const uint32_t bin = kOrderBins - 1 - min(wt(i, t) >> 2, (uint32_t)(kOrderBins - 1));
        const uint32_t pos = atomicAdd(&s_hist[x * kOrderBins + bin], 1u);
        order[8u * pos + x] = (uint32_t)t;
    }
}

__global__ void __launch_bounds__(1024) tile_order_kernel(const uint2* ranges, const uint32_t* weights, int tiles_x, int tiles_y, int mode,
                                                          uint32_t* order, uint32_t* group_xcd)
{
    __shared__ uint32_t s_hist[8 * kOrderBins];
    __shared__ uint32_t s_gw[kOrderMaxGroups], s_gx[kOrderMaxGroups];
    __shared__ uint32_t s_wsum[16];
    if (mode == 4) tile_order_xcd_body(ranges, weights, tiles_x, tiles_y, order, group_xcd, false, s_hist, s_gw, s_gx, s_wsum);
    else tile_order_body(ranges, weights, tiles_x * tiles_y, order, s_hist, s_wsum);
}

// The backward's two preparations as ONE launch: workgroups 0 .. n-2 clear the surfels' accumulator rows (what hipMemsetAsync did),
// the last one orders the tiles by the length the forward traversed (tile_order_kernel) -- the two were 7 + 6 us back to back in
// front of the backward blend of every step.
__global__ void __launch_bounds__(1024) prep_bwd_kernel(float4* __restrict__ acc, size_t n4, const uint32_t* weights, int tiles_x, int tiles_y, int mode,
                                                        uint32_t* order, uint32_t* group_xcd, uint32_t* long_thr /*[2]: [1] written here*/, uint32_t long_div,
                                                        const uint32_t* clean_flag /*1: the forward blend's rider zeroed the rows, or null*/)
{
    __shared__ uint32_t s_hist[8 * kOrderBins];
    __shared__ uint32_t s_gw[kOrderMaxGroups], s_gx[kOrderMaxGroups];
    __shared__ uint32_t s_wsum[16];
    if (blockIdx.x == gridDim.x - 1) {
        {   // long-tile path of the backward blend: a tile is long from 512 traversed entries and (sum of the traversed lengths) / long_div on
            const int ntiles = tiles_x * tiles_y;
            uint32_t sum = 0;
            for (int t0 = threadIdx.x; t0 < ntiles; t0 += 4096) {   // four loads in flight per trip (tile_order_body)
                uint32_t w[4];
#pragma unroll
                for (int l = 0; l < 4; l++) w[l] = t0 + 1024 * l < ntiles ? weights[t0 + 1024 * l] : 0u;
                sum += (w[0] + w[1]) + (w[2] + w[3]);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
            if ((threadIdx.x & 63) == 0) s_wsum[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tot = 0;
                for (int w = 0; w < 16; w++) tot += s_wsum[w];
                long_thr[1] = max(512u, tot / max(long_div, 1u));
            }
            __syncthreads();
        }
        if (mode == 4) tile_order_xcd_body(nullptr, weights, tiles_x, tiles_y, order, group_xcd, false, s_hist, s_gw, s_gx, s_wsum);
        else tile_order_body(nullptr, weights, tiles_x * tiles_y, order, s_hist, s_wsum);
        return;
    }
    if (clean_flag && *clean_flag == 1u) return;
    const size_t stride = (size_t)(gridDim.x - 1) * 1024;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) acc[i] = z;
}

// grid size that covers every tile under `mode`
inline int blend_grid_size(int tiles_x, int tiles_y, int mode)
{
    const int ntiles = tiles_x * tiles_y;
    if (mode == 0 || mode == 3) return ntiles;
    if (mode == 4) return order_slots(tiles_x, tiles_y);
    if (mode == 1) return ((ntiles + 7) / 8) * 8;
    return ((tiles_y + 7) / 8) * tiles_x * 8;
}

// What the instruction mix of the loop below costs on gfx950 (tools/micro/valu_issue_bench, profiles/r03_valu_issue_gfx950.txt):
// FMA / mul / add with VGPR operands 2.5 cycles per wave, anything with an SGPR operand, comparisons, min/max, selects 4.4-4.8,
// v_rcp / v_exp 8.5-12.7, and every scalar instruction takes one of the CU's ~1 per cycle scalar issue slots.  So the visit is
// built from plain FMAs on VGPR operands (entry constants arrive through LDS broadcasts, not SGPRs), one comparison decides the
// alpha test (alpha_affine), finished and outside pixels are poisoned with NaN coordinates instead of being masked, and the
// median bookkeeping is skipped once no pixel of the wave has T > 0.5.
// does the record's exact pixel box (q5) reach a pixel centre of the 8x8 block whose first pixel is (qx, qy)?
__device__ __forceinline__ bool block_box_hit(const float4& bx, float qx, float qy)
{
    return (bx.y >= qx + 0.5f) & (bx.x <= qx + 7.5f) & (bx.w >= qy + 0.5f) & (bx.z <= qy + 7.5f);
}

// slot of this lane among the set bits of m (number of set bits below the lane)
__device__ __forceinline__ int lane_rank(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#if !DGS_FWD_ROWS
#include "ab/blend_fwd_quadrant.h"   // (DGS_AB_BUILD only, checked above)
#endif

// ---- forward blend, one list per 16-lane row (round 4) ----------------------------------------------------------------------
// Same idea as ab/blend_bwd_rows.h, where it is described; in the forward nothing is summed across lanes and nothing is
// added to global memory, so the design pays here: every DPP row of the wave -- one 4x4 pixel block of the quadrant -- walks the
// list of the entries that can reach ITS block (blocks_hit_linear while staging; a block whose 16 pixels are all finished takes no
// more entries), the planes of an entry are stored once per wave, and the visit loop runs max-over-rows(list length) times with
// four slot addresses per LDS read instead of one.  Per-pixel arithmetic and entry order are those of the round-3 forward (ab/blend_fwd_quadrant.h): the results
// are bit-identical.  0.87 of that kernel's visits on the 200k / 800x800 scene (tools/blend_stats.py).

// ---- long tiles (round 4) ---------------------------------------------------------------------------------------------------
// A tile is one workgroup and a quadrant one wave that walks the tile's list serially, so a launch is never shorter than its longest
// list -- and a lone wave issues one instruction per ~8 cycles however much of the device is idle: on a densified scene, whose few
// covered tiles hold lists of 1-2 k entries, the heaviest tile alone was the launch.  The `kLongSlots` tiles at
// the head of the longest-first dispatch order whose length exceeds a per-launch threshold (long_thr: written next to the order by the
// kernel that sorts it) therefore get FOUR workgroups, one per quadrant, whose four waves share the list: both recurrences of the
// backward and the forward's T <- T - alpha T are linear in the state that enters a stretch of the list, so a stretch can be walked
// before that state is known (a light pass leaves the stretch's transfer per pixel), the transfers are composed, and the real pass
// starts from the state that reaches the stretch.  Backward: quarters of the traversed range (bwd_quadrant<.., LONG = true>); forward:
// rounds of four chunks, which end where the serial walk ends (blend_fwd_long).  1.4-1.5 x the arithmetic on a quarter of the critical
// path.  What differs from the serial walk is rounding (products and sums taken stretch-wise); the results are deterministic.
// Grid (tile order 3 only): blocks [0, 4 kLongSlots) = (slot, quadrant) pairs, quadrant-major -- a short tile in one of these slots is
// rendered by its quadrant-0 block, the other three exit --, blocks behind them the remaining slots of the order.
constexpr int kLongSlots = 256;

// blockIdx -> (tile, quadrant of a long tile or -1).  Returns false when the block has nothing to do.
__device__ __forceinline__ bool long_decode(int bid, const uint32_t* order, int ntiles, uint32_t thr, const uint32_t* lengths /*[T] or null*/,
                                            const uint2* ranges, int& tile, int& lq)
{
    lq = -1;
    if (bid >= 4 * kLongSlots) {
        const int slot = bid - 3 * kLongSlots;
        if (slot >= ntiles) return false;
        tile = (int)order[slot];
        return tile < ntiles;
    }
    // (slot, quadrant) = (bid % kLongSlots, bid / kLongSlots): the quadrant-0 blocks -- all a short tile needs -- are the first kLongSlots
    // blocks of the launch, in order, spread over the 8 XCDs like any other run of blocks (bid >> 2 / bid & 3 put every one of them
    // on XCD 0 or 4: +28 % on the uniform scene)
    const int slot = bid % kLongSlots, sub = bid / kLongSlots;
    if (slot >= ntiles) return false;
    tile = (int)order[slot];
    if (tile >= ntiles) return false;
    const uint32_t len = lengths ? lengths[tile] : ranges[tile].y - ranges[tile].x;
    if (len > thr) { lq = sub; return true; }
    return sub == 0;
}

inline int long_grid_size(int ntiles) { return ntiles + 3 * kLongSlots; }

constexpr int kChunkF = 64;          // entries staged per wave and step by the row kernel: one per lane
constexpr int kNullSlotF = kChunkF;  // the slot behind the staged ones: planes of an entry that fails the alpha test for every pixel
struct FwdRowStage {
    f32x4 a[3][kChunkF + 1];
    f32x4 tw[kChunkF + 1];     // (Tw.x Tw.y Tw.z, 1-based list position as bits)
    f32x4 q3[kChunkF + 1];     // (n.x n.y n.z r)
    f32x4 q4[kChunkF + 1];     // (g b - -)
    uint32_t idx[4][17];       // row r: the slots of its block's entries in list order, one byte each (68: the loop reads two ahead of a full list)
};

// ---- forward, long tile: one quadrant per workgroup, the list in ROUNDS of four chunks ---------------------------------------
// Wave w of the workgroup takes chunk 4 r + w of round r (64 list entries).  Per round: stage the chunk once; pass 1 over the staged
// entries = alpha only, product of (1 - alpha) per pixel -> LDS, barrier; T_in of the wave = T at the start of the round x the
// products of the waves in front of it; pass 2 over the SAME staged entries = the short path's blend from T_in; the round's weights
// and distortion sums -> LDS, barrier; every wave adds the cross terms of the distortion (its entries against the running sums of
// everything in front of them, which enter linearly) and takes T behind the round from the last wave.  The walk ends like the serial
// one, when no pixel of the quadrant takes entries any more -- a quarter-per-wave split of the WHOLE list (the first version) walked
// the list behind the saturation point as well: 3.6 x slower than the serial kernel on an opaque knot (40 k surfels on a few tiles).
enum FwdLongSlot { kLPC0, kLPC1, kLPC2, kLPD, kLPN0, kLPN1, kLPN2, kLPdist, kLPT, kLPmedd, kLPmedw, kLPlast, kLPmedc, kLPCount };
struct FwdLongPart { float v[kLPCount][64]; };            // a wave's state per pixel for the final combine (aliases the staging slices)
static_assert(sizeof(FwdLongPart) <= sizeof(FwdRowStage), "the partial states alias the staging slices");
struct FwdLongX { float P[4][64], W[4][64], d1[4][64], d2[4][64]; unsigned long long done[4]; };   // one round's exchange (single copy: the
// products are read between the round's two barriers and next written behind the second one; the other arrays the other way round)

__device__ __forceinline__ void blend_fwd_long(const BlendFwdArgs& a, int tile, int q, FwdRowStage* s_stage /*[4]*/, FwdLongX& X)
{
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_pixel(64 * q + lane, lx_, ly_);
    const int px = tx * kTileX + lx_, py = ty * kTileY + ly_;
    const bool inside = px < a.W && py < a.H;
    const float tpx = (float)(tx * kTileX), tpy = (float)(ty * kTileY);
    const float X0 = tpx + 8.0f, Y0 = tpy + 8.0f;
    const float qx = tpx + (float)(8 * (q & 1)), qy = tpy + (float)(8 * (q >> 1));
    const float qus0 = kSqrt2 * ((q & 1) ? 0.5f : -7.5f), qvs0 = kSqrt2 * ((q & 2) ? 0.5f : -7.5f);
    const float us_px = inside ? kSqrt2 * ((float)lx_ - 7.5f) : __builtin_nanf("");
    const float vs = kSqrt2 * ((float)ly_ - 7.5f);
    const uint2 range = a.ranges[tile];
    const uint32_t len = range.y - range.x;
    FwdRowStage& S = s_stage[seg];

    PixFwd st;
    pixfwd_init(st);
    float T_round = 1.0f;        // T in front of the current round (pixels that still take entries)
    float T_final = 1.0f;        // T behind this wave's last blend
    float d1_tot = 0.f, d2_tot = 0.f;   // the distortion's running sums in front of the current round (the same in all four waves)
    bool done_pix = !inside;     // saturated (forward.cu:402-406) in some wave's chunk, or outside the image

    uint32_t id_next = 64u * (uint32_t)seg + (uint32_t)lane < len ? a.point_list[range.x + 64u * seg + lane] : 0u;
    for (uint32_t base = 0; base < len; base += 4 * kChunk) {
        const uint32_t e_mine = base + 64u * (uint32_t)seg + (uint32_t)lane;
        const uint32_t id = id_next;
        id_next = e_mine + 4 * kChunk < len ? a.point_list[range.x + e_mine + 4 * kChunk] : 0u;
        // stage the wave's chunk (as blend_fwd_kernel: the entries that can touch the quadrant, compacted)
        const float4* src = a.rec + (size_t)id * kRecQuads;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3r = src[3], q4r = src[4], bx = src[5];
        const TileAffine ta = tile_affine(as_quad(q0), as_quad(q1), as_quad(q2), X0, Y0);
        const bool hit = (e_mine < len) & block_box_hit(bx, qx, qy) & block_hit_affine(ta, qus0, qus0 + 7.0f * kSqrt2, qvs0, qvs0 + 7.0f * kSqrt2);
        const unsigned long long m = ballot64(hit);
        if (hit) {
            const int slot = lane_rank(m);
            S.a[0][slot] = mk4(ta.a0.x, ta.a0.y, ta.a0.z, ta.a0.w);
            S.a[1][slot] = mk4(ta.a1.x, ta.a1.y, ta.a1.z, ta.a1.w);
            S.a[2][slot] = mk4(ta.a2.x, ta.a2.y, ta.a2.z, ta.a2.w);
            S.tw[slot] = mk4(q1.z, q1.w, q2.x, __uint_as_float(e_mine + 1u));
            S.q3[slot] = mk4(q3r);
            S.q4[slot] = mk4(q4r);
        }
        __builtin_amdgcn_wave_barrier();
        const int nhit = __builtin_popcountll(m);
        // ---- pass 1: the chunk's product of (1 - alpha) over the entries that pass the alpha and near tests
        float P = 1.0f;
        {
            f32x4 a0 = S.a[0][0], a1 = S.a[1][0], a2 = S.a[2][0], tw = S.tw[0];
            for (int i = 0; i < nhit; i++) {
                AlphaEval e;
                const bool pass = alpha_affine(us_px, vs, as_quad(a0), as_quad(a1), as_quad(a2), e);
                asm volatile("" : "+v"(e.a), "+v"(e.alpha) : : "memory");
                a0 = S.a[0][i + 1]; a1 = S.a[1][i + 1]; a2 = S.a[2][i + 1];
                bool use3d;
                const float depth = alpha_depth(e, tw.x, tw.y, tw.z, use3d);
                const float ae = (pass & (depth >= kNear)) ? e.alpha : 0.0f;
                P = P - ae * P;
                tw = S.tw[i + 1];
            }
        }
        X.P[seg][lane] = P;
        __syncthreads();
        float T_in = T_round, T_in3 = T_round;    // T in front of this wave's chunk / of the last wave's
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float Pj = X.P[j][lane];
            T_in = j < seg ? T_in * Pj : T_in;
            T_in3 *= Pj;
        }
        // ---- pass 2: the short path's blend of the same staged entries, from T_in
        float us = (!done_pix && T_in >= kTmin) ? us_px : __builtin_nanf("");   // (a pixel below 1e-4 blends nothing: test_T <= T)
        const bool took = us == us;
        const uint32_t last_before = st.last;
        st.T = T_in;
        st.dist1 = 0.f; st.dist2 = 0.f;    // the round's own sums; the sums in front of the chunk are added in below
        if (ballot64(took) != 0ull) {
            f32x4 a0 = S.a[0][0], a1 = S.a[1][0], a2 = S.a[2][0];
            f32x4 tw = S.tw[0], q3 = S.q3[0], q4 = S.q4[0];
            for (int i = 0; i < nhit; i++) {
                AlphaEval e;
                const bool pass = alpha_affine(us, vs, as_quad(a0), as_quad(a1), as_quad(a2), e);
                asm volatile("" : "+v"(e.a), "+v"(e.alpha) : : "memory");
                a0 = S.a[0][i + 1]; a1 = S.a[1][i + 1]; a2 = S.a[2][i + 1];
                bool use3d;
                const float depth = alpha_depth(e, tw.x, tw.y, tw.z, use3d);
                float w, test_T;
                pixfwd_weight(st, e.alpha, w, test_T);
                const bool ok = pass & (depth >= kNear);
                const bool blend = ok & !(test_T < kTmin);
                if (blend) {
                    st.contributor = __float_as_uint(tw.w);
                    pixfwd_accumulate<true>(st, w, test_T, depth, as_quad(q3), Quad{q4.x, q4.y, 0.f, 0.f});
                }
                us = (ok ^ blend) ? __builtin_nanf("") : us;
                asm volatile("" : "+v"(st.T), "+v"(us) : : "memory");
                tw = S.tw[i + 1]; q3 = S.q3[i + 1]; q4 = S.q4[i + 1];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float W_own = T_in - st.T, d1_own = st.dist1, d2_own = st.dist2;   // (T only moves by the blending weights)
        X.W[seg][lane] = W_own; X.d1[seg][lane] = d1_own; X.d2[seg][lane] = d2_own;
        const unsigned long long sat = ballot64(took && !(us == us));   // pixels that saturated in this chunk
        if (lane == 0) X.done[seg] = sat;
        if (st.last != last_before) T_final = st.T;
        __syncthreads();
        // distortion (forward.cu:413-417): err_i = m_i^2 A_i + dist2_i - 2 m_i dist1_i with the running sums of ALL earlier entries; the
        // chunk used its own sums, everything in front of it enters linearly
        float p1 = d1_tot, p2 = d2_tot;
        for (int j = 0; j < seg; j++) { p1 += X.d1[j][lane]; p2 += X.d2[j][lane]; }
        st.distortion += p2 * W_own - 2.0f * p1 * d1_own;
#pragma unroll
        for (int j = 0; j < 4; j++) { d1_tot += X.d1[j][lane]; d2_tot += X.d2[j][lane]; }
        const unsigned long long any_sat = X.done[0] | X.done[1] | X.done[2] | X.done[3];
        done_pix = done_pix | (((any_sat >> lane) & 1ull) != 0ull);
        T_round = T_in3 - X.W[3][lane];   // T behind the last wave's chunk (its T_in minus its weights: the operation it performed)
        if (ballot64(!done_pix && T_round >= kTmin) == 0ull) break;   // (the same verdict in all four waves: same data)
    }

    __syncthreads();   // every wave is done with its staging slice: the waves' states go there
    FwdLongPart* part = reinterpret_cast<FwdLongPart*>(s_stage);
    {
        float (*v)[64] = part[seg].v;
        v[kLPC0][lane] = st.C[0]; v[kLPC1][lane] = st.C[1]; v[kLPC2][lane] = st.C[2]; v[kLPD][lane] = st.D;
        v[kLPN0][lane] = st.N[0]; v[kLPN1][lane] = st.N[1]; v[kLPN2][lane] = st.N[2];
        v[kLPdist][lane] = st.distortion; v[kLPT][lane] = T_final; v[kLPmedd][lane] = st.med_d; v[kLPmedw][lane] = st.med_w;
        v[kLPlast][lane] = __uint_as_float(st.last); v[kLPmedc][lane] = __uint_as_float(st.med_c);
    }
    __syncthreads();
    if (seg != 0) return;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, dist = 0.f, T = 1.0f, medd = 0.f, medw = 0.f;
    uint32_t last = 0u, medc = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float (*v)[64] = part[k].v;
        C0 += v[kLPC0][lane]; C1 += v[kLPC1][lane]; C2 += v[kLPC2][lane]; D += v[kLPD][lane];
        N0 += v[kLPN0][lane]; N1 += v[kLPN1][lane]; N2 += v[kLPN2][lane]; dist += v[kLPdist][lane];
        const uint32_t lk = __float_as_uint(v[kLPlast][lane]), mk = __float_as_uint(v[kLPmedc][lane]);
        if (lk > last) { last = lk; T = v[kLPT][lane]; }                  // the wave that blended the pixel's last contributor holds its final T
        if (mk > medc) { medc = mk; medd = v[kLPmedd][lane]; medw = v[kLPmedw][lane]; }
    }
    uint32_t mx = inside ? last : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) atomicMax(&a.tile_last[tile], mx);
    const size_t plane = (size_t)ntiles * kTilePix;
    const size_t slot = (size_t)tile * kTilePix + 64 * q + lane;
    a.final_T[slot] = T;
    a.final_T[plane + slot] = d1_tot;
    a.final_T[2 * plane + slot] = d2_tot;
    a.n_contrib[slot] = last;
    a.n_contrib[plane + slot] = medc;
    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)py * a.W + px;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[HW + pix] = C1 + T * a.bg[1];
        a.out_color[2 * HW + pix] = C2 + T * a.bg[2];
        a.out_others[pix] = D;
        a.out_others[HW + pix] = 1.f - T;
        a.out_others[2 * HW + pix] = N0;
        a.out_others[3 * HW + pix] = N1;
        a.out_others[4 * HW + pix] = N2;
        a.out_others[5 * HW + pix] = medd;
        a.out_others[6 * HW + pix] = dist;
        a.out_others[7 * HW + pix] = medw;
    }
}

__global__ void __launch_bounds__(kTilePix, DGS_FWD_MINWAVES) blend_fwd_rows_kernel(BlendFwdArgs a)
{
    __shared__ FwdRowStage s_stage[4];
    __shared__ uint32_t s_max[4];
    __shared__ FwdLongX s_x;      // long tiles: the rounds' exchange

    if (a.clean_flag && blockIdx.x == 0 && threadIdx.x == 0) *a.clean_flag = a.clear_blocks > 0 ? 1u : 0u;   // read after this launch has ended
    if ((int)blockIdx.x < a.clear_blocks) {   // accumulator rider (BlendFwdArgs)
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t i = blockIdx.x * kTilePix + threadIdx.x; i < a.clear_n4; i += (uint32_t)a.clear_blocks * kTilePix) a.clear[i] = z;
        return;
    }
    const int bid = (int)blockIdx.x - a.clear_blocks;
    const int ntiles = a.tiles_x * a.tiles_y;
    int tile;
    if (a.long_thr) {   // tile order 3 with the long-tile path: see long_decode
        int lq;
        if (!long_decode(bid, a.order, ntiles, a.long_thr->fwd, nullptr, a.ranges, tile, lq)) return;
        if (lq >= 0) {
            blend_fwd_long(a, tile, lq, s_stage, s_x);
            return;
        }
    } else {
        tile = tile_for_block(bid, a.tiles_x, a.tiles_y, a.mode);
        if (a.mode < 3 && tile >= ntiles) return;
        if (a.mode >= 3) tile = (int)a.order[tile];
        if (tile >= ntiles) return;   // mode 4: empty slot
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_pixel(tid, lx_, ly_);
    const int px = tx * kTileX + lx_, py = ty * kTileY + ly_;
    const bool inside = px < a.W && py < a.H;
    const float tpx = (float)(tx * kTileX), tpy = (float)(ty * kTileY);
    const float X0 = tpx + 8.0f, Y0 = tpy + 8.0f;
    const float qx = tpx + (float)(8 * (wave & 1)), qy = tpy + (float)(8 * (wave >> 1));
    const float qus0 = kSqrt2 * ((wave & 1) ? 0.5f : -7.5f), qvs0 = kSqrt2 * ((wave & 2) ? 0.5f : -7.5f);
    float us = inside ? kSqrt2 * ((float)lx_ - 7.5f) : __builtin_nanf("");   // NaN = this pixel takes no further entry
    const float vs = kSqrt2 * ((float)ly_ - 7.5f);

    const uint2 range = a.ranges[tile];
    const uint32_t len = range.y - range.x;
    FwdRowStage& S = s_stage[wave];
    if (lane < 6) {   // the null slot: opacity 0 (alpha = 0 * G fails the test, also for G = NaN), everything else finite
        f32x4* planes[6] = {&S.a[0][kNullSlotF], &S.a[1][kNullSlotF], &S.a[2][kNullSlotF], &S.tw[kNullSlotF], &S.q3[kNullSlotF], &S.q4[kNullSlotF]};
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (lane == k) *planes[k] = mk4(0.f, 0.f, 0.f, 0.f);
    }
    const uint8_t* my_list = (const uint8_t*)&S.idx[row][0];

    PixFwd st;
    pixfwd_init(st);

    auto visit = [&](auto track_median, int niter) {
        int sl = my_list[0];
        int sl_next = my_list[1];
        f32x4 a0 = S.a[0][sl], a1 = S.a[1][sl], a2 = S.a[2][sl];
        f32x4 tw = S.tw[sl], q3 = S.q3[sl], q4 = S.q4[sl];
        for (int i = 0; i < niter; i++) {
            AlphaEval e;
            const bool pass = alpha_affine(us, vs, as_quad(a0), as_quad(a1), as_quad(a2), e);
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(e.a), "+v"(e.alpha) : : "memory");   // see blend_fwd_rows_kernel: one register set, loads pinned behind their last use
#endif
            sl = sl_next;
            a0 = S.a[0][sl]; a1 = S.a[1][sl]; a2 = S.a[2][sl];
            bool use3d;
            const float depth = alpha_depth(e, tw.x, tw.y, tw.z, use3d);
            float w, test_T;
            pixfwd_weight(st, e.alpha, w, test_T);
            const bool ok = pass & (depth >= kNear);      // forward.cu:388
            const bool blend = ok & !(test_T < kTmin);
            if (blend) {
                st.contributor = __float_as_uint(tw.w);   // 1-based list position (forward.cu:356)
                pixfwd_accumulate<decltype(track_median)::value>(st, w, test_T, depth, as_quad(q3), Quad{q4.x, q4.y, 0.f, 0.f});
            }
            us = (ok ^ blend) ? __builtin_nanf("") : us;   // passed but saturated: the pixel is finished
#if DGS_PIN_PREFETCH
            asm volatile("" : "+v"(st.T), "+v"(us) : : "memory");
#endif
            tw = S.tw[sl]; q3 = S.q3[sl]; q4 = S.q4[sl];
            sl_next = my_list[i + 2];
            asm volatile("" : "+v"(sl_next));   // requested here, a visit before the address is formed from it
        }
    };

    uint32_t id_next = (uint32_t)lane < len ? a.point_list[range.x + lane] : 0u;
    unsigned long long alive = ballot64(inside);   // lanes that still take entries
    for (uint32_t base = 0; base < len && alive != 0ull; base += kChunkF) {
        const uint32_t e_mine = base + (uint32_t)lane;
        const uint32_t id = id_next;   // (lanes beyond the end of the list hold id 0: a valid record, masked out below)
        const float4* src = a.rec + (size_t)id * kRecQuads;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4], bx = src[5];
        id_next = e_mine + kChunkF < len ? a.point_list[range.x + e_mine + kChunkF] : 0u;
        const TileAffine ta = tile_affine(as_quad(q0), as_quad(q1), as_quad(q2), X0, Y0);
        uint32_t bm = e_mine < len ? blocks_hit_linear(ta, qus0, qvs0, as_quad(bx), qx, qy) : 0u;
        // a block whose pixels are all finished takes no more entries
        bm &= (((uint32_t)alive & 0xffffu) ? 1u : 0u) | (((uint32_t)(alive >> 16) & 0xffffu) ? 2u : 0u) |
              (((uint32_t)(alive >> 32) & 0xffffu) ? 4u : 0u) | (((uint32_t)(alive >> 48) & 0xffffu) ? 8u : 0u);
        const bool hit = bm != 0u;
        const unsigned long long m = ballot64(hit);
        if (m == 0ull) continue;
        const unsigned long long m0 = ballot64((bm & 1u) != 0u), m1 = ballot64((bm & 2u) != 0u), m2 = ballot64((bm & 4u) != 0u), m3 = ballot64((bm & 8u) != 0u);
        ((uint32_t*)&S.idx[0][0])[lane] = 0x01010101u * (uint32_t)kNullSlotF;   // every list: null slots behind its entries
        if (lane < 4) ((uint32_t*)&S.idx[0][0])[64 + lane] = 0x01010101u * (uint32_t)kNullSlotF;
        if (hit) {
            const int slot = lane_rank(m);
            S.a[0][slot] = mk4(ta.a0.x, ta.a0.y, ta.a0.z, ta.a0.w);
            S.a[1][slot] = mk4(ta.a1.x, ta.a1.y, ta.a1.z, ta.a1.w);
            S.a[2][slot] = mk4(ta.a2.x, ta.a2.y, ta.a2.z, ta.a2.w);
            S.tw[slot] = mk4(q1.z, q1.w, q2.x, __uint_as_float(e_mine + 1u));
            S.q3[slot] = mk4(q3);
            S.q4[slot] = mk4(q4);
            uint8_t* lists = (uint8_t*)&S.idx[0][0];
            if (bm & 1u) lists[lane_rank(m0)] = (uint8_t)slot;
            if (bm & 2u) lists[68 + lane_rank(m1)] = (uint8_t)slot;
            if (bm & 4u) lists[136 + lane_rank(m2)] = (uint8_t)slot;
            if (bm & 8u) lists[204 + lane_rank(m3)] = (uint8_t)slot;
        }
        __builtin_amdgcn_wave_barrier();   // the slice is private to this wave: its LDS writes above are ordered before its reads below
        const int n01 = max(__builtin_popcountll(m0), __builtin_popcountll(m1)), n23 = max(__builtin_popcountll(m2), __builtin_popcountll(m3));
        const int niter = __builtin_amdgcn_readfirstlane(max(n01, n23));
        // median bookkeeping (forward.cu:421-425) only while some pixel of the wave still has T > 0.5
        if (ballot64(st.T > 0.5f && us == us) != 0ull) visit(std::true_type{}, niter);
        else visit(std::false_type{}, niter);
        __builtin_amdgcn_wave_barrier();
        alive = ballot64(us == us);   // wave-level early out (forward.cu:334-336 votes per block)
    }

    uint32_t mx = inside ? st.last : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) s_max[wave] = mx;
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

struct BlendBwdArgs {
    const uint2* ranges;
    const uint32_t* order;
    const uint32_t* point_list;
    const float4* rec;
    int W, H, tiles_x, tiles_y, mode;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const uint32_t* tile_last;
    const float* dL_dpix;     // [3,H,W]
    const float* dL_dothers;  // [8,H,W]
    float* acc;               // [P, kAccFloats], zeroed
    float* det_part;          // deterministic variant 1 only: [num_rendered][4 waves][kAccFloats], zeroed (see det_reduce_kernel)
    unsigned long long* acc64; // deterministic variant 2 only: [P, kAccFloats] fixed-point sums (2^-44), zero on entry (fixed_to_acc_kernel)
    uint32_t* clean_flag;      // BlendFwdArgs::clean_flag: the accumulator rows are written from here on (or null)
    const LongThr* long_thr;  // thresholds of the long-tile path, or null (path off: grid = blend_grid_size)
};

// ---- wave reduction of the 16 per-surfel partials of one (wave, entry) visit --------------------------------------------
// Transposition through the wave's own LDS (wave_reduce.h: 16 ds_write_addtid_b32 + 4 ds_read_b128 + 17 VALU), in two rounds of
// 8 values through 2 KB per wave.  The four earlier implementations (butterfly on the LDS crossbar 0.338 ms, matrix pipe 0.548,
// permlane + MFMA hybrid 0.455, permlane + DPP 0.304-0.315; this one 0.267) and their measurements: ab/reduce_variants.h.
// On return lane l with (l & 3) == 0 holds the wave total of v[(l >> 3) + 2 (l & 4)]; reduce16_slot() says so.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef DGS_RED_PHASES
#define DGS_RED_PHASES 2     // 1 (A/B): one round of 16 values through 4 KB -- 3 workgroups per CU instead of 5: 0.304 ms
#endif
#if !defined(DGS_AB_BUILD) && DGS_RED_PHASES != 2
#error "DGS_RED_PHASES is an A/B switch: -DDGS_AB_BUILD"
#endif

#if DGS_BWD_REDUCE != 4
#include "ab/reduce_variants.h"   // defines BwdRedCtx, wave_reduce16, reduce16_slot for variants 0-3
#else
struct BwdRed { float v[16 / DGS_RED_PHASES][64]; };   // one wave's transposition buffer
typedef RedLds<DGS_RED_PHASES> BwdRedCtx;

__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane, const BwdRedCtx& rc)
{
#if DGS_RED_PHASES == 1
    return wave_reduce16_lds(v, rc);
#else
    return wave_reduce16_lds(v, rc, lane);
#endif
}

// which of the 16 values lane `lane` holds after wave_reduce16, or -1 if the lane holds a duplicate
__device__ __forceinline__ int reduce16_slot(int lane)
{
#if DGS_RED_PHASES == 2
    return (lane & 3) == 0 ? (lane >> 3) + 2 * (lane & 4) : -1;
#else
    return (lane & 3) == 0 ? (lane >> 2) : -1;
#endif
}
#endif

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

constexpr int kDetRows = DGS_BWD_ROWS ? 16 : 4;   // rows of det_part per list entry: one per (wave, 16-lane row) or per wave

#ifndef DGS_BWD_MINWAVES
#define DGS_BWD_MINWAVES 5
#endif
// DET = 0: the per-(wave, entry) sums go into the surfel's accumulator row with hardware float atomics (order of arrival:
// results differ at the rounding level from run to run, like the reference's).  DET = 1 (dgs_set_option(7, 1), tests): every
// (list entry, wave) owns a row of det_part and stores its sums there; det_reduce_kernel adds the rows of a surfel in a fixed order.
// DET = 2 (dgs_set_option(7, 2), round 5): the sums are added as 64-bit FIXED-POINT numbers (units of 2^-44, |sum| < 2^18) with
// integer atomics -- integer addition is associative, so the result does not depend on the order of arrival: bit-identical
// gradients from run to run at the speed of the default kernel, legal inside a captured graph (variant 1 is neither: it allocates
// per call and searches lists).  The price is the fixed quantum: a partial sum below 6e-14 is lost and one of 1e-9 keeps 4-5
// digits, where fp32 keeps 7 -- a different (slightly noisier) optimisation than the default, which is why it is an option:
// reproducible pre-fits (bench.py --workload trained, fit(deterministic=True)) and tests.
__device__ __forceinline__ unsigned long long to_fixed44(float v)
{
    v = fminf(fmaxf(v, -262144.0f), 262144.0f);
    return (unsigned long long)__float2ll_rn(v * 17592186044416.0f);   // 2^44; two's complement: unsigned wrap-around adds signed numbers
}

#ifdef DGS_COUNT_VISITS   // development build (tools/diag/visit_counts.py): what the backward's visits are made of
__device__ unsigned long long g_visit_counts[4];   // visits of the real pass | of them with no blending lane | blending lanes of the others | staged entries
#endif
#ifndef DGS_BWD_CHUNK
#define DGS_BWD_CHUNK 48
#endif
constexpr int kChunkB = DGS_BWD_CHUNK;   // list entries the backward stages per wave and step (lanes >= kChunkB stage nothing): sizes the slice
struct BwdStage {            // one wave's staging slice: the chunk's visited entries, compacted (+1: the visit loop reads one slot ahead)
    f32x4 a[3][kChunkB + 1];  // alpha part of the entry's affine image (tile_affine)
    f32x4 tw[kChunkB + 1];    // (Tw.x Tw.y Tw.z opacity)
    f32x4 tuv[kChunkB + 1];   // (Tu.x Tu.y Tv.x Tv.y): k.xy, l.xy of a pixel are rebuilt from them
    f32x4 q3[kChunkB + 1];    // (n.x n.y n.z r)
    f32x4 q4[kChunkB + 1];    // (g b, 0-based list index as bits, surfel id as bits)
};

// One quadrant of a tile, back to front.  LONG = false: the whole list behind the quadrant's last contributor, one wave (the kernel's
// normal path, q = wave).  LONG = true ("long tiles" above blend_fwd_rows_kernel): the workgroup is ONE quadrant of a long tile and its
// wave `seg` takes the seg-th quarter of that range.  Both recurrences of the backward are linear in the state that enters a quarter --
// T <- T / (1 - alpha) and acc <- (1 - alpha) acc + alpha u, with alpha and u functions of the entry and the pixel only -- so pass 1
// walks the quarter once without the gradient arithmetic and leaves (F, A, B) per pixel: T_out = F T_in, acc_out = A acc_in + B; every
// wave then composes the quarters behind its own (they were walked by the other waves at the same time) and runs the real pass from
// the state that reaches it.  1.4 x the arithmetic on a quarter of the critical path; the gradients differ from the serial walk in
// rounding only (products and sums taken quarter-wise), like two runs of the atomic backward do.
template <int DET, bool LONG>
__device__ __forceinline__ void bwd_quadrant(const BlendBwdArgs& a, int tile, int q, int seg, BwdStage& S, const BwdRedCtx& rc, float* xfer /*LONG: [4][3][64]*/)
{
    const int ntiles = a.tiles_x * a.tiles_y;
    const int lane = threadIdx.x & 63;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int lx_, ly_;
    lane_pixel(64 * q + lane, lx_, ly_);
    const int px = tx * kTileX + lx_, py = ty * kTileY + ly_;
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
    const float tpx = (float)(tx * kTileX), tpy = (float)(ty * kTileY);
    const float X0 = tpx + 8.0f, Y0 = tpy + 8.0f;
    const float qx = tpx + (float)(8 * (q & 1)), qy = tpy + (float)(8 * (q >> 1));
    const float qus0 = kSqrt2 * ((q & 1) ? 0.5f : -7.5f), qvs0 = kSqrt2 * ((q & 2) ? 0.5f : -7.5f);
    const float us = kSqrt2 * ((float)lx_ - 7.5f), vs = kSqrt2 * ((float)ly_ - 7.5f);
    const uint2 range = a.ranges[tile];

    const size_t plane = (size_t)ntiles * kTilePix;
    const size_t slot_px = (size_t)tile * kTilePix + 64 * q + lane;
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
        // A pixel nothing contributed to never uses its gradient in the reference (backward.cu:283-300: the loop over its contributors is
        // empty), whatever that gradient is -- and PyTorch hands such pixels NaN: the reference's own render() divides the depth map by an
        // alpha of 0 there (gaussian_renderer/__init__.py:186-187; 0 / 0 in the division's backward).  Here the lanes of a visit multiply
        // before they mask, so the value must not be read at all.
        if (last == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) gpix[c] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c++) goth[c] = 0.f;
        }
        pixbwd_init_affine(st, inside ? a.final_T[slot_px] : 0.f, a.final_T[plane + slot_px], a.final_T[2 * plane + slot_px], last, medc, gpix,
                           goth, a.bg);
    }
    // highest entry any lane of this wave needs: the wave walks the list back to front from there (backward.cu:276-279 skips the
    // entries behind a pixel's last contributor one by one)
    int wave_last = st.last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int o = __shfl_xor(wave_last, d, 64);
        wave_last = o > wave_last ? o : wave_last;
    }
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);  // uniform by construction
    const int rslot = reduce16_slot(lane);
    // the range of list entries this wave walks: [lo, hi), back to front
    const int lo = LONG ? (int)(((long long)wave_last * seg) >> 2) : 0, hi = LONG ? (int)(((long long)wave_last * (seg + 1)) >> 2) : wave_last;
    float xF = 1.0f, xA = 1.0f, xB = 0.0f;   // pass 1 (LONG): the quarter's transfer of (T, acc)

    // chunk lane l holds list entry e = top - l; the entries that can touch the quadrant are compacted in that (back to front) order
    const bool stager = kChunkB == 64 || lane < kChunkB;
#ifdef DGS_COUNT_VISITS
    unsigned long long n_visit = 0, n_empty = 0, n_lanes = 0, n_staged = 0;
#endif
    auto walk = [&](auto light_tag) {
        constexpr bool kLight = decltype(light_tag)::value;
        uint32_t id_next = stager && hi - 1 - lane >= lo ? a.point_list[range.x + (uint32_t)(hi - 1 - lane)] : 0u;
        for (int top = hi - 1; top >= lo; top -= kChunkB) {
            const int e_mine = top - lane;
            const uint32_t id = id_next;   // (lanes beyond the front of the range hold id 0: a valid record, masked out below)
            // straight-line staging, see blend_fwd_rows_kernel
            const float4* src = a.rec + (size_t)id * kRecQuads;
            const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3s = src[3], q4s = src[4], bx = src[5];
            id_next = stager && e_mine - kChunkB >= lo ? a.point_list[range.x + (uint32_t)(e_mine - kChunkB)] : 0u;
            const TileAffine ta = tile_affine(as_quad(q0), as_quad(q1), as_quad(q2), X0, Y0);
            const bool hit = stager & (e_mine >= lo) & block_box_hit(bx, qx, qy) & block_hit_affine(ta, qus0, qus0 + 7.0f * kSqrt2, qvs0, qvs0 + 7.0f * kSqrt2);
            const unsigned long long m = ballot64(hit);
            if (m == 0ull) continue;
            if (hit) {
                const int slot = lane_rank(m);
                S.a[0][slot] = mk4(ta.a0.x, ta.a0.y, ta.a0.z, ta.a0.w);
                S.a[1][slot] = mk4(ta.a1.x, ta.a1.y, ta.a1.z, ta.a1.w);
                S.a[2][slot] = mk4(ta.a2.x, ta.a2.y, ta.a2.z, ta.a2.w);
                S.tw[slot] = mk4(q1.z, q1.w, q2.x, q2.w);
                S.tuv[slot] = mk4(q0.x, q0.y, q0.w, q1.x);
                S.q3[slot] = mk4(q3s);
                S.q4[slot] = mk4(q4s.x, q4s.y, __int_as_float(e_mine), __uint_as_float(id));
            }
            __builtin_amdgcn_wave_barrier();   // the slice is private to this wave: its LDS writes above are ordered before its reads below
            const int nhit = __builtin_popcountll(m);
#ifdef DGS_COUNT_VISITS
            if (!kLight) n_staged += (unsigned long long)__builtin_popcountll(ballot64(stager & (e_mine >= lo)));
#endif
            f32x4 a0 = S.a[0][0], a1 = S.a[1][0], a2 = S.a[2][0];
            f32x4 tw = S.tw[0], tuv = S.tuv[0], q3 = S.q3[0], q4 = S.q4[0];
            for (int i = 0; i < nhit; i++) {
                AlphaEval ev;
                bool ok = alpha_affine(us, vs, as_quad(a0), as_quad(a1), as_quad(a2), ev);
#if DGS_PIN_PREFETCH
                asm volatile("" : "+v"(ev.a), "+v"(ev.alpha) : : "memory");   // see blend_fwd_rows_kernel: one register set, loads pinned behind their last use
#endif
                a0 = S.a[0][i + 1]; a1 = S.a[1][i + 1]; a2 = S.a[2][i + 1];
                DGS_PIN4(a0); DGS_PIN4(a1); DGS_PIN4(a2);
                const int e = __builtin_amdgcn_readfirstlane(__float_as_int(q4.z));  // 0-based list index == the reference's `contributor`
                ok = ok & (e < st.last_contributor);
#ifdef DGS_COUNT_VISITS
                if (!kLight) { n_visit++; if (ballot64(ok) == 0ull) n_empty++; }
#endif
                if (ballot64(ok) != 0ull) {
                    bool use3d;
                    const float depth = alpha_depth(ev, tw.x, tw.y, tw.z, use3d);
                    ok = ok & (depth >= kNear);
#ifdef DGS_COUNT_VISITS
                    if (!kLight) n_lanes += (unsigned long long)__builtin_popcountll(ballot64(ok));
#endif
                    if (kLight) {
                        // pass 1 of a long tile: only what the two recurrences need (pixbwd_step_affine's alpha, 1 / (1 - alpha) and u)
                        const float alpha = ok ? ev.alpha : 0.f;
                        const float u = pixbwd_u_affine(st, ok, depth, e, as_quad(q3), Quad{q4.x, q4.y, 0.f, 0.f});
                        xF = xF * fast_rcp(1.f - alpha);
                        xA = xA - alpha * xA;
                        xB = xB + alpha * (u - xB);
                    } else {
                    // every lane runs the step; a lane that does not blend the entry contributes exact zeros (pixbwd_step_affine)
                    float out[16], out2d[2];
#if DGS_DIAG_BWD == 3
                    for (int k = 0; k < 16; k++) out[k] = depth;
                    out2d[0] = out2d[1] = 0.f;
                    asm volatile("" : : "v"(depth), "v"(tuv.x), "v"(q3.x), "v"(q4.x));
#else
                    pixbwd_step_affine(st, ev, ok, use3d, depth, e, pfx, pfy, tw.x, tw.y, as_quad(tuv), tw.w, as_quad(q3),
                                       Quad{q4.x, q4.y, 0.f, 0.f}, out, out2d);
#endif
                    // (two ballots of plain comparisons and scalar logic: ballot64(ok && !use3d) compiled to a select and a compare per visit)
                    const bool any2d = (ballot64(ok) & ~ballot64(use3d)) != 0ull;
                    // wave-uniform row address: keep it on the scalar unit (SGPR base + per-lane offset in the atomic)
                    const size_t row = DET == 1 ? ((size_t)(range.x + (uint32_t)e) * kDetRows + q) * kAccFloats
                                                : (size_t)__builtin_amdgcn_readfirstlane(__float_as_uint(q4.w)) * kAccFloats;
                    float* dst = (DET == 1 ? a.det_part : a.acc) + row;
                    unsigned long long* dst64 = DET == 2 ? a.acc64 + row : nullptr;
#if DGS_DIAG_BWD >= 2
                    float tot = 0.f;
                    for (int k = 0; k < 16; k++) asm volatile("" : : "v"(out[k]));
                    asm volatile("" : : "s"(dst));
#else
                    const float tot = wave_reduce16(out, lane, rc);
#endif
#if DGS_DIAG_BWD == 0
                    if (rslot >= 0) {
                        if (DET == 1) dst[rslot] = tot;
                        else if (DET == 2) atomicAdd(dst64 + rslot, to_fixed44(tot));
                        else atomicAdd(dst + rslot, tot);
                    }
#else
                    asm volatile("" : : "v"(tot));
#endif
                    if (any2d) {  // rare 2-D filter branch (backward.cu:436-443)
                        const float mx = wave_sum(out2d[0]);
                        const float my = wave_sum(out2d[1]);
                        if (lane == 0) {
                            if (DET == 1) { dst[kAccMean2D] = mx; dst[kAccMean2D + 1] = my; }
                            else if (DET == 2) { atomicAdd(dst64 + kAccMean2D, to_fixed44(mx)); atomicAdd(dst64 + kAccMean2D + 1, to_fixed44(my)); }
                            else { atomicAdd(dst + kAccMean2D, mx); atomicAdd(dst + kAccMean2D + 1, my); }
                        }
                    }
                    }
                }
#if DGS_PIN_PREFETCH
                asm volatile("" : "+v"(st.T) : : "memory");
#endif
                tw = S.tw[i + 1]; tuv = S.tuv[i + 1]; q3 = S.q3[i + 1]; q4 = S.q4[i + 1];
                DGS_PIN4(tw); DGS_PIN4(tuv); DGS_PIN4(q3); DGS_PIN4(q4);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    if (LONG) {
        walk(std::true_type{});
        xfer[(seg * 3 + 0) * 64 + lane] = xF;
        xfer[(seg * 3 + 1) * 64 + lane] = xA;
        xfer[(seg * 3 + 2) * 64 + lane] = xB;
        __syncthreads();
        // the state that reaches this quarter: (T_final, 0) through the quarters behind it, back to front
        float T = st.T, acc = 0.0f;
        for (int j = 3; j > seg; j--) {
            T = T * xfer[(j * 3 + 0) * 64 + lane];
            acc = xfer[(j * 3 + 1) * 64 + lane] * acc + xfer[(j * 3 + 2) * 64 + lane];
        }
        st.T = T;
        st.acc = acc;
        __syncthreads();   // xfer lives in the reduction buffers, which the real pass writes
    }
    walk(std::false_type{});
#ifdef DGS_COUNT_VISITS
    if (lane == 0) {
        atomicAdd(&g_visit_counts[0], n_visit); atomicAdd(&g_visit_counts[1], n_empty);
        atomicAdd(&g_visit_counts[2], n_lanes); atomicAdd(&g_visit_counts[3], n_staged);
    }
#endif
}

template <int DET>
__global__ void __launch_bounds__(kTilePix, DGS_BWD_MINWAVES) blend_bwd_kernel(BlendBwdArgs a)  // workgroups per CU = waves per SIMD the register allocator must allow
{
    __shared__ BwdStage s_stage[4];
#if DGS_BWD_REDUCE == 4
    __shared__ BwdRed s_red[4];
    static_assert(sizeof(BwdRed) * 4 >= 4 * 3 * 64 * sizeof(float), "the long path's transfer table lives in the reduction buffers");
#endif

    const int ntiles = a.tiles_x * a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.clean_flag && blockIdx.x == 0 && threadIdx.x == 0) *a.clean_flag = 0u;   // (prep_bwd_kernel read it in the launch before)
    int tile, lq = -1;
#if DGS_BWD_REDUCE == 4
    if (a.long_thr) {   // tile order 3 with the long-tile path (long_decode)
        if (!long_decode(blockIdx.x, a.order, ntiles, a.long_thr->bwd, a.tile_last, a.ranges, tile, lq)) return;
    } else
#endif
    {
        tile = tile_for_block(blockIdx.x, a.tiles_x, a.tiles_y, a.mode);
        if (a.mode < 3 && tile >= ntiles) return;
        if (a.mode >= 3) tile = (int)a.order[tile];
        if (tile >= ntiles) return;   // mode 4: empty slot
    }
    if (a.tile_last[tile] == 0u) return;   // no pixel of the tile blended anything
    BwdRedCtx rc;
#if DGS_BWD_REDUCE == 4
    rc.init(&s_red[wave].v[0][0], lane);
    if (lq >= 0) {
        bwd_quadrant<DET, true>(a, tile, lq, wave, s_stage[wave], rc, &s_red[0].v[0][0]);
        return;
    }
#endif
    bwd_quadrant<DET, false>(a, tile, wave, 0, s_stage[wave], rc, nullptr);
}

#if DGS_BWD_ROWS
#include "ab/blend_bwd_rows.h"
#endif

// Deterministic variant 2: the fixed-point rows -> the float accumulator rows the per-surfel kernel reads; the fixed-point rows are
// left zeroed for the next backward.
__global__ void __launch_bounds__(256) fixed_to_acc_kernel(unsigned long long* __restrict__ acc64, float* __restrict__ acc, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long v = (long long)acc64[i];
    acc64[i] = 0ull;
    acc[i] = (float)((double)v * (1.0 / 17592186044416.0));
}

// Deterministic reduction of the backward blend (test option): one thread per surfel walks the tiles of its rectangle in
// row-major order, finds its entry in the tile's sorted list and adds the four waves' rows in order 0..3.  Same sums as the
// atomics in ONE fixed order: two runs give bit-identical gradients.  Slow by design (linear search of the lists).
__global__ void __launch_bounds__(256) det_reduce_kernel(int P, const int* radii, const uint2* rects, int tiles_x, const uint2* ranges,
                                                         const uint32_t* point_list, const float* part, float* acc)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P || !(radii[idx] > 0)) return;
    const uint2 r = rects[idx];
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    float sum[kAccFloats];
#pragma unroll
    for (int k = 0; k < kAccFloats; k++) sum[k] = 0.f;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const uint2 rg = ranges[y * tiles_x + x];
            for (uint32_t pos = rg.x; pos < rg.y; pos++) {
                if (point_list[pos] != (uint32_t)idx) continue;
                for (int w = 0; w < kDetRows; w++) {
                    const float* row = part + ((size_t)pos * kDetRows + w) * kAccFloats;
#pragma unroll
                    for (int k = 0; k < kAccFloats; k++) sum[k] += row[k];
                }
                break;
            }
        }
#pragma unroll
    for (int k = 0; k < kAccFloats; k++) acc[(size_t)idx * kAccFloats + k] = sum[k];
}

}  // namespace dgs
