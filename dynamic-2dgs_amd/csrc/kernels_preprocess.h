// kernels_preprocess.h -- per-surfel and binning kernels (gfx950): forward preprocess + per-tile counts, tile scan,
// key scatter, per-tile sort, fused per-surfel backward, frustum mark.
// Replaces preprocessCUDA / cub InclusiveSum / duplicateWithKeys / cub DeviceRadixSort / identifyTileRanges /
// computeAABB-bwd / preprocessCUDA-bwd / checkFrustum of the reference (forward.cu:166-260,
// rasterizer_impl.cu:54-138,278,304-319, backward.cu:533-649).
//
// Binning is a TILE-BUCKETED sort instead of the reference's global 44..46-bit LSD radix sort over
// (tile << 32 | depth) keys: count entries per tile (atomics), scan the T counts, scatter (depth, index) keys into
// the tile's bucket, then sort every bucket in LDS with one workgroup.  The global sort moves 12 B x R x ~12 through
// HBM; here every key is written once and read once, and tile ranges fall out of the scan.  The result is the same
// list: buckets ordered by tile, entries by (depth bits, surfel index) = the stable radix order.
#pragma once
#include <hip/hip_runtime.h>

#include "surfel_math.h"

namespace dgs {

constexpr int kSurfelBlock = 64;    // one wave per workgroup: 12.25 KB of LDS for its SH rows -> 13 workgroups per CU = 832 surfels, so that the 782
                                     // surfels per CU of a 200 k cloud are resident in ONE round (256-thread workgroups: 3 per CU = 768, a second round for 14)

struct PreprocessArgs {
    int P, D, M;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* colors_precomp;
    Camera cam;
    int* radii;             // [P] out
    float4* rec;            // [P*5] out
    uint2* rects;           // [P] out: packed (tight) tile rectangle of every surfel
    int tight;              // 1: exact opacity-aware rectangles (default), 0: the reference's rectangles
    unsigned row_inv;       // ceil(2^32 / (3 M)) (see surfel_bwd_kernel)
};

// The SH rows of a workgroup's surfels (`total` = rows x row floats, contiguous from `src`) -> LDS rows of `stride` floats.
// 16 bytes per lane and load: at degree 3 the 64 rows are 768 float4, i.e. ONE batch of 12 loads per thread, all in flight
// before the first LDS store -- the 4-byte version was four dependent batches of 12 (a 4-byte load keeps the CU's address unit busy as
// long as a 16-byte one), a third of both per-surfel kernels' run time.  The 4-byte path stays for row counts / alignments
// that do not divide.
__device__ __forceinline__ void stage_sh_rows(const float* __restrict__ src, int total, int row, int stride, unsigned row_inv, float* __restrict__ s_sh)
{
    if ((row & 3) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0) {   // (a float4 never straddles two rows: one row / column per load)
        const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
        const int nvec = total >> 2;
        for (int v0 = threadIdx.x; v0 < nvec; v0 += 12 * kSurfelBlock) {
            float4 q[12];
#pragma unroll
            for (int i = 0; i < 12; i++) { const int v = v0 + i * kSurfelBlock; q[i] = v < nvec ? src4[v] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int v = v0 + i * kSurfelBlock;
                if (v < nvec) {
                    const int e = 4 * v, r = (int)__umulhi((unsigned)e, row_inv);
                    float* d = s_sh + (r * stride + (e - r * row));
                    d[0] = q[i].x; d[1] = q[i].y; d[2] = q[i].z; d[3] = q[i].w;
                }
            }
        }
        return;
    }
    // batches of 12 elements per thread: all loads of a batch are in flight before the first LDS store (an
    // element-wise copy loop is a chain of dependent global-load latencies)
    for (int e0 = threadIdx.x; e0 < total; e0 += 12 * kSurfelBlock) {
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; i++) { const int e = e0 + i * kSurfelBlock; v[i] = e < total ? src[e] : 0.f; }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int e = e0 + i * kSurfelBlock;
            if (e < total) { const int r = (int)__umulhi((unsigned)e, row_inv); s_sh[r * stride + (e - r * row)] = v[i]; }
        }
    }
}

__global__ void __launch_bounds__(kSurfelBlock) preprocess_fwd_kernel(PreprocessArgs a)
{
    // SH rows reach the threads through LDS with coalesced loads (see surfel_bwd_kernel)
    extern __shared__ float s_sh[];
    const int base = blockIdx.x * kSurfelBlock;
    const int idx = base + threadIdx.x;
    const int row = a.M * 3, stride = row + 1;
    const bool use_sh = !a.colors_precomp && a.shs;
    // the surfel's own parameters are requested in front of the staging (one memory round trip for both instead of two)
    const int li = min(idx, a.P - 1);
    float pos[3] = {a.means3D[3 * li], a.means3D[3 * li + 1], a.means3D[3 * li + 2]};
    float sc[2] = {a.scales[2 * li], a.scales[2 * li + 1]};
    const float4 qv = reinterpret_cast<const float4*>(a.rotations)[li];
    const float opac = a.opacities[li];
    if (use_sh) {
        const int rows_here = min(kSurfelBlock, a.P - base);
        stage_sh_rows(a.shs + (size_t)base * row, rows_here * row, row, stride, a.row_inv, s_sh);
        __syncthreads();
    }
    if (idx >= a.P) return;
    SurfelRec rec;
    float q[4] = {qv.x, qv.y, qv.z, qv.w};
    const float* sh = use_sh ? s_sh + threadIdx.x * stride : nullptr;
    const float* cp = a.colors_precomp ? a.colors_precomp + 3 * idx : nullptr;
    TileRect tr;
    int tiles;
    const int radius = preprocess_surfel(a.cam, pos, sc, q, opac, a.D, sh, cp, rec, tiles, tr, a.tight != 0);
    a.radii[idx] = radius;
    a.rects[idx] = make_uint2(tr.xs, tr.ys);
    if (radius > 0) {
        const float4* src = reinterpret_cast<const float4*>(&rec);
        float4* dst = a.rec + (size_t)idx * kRecQuads;
#pragma unroll
        for (int c = 0; c < kRecQuads; c++) dst[c] = src[c];
    }
}

// ---- binning, step 1: per-tile entry counts -------------------------------------------------------------------
// kBinGroups workgroups each own a contiguous chunk of surfels and histogram its (surfel, tile) pairs in LDS
// (T counters: 10 KB at 800x800, 40 KB at 1600x1600); row g of the G x T matrix M receives the histogram.  Global
// atomics on 2500 hot counters measured 7.5 G/s on MI355X (1.5 M pairs = 200 us); LDS atomics make this ~20 us and
// the scatter below reuses the same chunking, so bucket positions need no global atomics either.
constexpr int kBinGroups = 256;
constexpr int kBinThreads = 1024;   // 16 waves per workgroup (one workgroup per CU): a chunk of ~782 surfels is ONE trip of the loop below instead of
                                    // three dependent ones (radii -> rectangle -> LDS atomics); count 12 -> 7, scatter 17 -> 14 us

#ifndef DGS_BIG_RECT
#define DGS_BIG_RECT 64
#endif
constexpr int kBigRect = DGS_BIG_RECT;   // tiles: a larger tile rectangle is walked by the whole wave instead of its own thread

struct BinArgs {
    int P, ntiles, tiles_x, chunk;   // chunk = surfels per workgroup
    const uint32_t* state;  // [3] num_rendered, longest, overflow (scatter is skipped when overflow is set)
    const int* radii;
    const uint2* rects;
    const float4* rec;
    uint32_t* M;            // [kBinGroups, T] counts, then bucket write cursors
    uint64_t* keys;         // [R] (scatter only)
    uint32_t* sync;         // count only: nsync words cleared for bin_offsets_kernel (its workgroups' aggregates and ticket), or null
    int nsync;
    // scatter only, rider (order != null): one more workgroup at the end of the grid clears tile_last and writes the forward blend's
    // dispatch order from the ranges -- next to the scatter instead of as the serial tail of bin_offsets_kernel's last workgroup
    const uint2* ranges; uint32_t* order; uint32_t* group_xcd; uint32_t* tile_last; int tiles_y, order_mode;
};

__global__ void __launch_bounds__(kBinThreads) count_tiles_lds_kernel(BinArgs a)
{
    extern __shared__ uint32_t s_hist[];
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < a.ntiles; t += kBinThreads) s_hist[t] = 0;
    if (g == 0 && a.sync)
        for (int i = tid; i < a.nsync; i += kBinThreads) a.sync[i] = 0u;
    __syncthreads();
    const int end = min(a.P, (g + 1) * a.chunk);
    const int lane = tid & 63;
    for (int base = g * a.chunk; base < end; base += kBinThreads) {   // (all threads of a wave take the trip together: wave-wide votes below)
        const int idx = base + tid;
        const bool vis = idx < end && a.radii[idx] > 0;
        const uint2 r = vis ? a.rects[idx] : make_uint2(0u, 0u);
        const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
        const int w = max(x1 - x0, 0), area = w * max(y1 - y0, 0);
        const bool big = area > kBigRect;
        if (!big)
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) atomicAdd(&s_hist[y * a.tiles_x + x], 1u);
        // (more than 64 tiles: below that the lanes of the wave would idle, and on the uniform scene the vote alone costs 1 us)
        // a rectangle of hundreds of tiles in ONE thread's loop is what the other 63 lanes wait for (a densified scene keeps a few
        // screen-filling splats: count 36 us / scatter 43 us for a third of the uniform scene's pairs): the wave walks it together
        unsigned long long todo = __ballot(big);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int X0 = __builtin_amdgcn_readlane(x0, src), Y0 = __builtin_amdgcn_readlane(y0, src);
            const int Wd = __builtin_amdgcn_readlane(w, src), n = __builtin_amdgcn_readlane(area, src);
            const float inv_w = __frcp_rn((float)Wd);
            for (int t = lane; t < n; t += 64) {
                const int row = (int)(((float)t + 0.5f) * inv_w);   // t / Wd without the integer division (t < 2^16, Wd <= 2^8: exact)
                atomicAdd(&s_hist[(Y0 + row) * a.tiles_x + X0 + (t - row * Wd)], 1u);
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < a.ntiles; t += kBinThreads) a.M[(size_t)g * a.ntiles + t] = s_hist[t];
}

// Column pass over M.  Block = 64 tiles x kColGroups row groups (kBinGroups / kColGroups rows per thread, all loads
// independent and in flight together); the partial sums of a tile meet in LDS.  The grid is only T / 64 workgroups, so the
// parallelism has to come from inside the workgroup: 16 groups of 16 rows (1024 threads) instead of 4 of 64.
//   counts == nullptr : M[g][t] <- ranges[t].x + sum_{g' < g} M[g'][t]   (where workgroup g starts writing tile t)
//   counts != nullptr : counts[t] = sum_g M[g][t]
constexpr int kColGroups = 16;
__global__ void __launch_bounds__(64 * kColGroups) column_pass_kernel(uint32_t* M, int ntiles, const uint2* ranges, uint32_t* counts)
{
    constexpr int kRows = kBinGroups / kColGroups;
    __shared__ uint32_t s_part[kColGroups][64];
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx;
    uint32_t v[kRows];
    uint32_t sum = 0;
    if (t < ntiles) {
#pragma unroll
        for (int i = 0; i < kRows; i++) v[i] = M[(size_t)(ry * kRows + i) * ntiles + t];
#pragma unroll
        for (int i = 0; i < kRows; i++) sum += v[i];
    }
    s_part[ry][tx] = sum;
    __syncthreads();
    if (t >= ntiles) return;
    if (counts) {
        if (ry == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int r = 0; r < kColGroups; r++) tot += s_part[r][tx];
            counts[t] = tot;
        }
        return;
    }
    uint32_t run = ranges[t].x;
    for (int r = 0; r < ry; r++) run += s_part[r][tx];
#pragma unroll
    for (int i = 0; i < kRows; i++) {
        M[(size_t)(ry * kRows + i) * ntiles + t] = run;
        run += v[i];
    }
}

// duplicateWithKeys (rasterizer_impl.cu:70-111) for the tile-bucketed sort: the tile id is implied by the bucket, so
// the key only carries (depth, surfel index) -- the tie-break order of the reference's stable radix sort.
constexpr size_t kScatterRiderLds = (8 * kOrderBins + 2 * kOrderMaxGroups + 16) * sizeof(uint32_t);   // the rider workgroup's tables (below)
__global__ void __launch_bounds__(kBinThreads) scatter_keys_lds_kernel(BinArgs a)
{
    extern __shared__ uint32_t s_cur[];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (g == kBinGroups) {   // the rider (BinArgs::order); runs on an overflowed frame as well (its ranges are empty, the order still has to exist)
        // its tables live in the DYNAMIC buffer (the rider never uses the cursors; the host sizes the buffer for whichever is larger,
        // kScatterRiderLds): static arrays here would add 40 KB to every workgroup of the launch and push a 3840 x 2160 image
        // (32 400 tiles = 127 KB of cursors) past the 160 KB a workgroup can have
        uint32_t* s_hist = s_cur;
        uint32_t* s_gw = s_hist + 8 * kOrderBins;
        uint32_t* s_gx = s_gw + kOrderMaxGroups;
        uint32_t* s_osum = s_gx + kOrderMaxGroups;
        for (int i = tid; i < a.ntiles; i += kBinThreads) a.tile_last[i] = 0u;
        if (a.order_mode == 4) tile_order_xcd_body(a.ranges, nullptr, a.tiles_x, a.tiles_y, a.order, a.group_xcd, true, s_hist, s_gw, s_gx, s_osum);
        else tile_order_body(a.ranges, nullptr, a.ntiles, a.order, s_hist, s_osum);
        return;
    }
    if (a.state[2] != 0u) return;  // capacity overflow: nothing may be written
    for (int t0 = tid; t0 < a.ntiles; t0 += 4 * kBinThreads) {   // four loads in flight per trip (2500 tiles: one trip)
        uint32_t c[4];
#pragma unroll
        for (int l = 0; l < 4; l++) { const int t = t0 + l * kBinThreads; c[l] = t < a.ntiles ? a.M[(size_t)g * a.ntiles + t] : 0u; }
#pragma unroll
        for (int l = 0; l < 4; l++) { const int t = t0 + l * kBinThreads; if (t < a.ntiles) s_cur[t] = c[l]; }
    }
    __syncthreads();
    const int end = min(a.P, (g + 1) * a.chunk);
    const int lane = tid & 63;
    for (int base = g * a.chunk; base < end; base += kBinThreads) {
        const int idx = base + tid;
        const bool vis = idx < end && a.radii[idx] > 0;
        const uint2 r = vis ? a.rects[idx] : make_uint2(0u, 0u);
        const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
        const int w = max(x1 - x0, 0), area = w * max(y1 - y0, 0);
        const uint32_t depth = area > 0 ? __float_as_uint(a.rec[(size_t)idx * kRecQuads + 4].z) : 0u;
        const uint64_t key = ((uint64_t)depth << 32) | (uint32_t)idx;
        const bool big = area > kBigRect;
        if (!big)
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) a.keys[atomicAdd(&s_cur[y * a.tiles_x + x], 1u)] = key;
        unsigned long long todo = __ballot(big);   // large rectangles: the wave walks them together (count_tiles_lds_kernel)
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int X0 = __builtin_amdgcn_readlane(x0, src), Y0 = __builtin_amdgcn_readlane(y0, src);
            const int Wd = __builtin_amdgcn_readlane(w, src), n = __builtin_amdgcn_readlane(area, src);
            const uint64_t k = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)depth, src) << 32) | (uint32_t)__builtin_amdgcn_readlane(idx, src);
            const float inv_w = __frcp_rn((float)Wd);
            for (int t = lane; t < n; t += 64) {
                const int row = (int)(((float)t + 0.5f) * inv_w);
                a.keys[atomicAdd(&s_cur[(Y0 + row) * a.tiles_x + X0 + (t - row * Wd)], 1u)] = k;
            }
        }
    }
}

// Fallback for images with more tiles than fit the LDS histogram (> ~36k tiles): global atomics.
__global__ void __launch_bounds__(kSurfelBlock) count_tiles_global_kernel(int P, const int* radii, const uint2* rects, int tiles_x,
                                                                          uint32_t* tile_counts)
{
    const int idx = blockIdx.x * kSurfelBlock + threadIdx.x;
    if (idx >= P || !(radii[idx] > 0)) return;
    const uint2 r = rects[idx];
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) atomicAdd(&tile_counts[y * tiles_x + x], 1u);
}

// What is OR-ed into the overflow flag (dgs_set_overflow_flag): bit 0 the lists do not fit the capacity, bit 1 a list is longer than
// promised (option 6), bit 2 that list is longer than the segmented sort reaches (kSegCap * kMaxSegs) -- the caller picks its next
// configuration from the bits instead of trying the tiers one skipped step at a time.
__device__ __forceinline__ int overflow_reason(uint32_t total, uint32_t cap, uint32_t longest, uint32_t list_hint)
{
    int r = total > cap ? 1 : 0;
    if (list_hint > 0 && longest > list_hint) r |= longest > 57344u ? 6 : 2;
    return r;
}

// Exclusive scan of the per-tile counts (T = 2500 at 800x800, 10000 at 1600x1600) by ONE workgroup:
// ranges[t] = [start, end) (identifyTileRanges, rasterizer_impl.cu:116-138), cursor[t] = start (scatter cursors),
// *total_out = num_rendered (rasterizer_impl.cu:281).
// `cap` > 0 (capacity mode, no host round trip): if the lists do not fit the pre-sized binning buffer every tile is
// left empty (the frame renders as background) and *overflow is raised instead of writing out of bounds.
__global__ void __launch_bounds__(1024) scan_tiles_kernel(const uint32_t* counts, int ntiles, uint2* ranges, uint32_t* cursor,
                                                          uint32_t* total_out /*[3]: num_rendered, longest list, overflow*/,
                                                          uint32_t cap, int* overflow, uint32_t* order /*or null*/, uint32_t list_hint,
                                                          int tiles_x, int tiles_y, int order_mode, uint32_t* group_xcd,
                                                          uint32_t* tile_last /*[T]: zeroed here*/, uint32_t* long_thr /*[2]: [0] written here*/, uint32_t long_div)
{
    __shared__ uint32_t s_wsum[4], s_wmax[4];
    // the forward blend writes the per-tile maximum of the last contributor with atomicMax (a long tile is four workgroups)
    for (int t = threadIdx.x; t < ntiles; t += 1024) tile_last[t] = 0u;
    __shared__ uint32_t s_hist[8 * kOrderBins];
    __shared__ uint32_t s_gw[kOrderMaxGroups], s_gx[kOrderMaxGroups];
    __shared__ uint32_t s_osum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) {   // the scan proper: 256 threads, `per` consecutive tiles each
        const int per = (ntiles + 255) / 256;
        const int t0 = tid * per, t1 = min(ntiles, t0 + per);
        uint32_t sum = 0, mx = 0;
        for (int t = t0; t < t1; t++) { const uint32_t c = counts[t]; sum += c; mx = max(mx, c); }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t n = __shfl_up(inc, d, 64);
            if (lane >= d) inc += n;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        if (lane == 63) s_wsum[wave] = inc;
        if (lane == 0) s_wmax[wave] = mx;
        s_hist[tid] = inc - sum;   // exclusive prefix inside the wave, parked for after the barrier
    }
    __syncthreads();
    if (tid < 256) {
        const int per = (ntiles + 255) / 256;
        const int t0 = tid * per, t1 = min(ntiles, t0 + per);
        const uint32_t total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        const uint32_t longest = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
        // capacity mode: the lists must fit the pre-sized buffer, and -- if the caller promised a longest list (list_hint:
        // only the sort kernels for lists up to it were launched) -- no list may be longer than promised
        const bool over = cap > 0 && (total > cap || (list_hint > 0 && longest > list_hint));
        uint32_t run = s_hist[tid];
        for (int w = 0; w < wave; w++) run += s_wsum[w];
        for (int t = t0; t < t1; t++) {
            const uint32_t c = over ? 0u : counts[t];
            if (over) run = 0;
            ranges[t] = make_uint2(run, run + c);
            if (cursor) cursor[t] = run;
            run += c;
        }
        if (tid == 0) {
            total_out[0] = total;
            total_out[1] = longest;
            total_out[2] = over ? 1u : 0u;
            if (over && overflow) atomicOr(overflow, overflow_reason(total, cap, longest, list_hint));
            // long-tile path of the forward blend (kernels_blend.h): worth its extra arithmetic only where ONE tile's serial walk is as
            // long as the whole launch's throughput-bound time -- a list is long from 768 entries and num_rendered / long_div on
            long_thr[0] = max(768u, total / max(long_div, 1u));
        }
    }
    if (order) {
        // dispatch order of the forward blend (longest list first, kernels_blend.h): the ranges were just written by this
        // workgroup, visible to all of its threads after the fence + barrier
        __threadfence_block();
        __syncthreads();
        if (order_mode == 4) tile_order_xcd_body(ranges, nullptr, tiles_x, tiles_y, order, group_xcd, true, s_hist, s_gw, s_gx, s_osum);
        else tile_order_body(ranges, nullptr, ntiles, order, s_hist, s_osum);
    }
}

// Column sums, scan of the tile counts, bucket cursors and the forward's dispatch order in ONE launch (LDS-histogram path): replaces
// column_pass(counts) + scan_tiles + column_pass(cursors), three dependent launches of 5 + 10 + 5 us, each mostly launch and memory
// latency.  Workgroup b = 64 tiles x all kBinGroups rows of M, as in column_pass_kernel; its 64 tile counts are scanned by its
// first wave and the offset of the block is the sum of the aggregates of the workgroups before it, which every workgroup
// publishes (flag in the top bit) the moment it has it -- a one-level decoupled look-back: nobody waits for anything but
// lower-numbered workgroups, which are dispatched first.  The rows of M were read once and are rewritten from registers.
// The workgroup that finishes LAST (ticket) does what needs every range: num_rendered, the longest list, the capacity
// check (on overflow it clears all ranges again), tile_last, long_thr[0] and the dispatch order of the forward blend.
// sync: [2 nb + 1] words, zero on entry (count_tiles_lds_kernel clears them): aggregates, maxima, ticket.
struct OffsetsArgs {
    uint32_t* M; int ntiles; uint2* ranges;
    uint32_t* state;        // [3]: num_rendered, longest list, overflow
    uint32_t cap; int* overflow; uint32_t list_hint;
    uint32_t* order;        // or null
    int tiles_x, tiles_y, order_mode;
    uint32_t* group_xcd; uint32_t* tile_last; uint32_t* long_thr; uint32_t long_div;
    uint32_t* sync;
    int order_later;        // tile_last and the dispatch order are left to the scatter launch's rider workgroup (capacity mode: it always runs)
};

__global__ void __launch_bounds__(64 * kColGroups) bin_offsets_kernel(OffsetsArgs a)
{
    constexpr int kRows = kBinGroups / kColGroups;
    constexpr uint32_t kFlag = 0x80000000u;
    __shared__ uint32_t s_part[kColGroups][64];
    __shared__ uint32_t s_start[64];
    __shared__ uint32_t s_last;
    __shared__ uint32_t s_hist[8 * kOrderBins];
    __shared__ uint32_t s_gw[kOrderMaxGroups], s_gx[kOrderMaxGroups];
    __shared__ uint32_t s_osum[16];
    const int tid = threadIdx.x, tx = tid & 63, ry = tid >> 6;
    const int b = blockIdx.x, nb = gridDim.x;
    const int t = b * 64 + tx;
    uint32_t* const agg = a.sync;
    uint32_t* const amax = a.sync + nb;
    uint32_t* const ticket = a.sync + 2 * nb;
    uint32_t v[kRows];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kRows; i++) v[i] = 0;
    if (t < a.ntiles) {
#pragma unroll
        for (int i = 0; i < kRows; i++) v[i] = a.M[(size_t)(ry * kRows + i) * a.ntiles + t];
#pragma unroll
        for (int i = 0; i < kRows; i++) sum += v[i];
    }
    s_part[ry][tx] = sum;
    __syncthreads();
    if (ry == 0) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < kColGroups; r++) c += s_part[r][tx];
        uint32_t inc = c, mx = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t n = __shfl_up(inc, d, 64);
            if (tx >= d) inc += n;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        if (tx == 63) {
            __hip_atomic_store(&amax[b], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&agg[b], kFlag | inc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t before = 0;
        for (int j = tx; j < b; j += 64) {
            uint32_t w = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (!(w & kFlag)) {
                __builtin_amdgcn_s_sleep(1);
                w = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            before += w & ~kFlag;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
        const uint32_t start = before + inc - c;
        s_start[tx] = start;
        if (t < a.ntiles) a.ranges[t] = make_uint2(start, start + c);
    }
    __syncthreads();
    if (t < a.ntiles) {
        uint32_t run = s_start[tx];
        for (int r = 0; r < ry; r++) run += s_part[r][tx];
#pragma unroll
        for (int i = 0; i < kRows; i++) {
            a.M[(size_t)(ry * kRows + i) * a.ntiles + t] = run;
            run += v[i];
        }
    }
    // ticket: the ranges of this workgroup are released before it, the last one acquires everybody's.  ONE thread fences (behind the
    // barrier that completes the workgroup's stores): an agent-scope fence writes back / invalidates the XCD's L2, and 1024 threads
    // x 40 workgroups doing that made this kernel 39 us instead of 12
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const bool last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(nb - 1);
        if (last) __threadfence();
        s_last = last ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    uint32_t tot = 0, lng = 0;
    for (int j = tid; j < nb; j += 64 * kColGroups) {
        tot += __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~kFlag;
        lng = max(lng, __hip_atomic_load(&amax[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        tot += (uint32_t)__shfl_xor((int)tot, d, 64);
        lng = max(lng, (uint32_t)__shfl_xor((int)lng, d, 64));
    }
    if (tx == 0) { s_part[0][ry] = tot; s_part[1][ry] = lng; }
    __syncthreads();
    uint32_t total = 0, longest = 0;
#pragma unroll
    for (int r = 0; r < kColGroups; r++) { total += s_part[0][r]; longest = max(longest, s_part[1][r]); }
    // capacity mode: the lists must fit the pre-sized buffer, and -- if the caller promised a longest list (list_hint: only the
    // sort kernels for lists up to it were launched) -- no list may be longer than promised (scan_tiles_kernel)
    const bool over = a.cap > 0 && (total > a.cap || (a.list_hint > 0 && longest > a.list_hint));
    if (over)
        for (int i = tid; i < a.ntiles; i += 64 * kColGroups) a.ranges[i] = make_uint2(0u, 0u);
    if (!a.order_later)
        for (int i = tid; i < a.ntiles; i += 64 * kColGroups) a.tile_last[i] = 0u;
    if (tid == 0) {
        a.state[0] = total;
        a.state[1] = longest;
        a.state[2] = over ? 1u : 0u;
        if (over && a.overflow) atomicOr(a.overflow, overflow_reason(total, a.cap, longest, a.list_hint));
        a.long_thr[0] = max(768u, total / max(a.long_div, 1u));
    }
    if (a.order && !a.order_later) {
        __threadfence_block();
        __syncthreads();
        if (a.order_mode == 4) tile_order_xcd_body(a.ranges, nullptr, a.tiles_x, a.tiles_y, a.order, a.group_xcd, true, s_hist, s_gw, s_gx, s_osum);
        else tile_order_body(a.ranges, nullptr, a.ntiles, a.order, s_hist, s_osum);
    }
}

struct ScatterArgs {
    int P;
    const int* radii;
    const float4* rec;
    const uint2* rects;
    const uint32_t* state;  // [3] num_rendered, longest, overflow
    uint32_t* cursor;       // [T] running write position of every tile bucket
    uint64_t* keys;         // [R] out: depth bits << 32 | surfel index, bucketed by tile (unordered inside a bucket)
    int tiles_x;
};

// global-atomics variant of scatter_keys_lds_kernel (fallback for very large tile counts)
__global__ void __launch_bounds__(kSurfelBlock) scatter_keys_kernel(ScatterArgs a)
{
    const int idx = blockIdx.x * kSurfelBlock + threadIdx.x;
    if (idx >= a.P || !(a.radii[idx] > 0) || a.state[2] != 0u) return;
    const uint2 r = a.rects[idx];
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    if (x1 <= x0 || y1 <= y0) return;
    const uint64_t key = ((uint64_t)__float_as_uint(a.rec[(size_t)idx * kRecQuads + 4].z) << 32) | (uint32_t)idx;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const uint32_t pos = atomicAdd(&a.cursor[y * a.tiles_x + x], 1u);
            a.keys[pos] = key;
        }
}

// Per-tile sort by (depth, index): one workgroup per tile, bitonic network on 64-bit keys.
//   sort_tiles_lds_kernel<CAP>: buckets with lo < n <= CAP are padded to a power of two and sorted in LDS
//       (CAP = 2048 -> 16 KB and ten resident workgroups per CU, the common case: a few hundred entries;
//        CAP = 16384 -> 128 KB for crowded tiles);
//   sort_tiles_global_kernel: buckets longer than that are copied, padded, into scratch [2*start, 2*start + n2)
//       (disjoint across tiles because n2 < 2n) and sorted there by the same network -- slow, but only reachable
//       by degenerate views (tens of thousands of surfels over one tile).
// Thread i of a pass handles the pair (l, l + j) with l = ((i & ~(j-1)) << 1) | (i & (j-1)).  For j <= 64 the 64 pairs
// of a wave (i = 64w .. 64w+63, and again every 256) live in the wave's own 128-key window, so consecutive stages with
// j <= 64 need no workgroup barrier -- only the LDS ordering of a single wave.  A 1024-key sort keeps 9 of its 55 barriers.  (`wave_local` must be false when `k` is global memory.)
__device__ __forceinline__ void bitonic_network(uint64_t* k, int n2, int tid, bool wave_local)
{
    const int half = n2 >> 1;
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            // four independent pairs per thread and trip: all eight loads are in flight before the first compare
            for (int i0 = tid; i0 < half; i0 += 1024) {
                int l[4];
                uint64_t a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 256 * u;
                    l[u] = ((i & ~(j - 1)) << 1) | (i & (j - 1));  // j is a power of two
                    if (i < half) { a[u] = k[l[u]]; b[u] = k[l[u] + j]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 256 * u;
                    const bool up = (l[u] & kk) == 0;
                    if (i < half && (a[u] > b[u]) == up) { k[l[u]] = b[u]; k[l[u] + j] = a[u]; }
                }
            }
            // the next stage is (kk, j/2), or (2kk, kk) after j == 1; a workgroup barrier is needed only when this
            // stage or the next one crosses the 128-key windows (stride > 64)
            const int j_next = j > 1 ? (j >> 1) : kk;
            if (wave_local && j <= 64 && j_next <= 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            else __syncthreads();
        }
}

template <int CAP>
__global__ void __launch_bounds__(256) sort_tiles_lds_kernel(const uint2* ranges, const uint64_t* keys, uint32_t* point_list, int lo)
{
    __shared__ uint64_t s_keys[CAP];
    const uint2 rg = ranges[blockIdx.x];
    const int n = (int)(rg.y - rg.x);
    if (n <= lo || n > CAP) return;
    const uint64_t* gk = keys + rg.x;
    const int tid = threadIdx.x;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) s_keys[i] = i < n ? gk[i] : ~0ull;
    __syncthreads();
    bitonic_network(s_keys, n2, tid, true);
    for (int i = tid; i < n; i += 256) point_list[rg.x + i] = (uint32_t)s_keys[i];
}

// Register-resident variant for buckets of up to 2048 entries (the common case; the LDS network above spends its time in
// the LDS pipe: four 64-bit LDS operations per compared pair and stage).  Thread t of the 256 keeps the keys of elements
// t + 256 s (s < SLOTS) in registers; the partner of element e in stage (kk, j) is e ^ j, and e keeps the smaller key iff
// ((e & j) == 0) == ((e & kk) == 0).  So
//   j >= 256 : the partner is another slot of the same thread  -> compare-exchange in registers, no data movement;
//   j <  64  : the partner is the same slot of lane ^ j        -> two ds_bpermute per key;
//   j = 64, 128 : another wave                                 -> one round trip through LDS (9 of the 66 stages at 2048).
// Value of `v` in lane (lane ^ j): two ds_bpermute.  (DPP quad permutes / bank-masked row shifts for j <= 8 were measured and
// are slower: 76 us against 49 us for the kernel at 200 k surfels -- the VALU, not the LDS pipe, is what the network saturates.)
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v, int j)
{
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return ((uint64_t)(uint32_t)__shfl_xor((int)hi, j, 64) << 32) | (uint32_t)__shfl_xor((int)lo, j, 64);
}

template <int SLOTS, int DS>
__device__ __forceinline__ void sort_cx_slots(uint64_t (&k)[SLOTS], int kk, int tid)
{
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        if (s & DS) continue;
        const bool up = ((tid + 256 * s) & kk) == 0;
        const uint64_t a = k[s], b = k[s | DS];
        const bool sw = (a > b) == up;
        k[s] = sw ? b : a;
        k[s | DS] = sw ? a : b;
    }
}

template <int SLOTS>
__device__ __forceinline__ void sort_tile_regs(uint64_t* s_keys, const uint64_t* gk, int n, uint32_t* out, int tid)
{
    uint64_t k[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const int e = tid + 256 * s;
        k[s] = e < n ? gk[e] : ~0ull;
    }
    for (int kk = 2; kk <= 256 * SLOTS; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= 256) {
                if constexpr (SLOTS >= 2) { if (j == 256) sort_cx_slots<SLOTS, 1>(k, kk, tid); }
                if constexpr (SLOTS >= 4) { if (j == 512) sort_cx_slots<SLOTS, 2>(k, kk, tid); }
                if constexpr (SLOTS >= 8) { if (j == 1024) sort_cx_slots<SLOTS, 4>(k, kk, tid); }
                continue;
            }
            const bool lower = (tid & j) == 0;
            if (j >= 64) {
#pragma unroll
                for (int s = 0; s < SLOTS; s++) s_keys[tid + 256 * s] = k[s];
                __syncthreads();
#pragma unroll
                for (int s = 0; s < SLOTS; s++) {
                    const uint64_t p = s_keys[(tid ^ j) + 256 * s];
                    const bool keep_min = lower == ((((tid + 256 * s) & kk)) == 0);
                    k[s] = ((p < k[s]) == keep_min) ? p : k[s];
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int s = 0; s < SLOTS; s++) {
                    const uint64_t p = lane_xor_u64(k[s], j);
                    const bool keep_min = lower == ((((tid + 256 * s) & kk)) == 0);
                    k[s] = ((p < k[s]) == keep_min) ? p : k[s];
                }
            }
        }
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const int e = tid + 256 * s;
        if (e < n) out[e] = (uint32_t)k[s];
    }
}

__global__ void __launch_bounds__(256) sort_tiles_reg_kernel(const uint2* ranges, const uint64_t* keys, uint32_t* point_list)
{
    __shared__ uint64_t s_keys[2048];
    const uint2 rg = ranges[blockIdx.x];
    const int n = (int)(rg.y - rg.x);
    if (n <= 0 || n > 2048) return;
    const uint64_t* gk = keys + rg.x;
    uint32_t* out = point_list + rg.x;
    const int tid = threadIdx.x;
    if (n <= 256) sort_tile_regs<1>(s_keys, gk, n, out, tid);
    else if (n <= 512) sort_tile_regs<2>(s_keys, gk, n, out, tid);
    else if (n <= 1024) sort_tile_regs<4>(s_keys, gk, n, out, tid);
    else sort_tile_regs<8>(s_keys, gk, n, out, tid);
}

// ---- per-tile radix sort ---------------------------------------------------------------------------------------------
// One workgroup per tile bucket, LSD radix on the depth bits with 8-bit digits, everything in LDS:
//   * only the bits that DIFFER inside the bucket are sorted: depths of one tile share sign, most of the exponent and often
//     more (xor of the bucket's min and max), typically 3 passes instead of 4;
//   * a pass is a stable counting sort: wave w owns a contiguous quarter of the bucket; for each of its 64-element slots the
//     lanes that hold the same digit find each other with 8 ballots (one per digit bit), rank = population count of the
//     peers below the lane, a wave-private LDS histogram carries the running count from slot to slot; the 4 x 256 wave
//     histograms are scanned by the 256 threads (digit-major, then wave) and every element is written to its final place;
//   * the (depth, surfel index) order of the reference's stable radix sort (rasterizer_impl.cu:304-309: ties in depth keep the
//     emission order = ascending surfel index) is restored at the end: the bucket arrives in arbitrary order, so equal
//     depths -- exact float equality, i.e. duplicated surfels -- come out in arbitrary order; each such run is sorted by
//     index by the thread that finds its head.
// O(n) operations per pass instead of the O(n log^2 n) compare-exchanges of the bitonic network (which saturated the VALU:
// 62 us at 200 k surfels / 800x800); keys are 32-bit here (depth bits; the index rides along as payload).
// SEG = true (lists longer than CAP): blockIdx.y = segment; the workgroup sorts entries [y CAP, (y + 1) CAP) of every list of
// more than CAP and at most CAP * gridDim.y entries and writes the sorted (depth, index) KEYS into the list's scratch area;
// merge_segments_kernel then ranks every key among the other segments.  `seg_out` = scratch of 2 R keys (list at 2 * range.x).
// CAP = kSegCap = 2048 (36 KB of LDS), up to kMaxSegs = 28 segments per list.  History: round 3 used segments of 3584 (60 KB); the
// 8192-entry / 132-KB instantiation that round 2 used for lists of 2 k - 8 k entries never took less than ~73 us per launch, with or
// without work (36 workgroups that find nothing to sort: 73 us; the segment kernel: 2 us).  The cause was not isolated -- bare
// allocations of 16 .. 160 KB launch in 2.5 us whatever their size (tools/micro/lds_launch_bench.hip) -- the smaller instantiations do
// not show it, and their segments run side by side.
// Round 4: the segments are as long as the main kernel's lists (2048) and EVERY list beyond that is cut into them -- the 3584-entry
// single-workgroup instantiation that used to take the lists of 2 049 .. 3 584 entries is gone: one workgroup of four waves needs
// 25-50 us for such a list (the launch a densified scene waited for), two or more segments side by side + the rank / merge step half.
constexpr int kSegCap = 2048;
constexpr int kMaxSegs = 28;
template <int CAP, bool SEG = false>
__global__ void __launch_bounds__(256) sort_tiles_radix_kernel(const uint2* ranges, int ntiles, const uint64_t* keys, uint32_t* point_list, int lo,
                                                               uint64_t* seg_out = nullptr, const uint32_t* state = nullptr /*[1] = longest list*/)
{
    if (SEG && state && (uint32_t)blockIdx.y * (uint32_t)CAP >= state[1]) return;   // no list of this launch reaches this segment index
    __shared__ uint32_t s_key[2][CAP];
    __shared__ uint32_t s_val[2][CAP];
    __shared__ uint32_t s_hist[4][256];      // per wave: running digit counts, then exclusive global offsets
    __shared__ uint32_t s_red[8];
    // grid-stride over the tiles: the launch for long lists uses a small grid (a 132 KB workgroup per tile that exits at once
    // would cost ten dispatch rounds of 2500 workgroups for nothing)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint2 rg = ranges[tile];
    const int n_list = (int)(rg.y - rg.x);
    const int seg0 = SEG ? (int)blockIdx.y * CAP : 0;
    if (SEG ? (n_list <= CAP || n_list > CAP * (int)gridDim.y || seg0 >= n_list) : (n_list <= lo || n_list > CAP)) continue;
    const int n = SEG ? min(CAP, n_list - seg0) : n_list;
    __syncthreads();   // the previous tile's last reads of the LDS arrays
    const uint64_t* gk = keys + rg.x + seg0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- load, and the bits in which the bucket's depths differ
    uint32_t kand = 0xffffffffu, kor = 0u;
    {
        // all of the thread's keys are requested before the first is used (a load -> LDS store loop is one memory round trip per
        // trip: three for the average list of the metric scene, of a kernel whose workgroups run in 2.4 rounds)
        constexpr int kLoads = CAP / 256;
        uint64_t kb[kLoads];
#pragma unroll
        for (int l = 0; l < kLoads; l++) { const int i = tid + 256 * l; kb[l] = i < n ? gk[i] : 0ull; }
#pragma unroll
        for (int l = 0; l < kLoads; l++) {
            const int i = tid + 256 * l;
            if (i < n) {
                const uint32_t d = (uint32_t)(kb[l] >> 32);
                s_key[0][i] = d;
                s_val[0][i] = (uint32_t)kb[l];
                kand &= d; kor |= d;
            }
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        kand &= (uint32_t)__shfl_xor((int)kand, sft, 64);
        kor |= (uint32_t)__shfl_xor((int)kor, sft, 64);
    }
    if (lane == 0) { s_red[wave] = kand; s_red[4 + wave] = kor; }
    __syncthreads();
    const uint32_t diff = (s_red[0] & s_red[1] & s_red[2] & s_red[3]) ^ (s_red[4] | s_red[5] | s_red[6] | s_red[7]);
    const int nbits = diff ? 32 - __builtin_clz(diff) : 0;
    // wave w owns elements [w0, w1): contiguous, so array order == (wave, slot, lane) order
    const int per = (n + 3) >> 2;
    const int w0 = wave * per, w1 = min(n, w0 + per);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int cur = 0;
    for (int shift = 0; shift < nbits; shift += 8, cur ^= 1) {
        s_hist[wave][lane] = 0; s_hist[wave][lane + 64] = 0; s_hist[wave][lane + 128] = 0; s_hist[wave][lane + 192] = 0;
        // (wave-private rows: the wave's own LDS operations are ordered, no barrier needed before the first slot)
        const uint32_t* kin = s_key[cur];
        const uint32_t* vin = s_val[cur];
        // ---- rank inside the wave's range
        constexpr int kMaxSlots = (CAP / 4 + 63) / 64;
        uint32_t my_rank[kMaxSlots];      // rank of my element of slot e among the wave's elements with the same digit
#pragma unroll
        for (int e = 0; e < kMaxSlots; e++) {
            const int pos = w0 + e * 64 + lane;
            if (w0 + e * 64 >= w1) break;                     // wave-uniform
            const bool live = pos < w1;
            const uint32_t d = live ? ((kin[pos] >> shift) & 255u) : 256u;   // 256: no peer among the live lanes
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(peers & lt_mask), total = (uint32_t)__popcll(peers);
            uint32_t base = 0;
            if (live) base = s_hist[wave][d];
            my_rank[e] = base + below;
            if (live && below == 0) s_hist[wave][d] = base + total;   // one leader per digit
        }
        __syncthreads();
        // ---- exclusive offsets over (digit, wave): thread t = digit t
        {
            const uint32_t c0 = s_hist[0][tid], c1 = s_hist[1][tid], c2 = s_hist[2][tid], c3 = s_hist[3][tid];
            const uint32_t tot = c0 + c1 + c2 + c3;
            uint32_t inc = tot;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, dd, 64);
                if (lane >= dd) inc += up;
            }
            if (lane == 63) s_red[wave] = inc;
            __syncthreads();
            uint32_t off = inc - tot;
            for (int w = 0; w < wave; w++) off += s_red[w];
            s_hist[0][tid] = off; s_hist[1][tid] = off + c0; s_hist[2][tid] = off + c0 + c1; s_hist[3][tid] = off + c0 + c1 + c2;
        }
        __syncthreads();
        // ---- scatter
        uint32_t* kout = s_key[cur ^ 1];
        uint32_t* vout = s_val[cur ^ 1];
#pragma unroll
        for (int e = 0; e < kMaxSlots; e++) {
            const int pos = w0 + e * 64 + lane;
            if (w0 + e * 64 >= w1) break;
            if (pos < w1) {
                const uint32_t k = kin[pos];
                const uint32_t dst = s_hist[wave][(k >> shift) & 255u] + my_rank[e];
                kout[dst] = k;
                vout[dst] = vin[pos];
            }
        }
        __syncthreads();
    }
    // ---- ties in depth: ascending surfel index (the head of a run sorts it; runs are duplicated surfels, i.e. short)
    uint32_t* kf = s_key[cur];
    uint32_t* vf = s_val[cur];
    for (int i = tid; i < n - 1; i += 256) {
        if (kf[i] == kf[i + 1] && (i == 0 || kf[i - 1] != kf[i])) {
            int end = i + 2;
            while (end < n && kf[end] == kf[i]) end++;
            for (int a = i + 1; a < end; a++) {          // insertion sort of vf[i, end)
                const uint32_t v = vf[a];
                int b = a - 1;
                while (b >= i && vf[b] > v) { vf[b + 1] = vf[b]; b--; }
                vf[b + 1] = v;
            }
        }
    }
    __syncthreads();
    if (SEG) {
        uint64_t* so = seg_out + 2 * (size_t)rg.x + seg0;
        for (int i = tid; i < n; i += 256) so[i] = ((uint64_t)kf[i] << 32) | vf[i];
    } else {
        for (int i = tid; i < n; i += 256) point_list[rg.x + i] = vf[i];
    }
    }
}

// Lists of more than kSegCap entries, second step: every key of segment y finds its place in the whole list -- its index in
// its own (sorted) segment plus, for every other segment, the number of keys below it.  Keys are unique inside a list (the
// surfel index is part of the key), so "below" needs no tie rule and the result is the reference's stable (depth, index) order.
// A thread owns 8 CONSECUTIVE keys of its segment; the other segments pass through LDS one at a time (17 KB, coalesced load):
// the number of its keys below each of the thread's keys is a branch-free binary search in LDS.
// Searching the other segments where they lie, in global memory, is a chain of ~200 dependent loads per thread: 0.67 ms.
// (LDS index i lives at i + i / 32: the threads' search positions are ~32 keys apart, which would be one bank.)
__global__ void __launch_bounds__(256) merge_segments_kernel(const uint2* ranges, int ntiles, const uint64_t* seg_keys /*scratch*/, uint32_t* point_list,
                                                             const uint32_t* state /*[1] = longest list*/)
{
    if ((uint32_t)blockIdx.y * (uint32_t)kSegCap >= state[1]) return;   // no list of this launch reaches this segment index
    constexpr int kPer = kSegCap / 256;
    __shared__ uint64_t s_other[kSegCap + kSegCap / 32];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint2 rg = ranges[tile];
        const int n_list = (int)(rg.y - rg.x);
        const int seg = (int)blockIdx.y, seg0 = seg * kSegCap;
        if (n_list <= kSegCap || n_list > kSegCap * (int)gridDim.y || seg0 >= n_list) continue;   // (workgroup-uniform)
        const uint64_t* base = seg_keys + 2 * (size_t)rg.x;
        const int n = min(kSegCap, n_list - seg0), nseg = (n_list + kSegCap - 1) / kSegCap;
        const int i0 = (int)threadIdx.x * kPer, cnt = max(0, min(kPer, n - i0));
        uint64_t mine[kPer];
        int rank[kPer];
#pragma unroll
        for (int i = 0; i < kPer; i++) { mine[i] = i < cnt ? base[seg0 + i0 + i] : ~0ull; rank[i] = i0 + i; }
        for (int o = 0; o < nseg; o++) {
            if (o == seg) continue;
            const int len = min(kSegCap, n_list - o * kSegCap);
            __syncthreads();
            for (int i = threadIdx.x; i < len; i += 256) s_other[i + (i >> 5)] = base[o * kSegCap + i];
            __syncthreads();
            // branch-free lower bound of all 8 keys at once: 12 rounds of 8 independent LDS reads (a merge-style walk from key to
            // key is a chain of dependent reads inside a divergent loop: 25 us per segment instead of ~2)
            int pos[kPer];
#pragma unroll
            for (int i = 0; i < kPer; i++) pos[i] = 0;
#pragma unroll
            for (int step = 2048; step >= 1; step >>= 1) {
#pragma unroll
                for (int i = 0; i < kPer; i++) {
                    const int p = pos[i] + step;
                    const int q = p <= len ? p - 1 : 0;
                    const bool below = (p <= len) & (s_other[q + (q >> 5)] < mine[i]);
                    pos[i] = below ? p : pos[i];
                }
            }
#pragma unroll
            for (int i = 0; i < kPer; i++) rank[i] += pos[i];   // (padding keys of a thread with fewer than 8 are never written)
        }
#pragma unroll
        for (int i = 0; i < kPer; i++)
            if (i < cnt) point_list[rg.x + rank[i]] = (uint32_t)mine[i];
    }
}

__global__ void __launch_bounds__(256) sort_tiles_global_kernel(const uint2* ranges, int ntiles, const uint64_t* keys, uint64_t* scratch /*[2R]*/,
                                                                uint32_t* point_list, int lo)
{
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // grid-stride: launched with a small grid
        const uint2 rg = ranges[tile];
        const int n = (int)(rg.y - rg.x);
        if (n <= lo) continue;
        const uint64_t* gk = keys + rg.x;
        uint64_t* sk = scratch + 2 * (size_t)rg.x;
        const int tid = threadIdx.x;
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = tid; i < n2; i += 256) sk[i] = i < n ? gk[i] : ~0ull;
        __syncthreads();
        bitonic_network(sk, n2, tid, false);
        for (int i = tid; i < n; i += 256) point_list[rg.x + i] = (uint32_t)sk[i];
        __syncthreads();
    }
}

struct SurfelBwdArgs {
    int P, D, M;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* shs;            // null when colours were precomputed
    unsigned row_inv;            // ceil(2^32 / (3 M)): e / (3 M) == __umulhi(e, row_inv) for the element counts of one workgroup
    Camera cam;
    const int* radii;
    const float4* rec;
    const float* acc;            // [P, kAccFloats] accumulated by the backward blend
    float* dL_dmean2D;           // [P,3]
    float* dL_dnormal;           // [P,3]
    float* dL_dopacity;          // [P]
    float* dL_dcolor;            // [P,3]
    float* dL_dmean3D;           // [P,3]
    float* dL_dtransMat;         // [P,9]
    float* dL_dsh;               // [P,M,3]
    int sh_all_rows;             // 1: dL_dsh is written for EVERY row and coefficient (zeros where the reference leaves the caller's values)
    float* dL_dscale;            // [P,2]
    float* dL_drot;              // [P,4]
};

// Fused computeAABB-bwd + preprocessCUDA-bwd (backward.cu:533-649).  The reference's outputs are caller-zeroed
// (rasterize_points.cu:194-202) and culled surfels left untouched; here the eight per-surfel arrays are written for EVERY row
// (zeros for culled surfels, 112 B each) so that the caller needs no 22 MB fill in front of the backward; dL_dsh keeps the
// reference's rule (visible rows, first (D+1)^2 coefficients): it is the array callers accumulate into (gradient sinks).
__global__ void __launch_bounds__(kSurfelBlock) surfel_bwd_kernel(SurfelBwdArgs a)   // 135 VGPRs; forcing 128 (full residency, 3 spills) measured 43 against 41 us: bandwidth-bound
{
    // SH rows (192 B per surfel at degree 3) are the bulk of this kernel's traffic and one-thread-per-surfel access to
    // them is a 64-way scatter per instruction: the workgroup's rows are staged through LDS with coalesced transfers in
    // both directions (coefficients in, dL_dsh out through the same rows); row stride M*3 + 1 keeps the per-thread row
    // accesses bank-conflict free.
    extern __shared__ float s_sh[];
    const int base = blockIdx.x * kSurfelBlock;
    const int idx = base + threadIdx.x;
    const bool active = idx < a.P && a.radii[idx] > 0;
    const int row = a.M * 3, stride = row + 1;
    const int rows_here = min(kSurfelBlock, a.P - base);
    if (a.shs) {
        stage_sh_rows(a.shs + (size_t)base * row, rows_here * row, row, stride, a.row_inv, s_sh);
        __syncthreads();
    }
    if (active) {
        // (requesting these in front of the staging saves a memory round trip but costs the third wave per SIMD: 191 registers)
        SurfelRec rec;
        {
            float4* dst = reinterpret_cast<float4*>(&rec);
            const float4* src = a.rec + (size_t)idx * kRecQuads;
#pragma unroll
            for (int c = 0; c < kRecQuads; c++) dst[c] = src[c];
        }
        float acc[kAccFloats];
        {
            const float4* src = reinterpret_cast<const float4*>(a.acc + (size_t)idx * kAccFloats);
#pragma unroll
            for (int c = 0; c < kAccFloats / 4; c++) {
                float4 v = src[c];
                acc[4 * c] = v.x; acc[4 * c + 1] = v.y; acc[4 * c + 2] = v.z; acc[4 * c + 3] = v.w;
            }
        }
        float pos[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        float sc[2] = {a.scales[2 * idx], a.scales[2 * idx + 1]};
        const float4 qv = reinterpret_cast<const float4*>(a.rotations)[idx];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        SurfelGrads g;
        surfel_backward(a.cam, pos, sc, q, rec, acc, g);
        if (a.shs) {
            // coefficients to registers first: sh_backward overwrites the LDS row with dL_dsh while it still reads sh
            float* my = s_sh + threadIdx.x * stride;
            float shr[48];
#pragma unroll
            for (int c = 0; c < 48; c++) shr[c] = c < row ? my[c] : 0.f;
            sh_backward(a.D, shr, pos, a.cam.campos, rec.flags, acc + kAccColor, my, g.dmean3D);
        }
        for (int c = 0; c < 3; c++) {
            a.dL_dmean3D[3 * idx + c] = g.dmean3D[c];
            if (a.dL_dcolor) a.dL_dcolor[3 * idx + c] = acc[kAccColor + c];      // (optional outputs: dgs_surfel_rasterizer.h)
            if (a.dL_dnormal) a.dL_dnormal[3 * idx + c] = acc[kAccNormal + c];
        }
        a.dL_dmean2D[3 * idx] = g.dmean2D[0];
        a.dL_dmean2D[3 * idx + 1] = g.dmean2D[1];
        a.dL_dmean2D[3 * idx + 2] = 0.f;
        a.dL_dopacity[idx] = acc[kAccOpacity];
        if (a.dL_dtransMat)
            for (int c = 0; c < 9; c++) a.dL_dtransMat[9 * idx + c] = g.dT[c];
        a.dL_dscale[2 * idx] = g.dscale[0];
        a.dL_dscale[2 * idx + 1] = g.dscale[1];
        reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(g.drot[0], g.drot[1], g.drot[2], g.drot[3]);
    } else if (idx < a.P) {
        for (int c = 0; c < 3; c++) {
            a.dL_dmean3D[3 * idx + c] = 0.f;
            if (a.dL_dcolor) a.dL_dcolor[3 * idx + c] = 0.f;
            if (a.dL_dnormal) a.dL_dnormal[3 * idx + c] = 0.f;
            a.dL_dmean2D[3 * idx + c] = 0.f;
        }
        a.dL_dopacity[idx] = 0.f;
        if (a.dL_dtransMat)
            for (int c = 0; c < 9; c++) a.dL_dtransMat[9 * idx + c] = 0.f;
        a.dL_dscale[2 * idx] = 0.f;
        a.dL_dscale[2 * idx + 1] = 0.f;
        reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a.shs) {
        // only visible surfels and only the first (D+1)^2 coefficients are written, as in the reference
        __syncthreads();
        static_assert(kSurfelBlock == 64, "one wave per workgroup: the vote below is the workgroup's visibility mask");
        const unsigned long long vis = __ballot(active);   // (was one radii load per element of the loop below)
        const int touched = 3 * (a.D + 1) * (a.D + 1);
        float* dst = a.dL_dsh + (size_t)base * row;
        const int total = rows_here * row;
        if (a.sh_all_rows && (row & 3) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
            // option 8 (every element is stored: the caller's buffer needs no fill -- a gradient bucket that is stored, not added to):
            // 16 bytes per lane and store
            float4* dst4 = reinterpret_cast<float4*>(dst);
            for (int v = threadIdx.x; v < (total >> 2); v += kSurfelBlock) {
                const int e = 4 * v, r = (int)__umulhi((unsigned)e, a.row_inv), c = e - r * row;
                const float* sp = s_sh + (r * stride + c);
                const bool rv = (vis >> r) & 1ull;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = (rv && c + k < touched) ? sp[k] : 0.f;
                dst4[v] = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int e = threadIdx.x; e < total; e += kSurfelBlock) {
                const int r = (int)__umulhi((unsigned)e, a.row_inv);
                const int c = e - r * row;
                const bool live = c < touched && ((vis >> r) & 1ull);
                if (live) dst[e] = s_sh[r * stride + c];
                else if (a.sh_all_rows) dst[e] = 0.f;   // option 8
            }
        }
    }
}

// checkFrustum (rasterizer_impl.cu:54-66): present = view.z > 0.2
__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* means3D, const float* vm, unsigned char* present)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float z = vm[2] * means3D[3 * i] + vm[6] * means3D[3 * i + 1] + vm[10] * means3D[3 * i + 2] + vm[14];
    present[i] = !(z <= 0.2f);
}

}  // namespace dgs
