// kernels_preprocess.h -- per-surfel kernels (gfx950): forward preprocess + tile count block sums,
// block-sum scan, key emission, tile ranges, fused per-surfel backward, frustum mark.
// Replaces preprocessCUDA / InclusiveSum / duplicateWithKeys / identifyTileRanges / computeAABB-bwd /
// preprocessCUDA-bwd / checkFrustum of the reference (forward.cu:166-260, rasterizer_impl.cu:54-138,278,
// backward.cu:533-649).  One thread per surfel, 256 threads (4 wave64) per workgroup.
#pragma once
#include <hip/hip_runtime.h>

#include "surfel_math.h"

namespace dgs {

constexpr int kSurfelBlock = 256;

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
    uint32_t* block_sums;   // [ceil(P/256)] out: sum of tile counts of the block
    int tight;              // 1: exact opacity-aware rectangles (default), 0: the reference's rectangles
};

// wave64 inclusive scan with DPP-friendly shuffles
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// exclusive scan over the 256 threads of a workgroup; returns the exclusive prefix, total in `total`
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) base += (w < wave) ? s_wave[w] : 0u;
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

__global__ void __launch_bounds__(kSurfelBlock) preprocess_fwd_kernel(PreprocessArgs a)
{
    __shared__ uint32_t s_wave[4];
    const int idx = blockIdx.x * kSurfelBlock + threadIdx.x;
    int tiles = 0;
    if (idx < a.P) {
        SurfelRec rec;
        float pos[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        float sc[2] = {a.scales[2 * idx], a.scales[2 * idx + 1]};
        const float4 qv = reinterpret_cast<const float4*>(a.rotations)[idx];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        const float* sh = a.colors_precomp ? nullptr : a.shs + (size_t)idx * a.M * 3;
        const float* cp = a.colors_precomp ? a.colors_precomp + 3 * idx : nullptr;
        TileRect tr;
        int radius = preprocess_surfel(a.cam, pos, sc, q, a.opacities[idx], a.D, sh, cp, rec, tiles, tr, a.tight != 0);
        a.radii[idx] = radius;
        a.rects[idx] = make_uint2(tr.xs, tr.ys);
        if (radius > 0) {
            const float4* src = reinterpret_cast<const float4*>(&rec);
            float4* dst = a.rec + (size_t)idx * kRecQuads;
#pragma unroll
            for (int c = 0; c < kRecQuads; c++) dst[c] = src[c];
        }
    }
    uint32_t total;
    block_exclusive_scan((uint32_t)tiles, s_wave, total);
    if (threadIdx.x == 0) a.block_sums[blockIdx.x] = total;
}

// Single-workgroup exclusive scan of the block sums (P/256 values: 782 at 200k surfels, 3907 at 1M).
// In place; the grand total (num_rendered, rasterizer_impl.cu:281) goes to *total_out.
__global__ void __launch_bounds__(kSurfelBlock) scan_block_sums_kernel(uint32_t* sums, int n, uint32_t* total_out)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (int base = 0; base < n; base += kSurfelBlock) {
        const int i = base + threadIdx.x;
        uint32_t v = i < n ? sums[i] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, s_wave, total);
        if (i < n) sums[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

struct EmitArgs {
    int P;
    const int* radii;
    const float4* rec;
    const uint2* rects;
    const uint32_t* block_offsets;  // exclusive scan of block sums
    uint64_t* keys;                 // [R]
    uint32_t* vals;                 // [R]
    int tiles_x, tiles_y;
};

// duplicateWithKeys (rasterizer_impl.cu:70-111): key = tile << 32 | depth bits, value = surfel index.
// The per-surfel offsets are recomputed from the block offset + an in-block scan instead of a
// materialised inclusive-sum array.
__global__ void __launch_bounds__(kSurfelBlock) emit_keys_kernel(EmitArgs a)
{
    __shared__ uint32_t s_wave[4];
    const int idx = blockIdx.x * kSurfelBlock + threadIdx.x;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint32_t depth_bits = 0;
    if (idx < a.P) {
        const int radius = a.radii[idx];
        if (radius > 0) {
            const uint2 r = a.rects[idx];
            x0 = (int)(r.x & 0xffffu); x1 = (int)(r.x >> 16);
            y0 = (int)(r.y & 0xffffu); y1 = (int)(r.y >> 16);
            if (x1 > x0 && y1 > y0) depth_bits = __float_as_uint(a.rec[(size_t)idx * kRecQuads + 4].z);
        }
    }
    const uint32_t cnt = (uint32_t)((x1 - x0) * (y1 - y0));
    uint32_t total;
    uint32_t off = a.block_offsets[blockIdx.x] + block_exclusive_scan(cnt, s_wave, total);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            uint64_t key = (uint64_t)(uint32_t)(y * a.tiles_x + x);
            key = (key << 32) | depth_bits;
            a.keys[off] = key;
            a.vals[off] = (uint32_t)idx;
            off++;
        }
}

// identifyTileRanges (rasterizer_impl.cu:116-138) on the sorted keys; `ranges` pre-zeroed.
__global__ void __launch_bounds__(256) tile_ranges_kernel(int R, const uint64_t* keys, uint2* ranges)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (cur != prev) { ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
    }
    if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

struct SurfelBwdArgs {
    int P, D, M;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* shs;            // null when colours were precomputed
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
    float* dL_dscale;            // [P,2]
    float* dL_drot;              // [P,4]
};

// Fused computeAABB-bwd + preprocessCUDA-bwd (backward.cu:533-649).  Outputs are caller-zeroed
// (rasterize_points.cu:194-202); culled surfels are left untouched.
__global__ void __launch_bounds__(kSurfelBlock) surfel_bwd_kernel(SurfelBwdArgs a)
{
    const int idx = blockIdx.x * kSurfelBlock + threadIdx.x;
    if (idx >= a.P || !(a.radii[idx] > 0)) return;
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
        // written straight to global memory: only the first (D+1)^2 coefficients are touched
        sh_backward(a.D, a.shs + (size_t)idx * a.M * 3, pos, a.cam.campos, rec.flags, acc + kAccColor,
                    a.dL_dsh + (size_t)idx * a.M * 3, g.dmean3D);
    }
    for (int c = 0; c < 3; c++) {
        a.dL_dmean3D[3 * idx + c] = g.dmean3D[c];
        a.dL_dcolor[3 * idx + c] = acc[kAccColor + c];
        a.dL_dnormal[3 * idx + c] = acc[kAccNormal + c];
    }
    a.dL_dmean2D[3 * idx] = g.dmean2D[0];
    a.dL_dmean2D[3 * idx + 1] = g.dmean2D[1];
    a.dL_dopacity[idx] = acc[kAccOpacity];
    for (int c = 0; c < 9; c++) a.dL_dtransMat[9 * idx + c] = g.dT[c];
    a.dL_dscale[2 * idx] = g.dscale[0];
    a.dL_dscale[2 * idx + 1] = g.dscale[1];
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(g.drot[0], g.drot[1], g.drot[2], g.drot[3]);
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
