// hostmath.cpp -- compiles the PRODUCT arithmetic header (dynamic-2dgs_amd/csrc/surfel_math.h) for the
// host with g++ and drives it with plain loops, so the CPU test-suite can compare the exact code the
// HIP kernels inline against the oracle without a GPU.  Test infrastructure only: the shipped library
// never runs this; control flow of the kernels (LDS staging, wave reductions, atomics) is NOT covered
// here and is checked by the -m gpu tests.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../dynamic-2dgs_amd/csrc/surfel_math.h"

using namespace dgs;

static Camera make_cam(const float* view, const float* campos, int W, int H, float tx, float ty)
{
    Camera c;
    c.view = view; c.campos = campos;
    c.focal_y = H / (2.0f * ty); c.focal_x = W / (2.0f * tx);
    c.tan_fovx = tx; c.tan_fovy = ty; c.width = W; c.height = H;
    c.tiles_x = (W + 15) / 16; c.tiles_y = (H + 15) / 16;
    return c;
}

static Quad Q(const float* r, int i) { return Quad{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]}; }

extern "C" {

void hm_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rots, const float* opac,
                   const float* shs, const float* colors_precomp, const float* view, const float* campos, int W, int H, float tx,
                   float ty, int* radii, float* rec /*[P,20]*/, int* tiles, uint32_t* rects /*[P,2]*/, int tight)
{
    Camera cam = make_cam(view, campos, W, H, tx, ty);
    for (int i = 0; i < P; i++) {
        SurfelRec r;
        std::memset(&r, 0, sizeof(r));
        int t = 0;
        TileRect tr;
        radii[i] = preprocess_surfel(cam, means3D + 3 * i, scales + 2 * i, rots + 4 * i, opac[i], D,
                                     colors_precomp ? nullptr : shs + (size_t)i * M * 3, colors_precomp ? colors_precomp + 3 * i : nullptr, r, t, tr,
                                     tight != 0);
        tiles[i] = t;
        rects[2 * i] = tr.xs; rects[2 * i + 1] = tr.ys;
        std::memcpy(rec + (size_t)i * kRecFloats, &r, sizeof(r));
    }
}

// Bit w of the result: some pixel of quadrant w (8x8, lane_pixel) of tile (tx,ty) passes the alpha test for this record -- in the
// reference's form (pair_eval) OR in the affine form the kernels evaluate (alpha_affine + alpha_depth).
// Bits 4..7: the quadrant mask the blend kernels use (box test refined by quad_hit_affine); bits 8..11: the box test alone.
// -1: the branchy and branch-free reference-form evaluations disagree; -2 / -3: a quadrant is reachable although the record's
// bounding box (q5) / the conic refinement says it is not (quadrant culling of the blend kernels would be wrong); -4: the same for
// a 4x4 block under blocks_hit_linear (the row-per-block kernels' test).
int hm_tile_reachable(int W, int H, int tx, int ty, const float* r)
{
    int any = 0;
    uint32_t any16 = 0;   // bit (4 (ly / 4) + lx / 4): some pixel of that 4x4 block passes
    const float X0 = (float)(tx * 16) + 8.0f, Y0 = (float)(ty * 16) + 8.0f;
    const TileAffine ta = tile_affine(Q(r, 0), Q(r, 1), Q(r, 2), X0, Y0);
    for (int ly = 0; ly < 16; ly++)
        for (int lx = 0; lx < 16; lx++) {
            const int px = tx * 16 + lx, py = ty * 16 + ly;
            if (px >= W || py >= H) continue;
            PairEval a, b;
            const bool ka = pair_eval((float)px + 0.5f, (float)py + 0.5f, Q(r, 0), Q(r, 1), Q(r, 2), a);
            const bool kb = pair_eval_bf((float)px + 0.5f, (float)py + 0.5f, Q(r, 0), Q(r, 1), Q(r, 2), b);
            if (ka != kb) return -1;
            AlphaEval ev;
            bool kc = alpha_affine(kSqrt2 * ((float)lx - 7.5f), kSqrt2 * ((float)ly - 7.5f), ta.a0, ta.a1, ta.a2, ev);
            bool use3d;
            kc = kc && alpha_depth(ev, r[6], r[7], r[8], use3d) >= kNear;
            any |= (ka || kc) ? (1 << ((ly >> 3) * 2 + (lx >> 3))) : 0;
            any16 |= (ka || kc) ? (1u << ((ly >> 2) * 4 + (lx >> 2))) : 0u;
        }
    // the row-per-block kernels' test (blocks_hit_linear): bit b of quadrant w = block (b & 1, b >> 1) of the quadrant
    for (int w = 0; w < 4; w++) {
        const float qx = (float)(tx * 16 + 8 * (w & 1)), qy = (float)(ty * 16 + 8 * (w >> 1));
        const uint32_t bm = blocks_hit_linear(ta, kSqrt2 * ((w & 1) ? 0.5f : -7.5f), kSqrt2 * ((w & 2) ? 0.5f : -7.5f), Quad{r[20], r[21], r[22], r[23]}, qx, qy);
        for (int b = 0; b < 4; b++) {
            const int bxi = 2 * (w & 1) + (b & 1), byi = 2 * (w >> 1) + (b >> 1);
            if (((any16 >> (byi * 4 + bxi)) & 1u) && !((bm >> b) & 1u)) return -4;   // a 4x4 block with a passing pixel was dropped
        }
    }
    const uint32_t box = quad_mask(r[20], r[21], r[22], r[23], (float)(tx * 16), (float)(ty * 16));
    uint32_t mask = 0;
    for (int w = 0; w < 4; w++) mask |= (((box >> w) & 1u) && quad_hit_affine(ta, w)) ? (1u << w) : 0u;
    if (any & ~(int)box) return -2;
    if (any & ~(int)mask) return -3;   // the conic refinement dropped a quadrant that holds a passing pixel
    return any | ((int)mask << 4) | ((int)box << 8);
}

// planar [c,H,W] outputs like the reference's image state.  Follows blend_fwd_kernel: per tile, every entry is turned into its
// tile-relative affine image (tile_affine) and the pixels evaluate alpha_affine / alpha_depth / pixfwd_blend_affine; a pixel that
// saturates is poisoned with NaN coordinates, as in the kernel.
void hm_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* rec, const float* bg,
                  float* out_color, float* out_others, float* final_T, uint32_t* n_contrib)
{
    const int tiles_x = (W + 15) / 16, HW = W * H;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tx = px / 16, ty = py / 16, tile = ty * tiles_x + tx;
            const float X0 = (float)(tx * 16) + 8.0f, Y0 = (float)(ty * 16) + 8.0f;
            float u = (float)(px - tx * 16) - 7.5f, v = (float)(py - ty * 16) - 7.5f;
            float us = kSqrt2 * u, vs = kSqrt2 * v;
            PixFwd st;
            pixfwd_init(st);
            for (uint32_t e = ranges[2 * tile]; e < ranges[2 * tile + 1]; e++) {
                const float* r = rec + (size_t)point_list[e] * kRecFloats;
                st.contributor++;
                const TileAffine ta = tile_affine(Q(r, 0), Q(r, 1), Q(r, 2), X0, Y0);
                AlphaEval ev;
                if (!alpha_affine(us, vs, ta.a0, ta.a1, ta.a2, ev)) continue;
                bool use3d;
                const float depth = alpha_depth(ev, r[6], r[7], r[8], use3d);
                if (!(depth >= kNear)) continue;
                float w, test_T;
                pixfwd_weight(st, ev.alpha, w, test_T);
                if (test_T < kTmin) { us = NAN; continue; }
                pixfwd_accumulate<true>(st, w, test_T, depth, Q(r, 3), Q(r, 4));
            }
            const int pix = py * W + px;
            final_T[pix] = st.T; final_T[HW + pix] = st.dist1; final_T[2 * HW + pix] = st.dist2;
            n_contrib[pix] = st.last; n_contrib[HW + pix] = st.med_c;
            for (int c = 0; c < 3; c++) out_color[c * HW + pix] = st.C[c] + st.T * bg[c];
            out_others[pix] = st.D; out_others[HW + pix] = 1.f - st.T;
            for (int c = 0; c < 3; c++) out_others[(2 + c) * HW + pix] = st.N[c];
            out_others[5 * HW + pix] = st.med_d; out_others[6 * HW + pix] = st.distortion; out_others[7 * HW + pix] = st.med_w;
        }
}

long hm_blend_bwd(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* rec, const float* bg,
                  const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dothers,
                  float* acc /*[P,20]*/)
{
    const int tiles_x = (W + 15) / 16, HW = W * H;
    std::vector<double> dacc((size_t)P * kAccFloats, 0.0);
    long bad = 0;   // non-zero (or NaN) outputs from a pixel that does not blend the entry
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tx = px / 16, ty = py / 16, tile = ty * tiles_x + tx;
            const int pix = py * W + px;
            const float X0 = (float)(tx * 16) + 8.0f, Y0 = (float)(ty * 16) + 8.0f;
            const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
            const float us = kSqrt2 * ((float)(px - tx * 16) - 7.5f), vs = kSqrt2 * ((float)(py - ty * 16) - 7.5f);
            float gp[3], go[8];
            for (int c = 0; c < 3; c++) gp[c] = dL_dpix[c * HW + pix];
            for (int c = 0; c < 8; c++) go[c] = dL_dothers[c * HW + pix];
            PixBwdA st;
            pixbwd_init_affine(st, final_T[pix], final_T[HW + pix], final_T[2 * HW + pix], (int)n_contrib[pix], (int)n_contrib[HW + pix], gp, go, bg);
            const uint32_t r0 = ranges[2 * tile];
            for (int e = st.last_contributor - 1; e >= 0; e--) {
                const uint32_t id = point_list[r0 + (uint32_t)e];
                const float* r = rec + (size_t)id * kRecFloats;
                const TileAffine ta = tile_affine(Q(r, 0), Q(r, 1), Q(r, 2), X0, Y0);
                AlphaEval ev;
                // like the kernel: entries the pixel does not blend go through the same step with ok = false
                bool ok = alpha_affine(us, vs, ta.a0, ta.a1, ta.a2, ev);
                bool use3d;
                const float depth = alpha_depth(ev, r[6], r[7], r[8], use3d);
                ok = ok & (depth >= kNear);
                float out[16], out2d[2];
                pixbwd_step_affine(st, ev, ok, use3d, depth, e, pfx, pfy, r[6], r[7], Quad{r[0], r[1], r[3], r[4]}, r[11], Q(r, 3), Q(r, 4), out, out2d);
                for (int c = 0; c < 16; c++) dacc[(size_t)id * kAccFloats + c] += out[c];
                dacc[(size_t)id * kAccFloats + kAccMean2D] += out2d[0];
                dacc[(size_t)id * kAccFloats + kAccMean2D + 1] += out2d[1];
                if (!ok) for (int c = 0; c < 18; c++) if (c < 16 ? out[c] != 0.f : out2d[c - 16] != 0.f) bad++;
            }
        }
    for (size_t i = 0; i < dacc.size(); i++) acc[i] = (float)dacc[i];
    return bad;
}

void hm_surfel_bwd(int P, int D, int M, const float* means3D, const float* scales, const float* rots, const float* shs,
                   const float* view, const float* campos, int W, int H, float tx, float ty, const int* radii, const float* rec,
                   const float* acc, float* dmean2D /*[P,3]*/, float* dmean3D, float* dT, float* dsh, float* dscale, float* drot)
{
    Camera cam = make_cam(view, campos, W, H, tx, ty);
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        SurfelRec r;
        std::memcpy(&r, rec + (size_t)i * kRecFloats, sizeof(r));
        SurfelGrads g;
        surfel_backward(cam, means3D + 3 * i, scales + 2 * i, rots + 4 * i, r, acc + (size_t)i * kAccFloats, g);
        if (shs) sh_backward(D, shs + (size_t)i * M * 3, means3D + 3 * i, campos, r.flags, acc + (size_t)i * kAccFloats + kAccColor,
                             dsh + (size_t)i * M * 3, g.dmean3D);
        for (int c = 0; c < 3; c++) dmean3D[3 * i + c] = g.dmean3D[c];
        dmean2D[3 * i] = g.dmean2D[0]; dmean2D[3 * i + 1] = g.dmean2D[1];
        for (int c = 0; c < 9; c++) dT[9 * i + c] = g.dT[c];
        dscale[2 * i] = g.dscale[0]; dscale[2 * i + 1] = g.dscale[1];
        for (int c = 0; c < 4; c++) drot[4 * i + c] = g.drot[c];
    }
}

}  // extern "C"

// ---- analysis helper (development): how much of the traversed work is useful?
static int g_shape = 1;   // wave shape for hm_blend_stats: 0 = 16x4 strips (round-1 layout), 1 = 8x8 quadrants (the kernels' layout)
extern "C" void hm_set_shape(int s) { g_shape = s; }
static inline int lane_of(int w, int k)   // pixel index (y*16 + x) of lane k of wave w
{
    if (g_shape >= 1) { const int x = 8 * (w & 1) + (k & 7), y = 8 * (w >> 1) + (k >> 3); return y * 16 + x; }
    return 64 * w + k;
}
static inline bool wave_sees(int w, const float* r, float px0, float py0)
{
    if (g_shape == 1) return ((quad_mask(r[20], r[21], r[22], r[23], px0, py0) >> w) & 1u) && quad_hit_affine(tile_affine(Q(r, 0), Q(r, 1), Q(r, 2), px0 + 8.0f, py0 + 8.0f), w);
    if (g_shape == 2) {   // 8x8 quadrants, bounding box only (before the conic refinement)
        const float xf = px0 + 8.0f * (w & 1) + 0.5f, yf = py0 + 8.0f * (w >> 1) + 0.5f;
        return r[21] >= xf && r[20] <= xf + 7.0f && r[23] >= yf && r[22] <= yf + 7.0f;
    }
    const float yf = py0 + 4.0f * w + 0.5f;
    return r[21] >= px0 + 0.5f && r[20] <= px0 + 15.5f && r[23] >= yf && r[22] <= yf + 3.0f;
}
// Iteration counts of a blend kernel whose 16-lane rows walk their own 4x4 block's list (round-4 design study).  Per (tile,
// quadrant wave): the list is staged in chunks of `chunk` entries; an entry goes on the list of block b (the four 4x4 blocks of the
// quadrant) when its pixel box and block_hit_affine at 4x4 granularity say it can reach the block and the block still has a live
// pixel at the start of the chunk; the wave then runs max_b(list length) iterations.  out[0] = visits of today's kernel (one per
// entry that hits the quadrant), out[1] = row iterations with rows synchronised per chunk, out[2] = with rows free to drift
// (max over blocks of the whole-list totals), out[3] = sum of (entry, block) pairs / 4 (perfect balance), out[4] = blending
// lane-visits (pixels that blend), out[5] = (entry, block) pairs, out[6] = (entry, block) pairs with a passing live pixel that the
// test dropped (must be 0).  linear = 1: the kernels' test (blocks_hit_linear, which replaces the quadrant test as well);
// 0: quadrant test + the exact minimum per block (block_hit_affine).
extern "C" void hm_row_stats(int W, int H, int chunk, int linear, const uint32_t* ranges, const uint32_t* point_list, const float* rec, double* out /*8*/)
{
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    double visits = 0, it_sync = 0, it_free = 0, pairs = 0, blend_px = 0, missed = 0;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            const int tile = ty * tiles_x + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float px0 = (float)(tx * 16), py0 = (float)(ty * 16);
            for (int w = 0; w < 4; w++) {
                PixFwd st[64]; bool done[64];
                for (int k = 0; k < 64; k++) {
                    pixfwd_init(st[k]);
                    const int x = 8 * (w & 1) + (k & 7), y = 8 * (w >> 1) + (k >> 3);
                    done[k] = !(tx * 16 + x < W && ty * 16 + y < H);
                }
                double tot_b[4] = {0, 0, 0, 0};
                for (uint32_t base = r0; base < r1; base += chunk) {
                    int alive = 0, balive[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 64; k++) if (!done[k]) { alive++; balive[((k >> 3) >> 2) * 2 + ((k & 7) >> 2)] = 1; }
                    if (!alive) break;
                    int nb[4] = {0, 0, 0, 0};
                    for (uint32_t e = base; e < r1 && e < base + chunk; e++) {
                        const float* r = rec + (size_t)point_list[e] * kRecFloats;
                        const TileAffine ta = tile_affine(Q(r, 0), Q(r, 1), Q(r, 2), px0 + 8.0f, py0 + 8.0f);
                        const bool qhit = ((quad_mask(r[20], r[21], r[22], r[23], px0, py0) >> w) & 1u) && quad_hit_affine(ta, w);
                        uint32_t bm = 0;
                        if (linear) {
                            const float qx = px0 + 8.0f * (w & 1), qy = py0 + 8.0f * (w >> 1);
                            bm = blocks_hit_linear(ta, kSqrt2 * ((w & 1) ? 0.5f : -7.5f), kSqrt2 * ((w & 2) ? 0.5f : -7.5f), Quad{r[20], r[21], r[22], r[23]}, qx, qy);
                        } else if (qhit) {
                            for (int b = 0; b < 4; b++) {
                                const float bx = 8.0f * (w & 1) + 4.0f * (b & 1), by = 8.0f * (w >> 1) + 4.0f * (b >> 1);   // first pixel of the block in the tile
                                const float xf = px0 + bx + 0.5f, yf = py0 + by + 0.5f;
                                if (!(r[21] >= xf && r[20] <= xf + 3.0f && r[23] >= yf && r[22] <= yf + 3.0f)) continue;
                                const float us0 = kSqrt2 * (bx - 7.5f), vs0 = kSqrt2 * (by - 7.5f);
                                if (block_hit_affine(ta, us0, us0 + 3.0f * kSqrt2, vs0, vs0 + 3.0f * kSqrt2)) bm |= 1u << b;
                            }
                        }
                        if (qhit) visits += 1;
                        for (int b = 0; b < 4; b++)
                            if (balive[b] && ((bm >> b) & 1u)) nb[b]++;
                        for (int k = 0; k < 64; k++) {
                            if (done[k]) continue;
                            const int x = 8 * (w & 1) + (k & 7), y = 8 * (w >> 1) + (k >> 3);
                            PairEval ev;
                            if (pair_eval_bf(px0 + x + 0.5f, py0 + y + 0.5f, Q(r, 0), Q(r, 1), Q(r, 2), ev)) {
                                if (!((bm >> (((k >> 3) >> 2) * 2 + ((k & 7) >> 2))) & 1u)) missed += 1;
                                blend_px += 1;
                                st[k].contributor = e - r0 + 1;
                                if (!pixfwd_blend(st[k], ev, Q(r, 3), Q(r, 4))) done[k] = true;
                            }
                        }
                    }
                    int mx = 0;
                    for (int b = 0; b < 4; b++) { mx = nb[b] > mx ? nb[b] : mx; tot_b[b] += nb[b]; pairs += nb[b]; }
                    it_sync += mx;
                }
                double mt = 0;
                for (int b = 0; b < 4; b++) mt = tot_b[b] > mt ? tot_b[b] : mt;
                it_free += mt;
            }
        }
    out[0] = visits; out[1] = it_sync; out[2] = it_free; out[3] = pairs / 4.0; out[4] = blend_px; out[5] = pairs; out[6] = missed;
}

extern "C" void hm_blend_stats(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* rec, double* out /*16*/)
{
    // per (tile, entry, 16x4 strip) with the wave still alive: visited = the entry's pixel box touches the strip (what the
    // kernels evaluate); any = some pixel passes the alpha test; blend = some live pixel blends it
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    double S = 0, strip_pairs = 0, strip_any = 0, pix_pairs = 0, pix_pass = 0, pix_blend = 0, strip_blend_any = 0;
    double visited = 0, visited_any = 0, visited_blend = 0, visited_pix_blend = 0, missed = 0;
    double sub44 = 0, sub84 = 0, sub44_pass = 0;   // 4x4 / 8x4 sub-blocks of a blending (wave, entry) visit that hold a blending (passing) pixel
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            const int tile = ty * tiles_x + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            PixFwd st[256]; bool done[256]; bool inside[256];
            for (int i = 0; i < 256; i++) { pixfwd_init(st[i]); int px = tx * 16 + (i & 15), py = ty * 16 + (i >> 4); inside[i] = px < W && py < H; done[i] = !inside[i]; }
            uint32_t e;
            for (e = r0; e < r1; e++) {
                int alive = 0; for (int i = 0; i < 256; i++) alive += !done[i];
                if (!alive) break;
                const float* r = rec + (size_t)point_list[e] * kRecFloats;
                for (int w = 0; w < 4; w++) {
                    int wa = 0; for (int k = 0; k < 64; k++) wa += !done[lane_of(w, k)];
                    if (!wa) continue;
                    strip_pairs += 1;
                    const bool vis = wave_sees(w, r, (float)(tx * 16), (float)(ty * 16));
                    int any = 0, anyb = 0, nb = 0;
                    unsigned m44 = 0, m84 = 0, m44p = 0;
                    for (int k = 0; k < 64; k++) {
                        const int i = lane_of(w, k);
                        if (!inside[i]) continue;
                        pix_pairs += 1;
                        PairEval ev;
                        const float pfx = tx * 16 + (i & 15) + 0.5f, pfy = ty * 16 + (i >> 4) + 0.5f;
                        bool ok = pair_eval_bf(pfx, pfy, Q(r, 0), Q(r, 1), Q(r, 2), ev);
                        if (ok) { pix_pass += 1; any = 1; m44p |= 1u << (((k >> 3) >> 2) * 2 + ((k & 7) >> 2)); }
                        if (ok && !done[i]) { m44 |= 1u << (((k >> 3) >> 2) * 2 + ((k & 7) >> 2)); m84 |= 1u << ((k >> 3) >> 2); }
                        if (ok && !done[i]) { anyb = 1; nb++; pix_blend += 1; st[i].contributor = e - r0 + 1; if (!pixfwd_blend(st[i], ev, Q(r, 3), Q(r, 4))) done[i] = true; }
                    }
                    strip_any += any; strip_blend_any += anyb;
                    if (vis) { visited += 1; visited_any += any; visited_blend += anyb; visited_pix_blend += nb;
                               sub44 += __builtin_popcount(m44); sub84 += __builtin_popcount(m84); sub44_pass += __builtin_popcount(m44p); }
                    else if (anyb) missed += 1;
                }
            }
            S += (e - r0);
        }
    out[0] = S; out[1] = strip_pairs; out[2] = strip_any; out[3] = pix_pairs; out[4] = pix_pass; out[5] = pix_blend; out[6] = strip_blend_any;
    out[7] = visited; out[8] = visited_any; out[9] = visited_blend; out[10] = visited_pix_blend; out[11] = missed;
    out[12] = sub44; out[13] = sub84; out[14] = sub44_pass;
}
