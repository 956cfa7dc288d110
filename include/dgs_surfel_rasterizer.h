/*
 * dgs_surfel_rasterizer.h -- C ABI of the MI355X-native differentiable 2D-Gaussian surfel rasterizer.
 *
 * This is the drop-in boundary for the hot path of hustvl/Dynamic-2DGS: the three entry points below
 * are what the reference's Python extension binds through
 *     submodules/diff-surfel-rasterization/cuda_rasterizer/rasterizer.h:20-87
 *         CudaRasterizer::Rasterizer::{markVisible, forward, backward}
 * (called from rasterize_points.cu:39-141, :143-240, :242-261).  Argument order and meaning follow
 * that header one for one; the only changes a C ABI forces are
 *   - std::function<char*(size_t)> allocator closures  ->  (dgs_alloc_fn, void* ctx) pairs,
 *   - an explicit HIP stream (the reference launches on the legacy default stream),
 *   - bool -> int, exceptions -> negative return code + dgs_last_error().
 *
 * All pointers are DEVICE pointers to contiguous fp32 (int32 for radii) arrays unless noted.
 * Pointer arguments that select a mode may be NULL exactly where the reference accepts nullptr:
 *   shs XOR colors_precomp (rasterizer_impl.cu:322), radii (rasterizer_impl.cu:230-233).
 * `rotations` and `dL_drot` must be 16-byte aligned.
 *
 * The geometry / binning / image scratch buffers are private to the library but must be kept alive,
 * unmodified, between a forward call and its backward call (the reference returns them to Python for
 * exactly this reason, diff_surfel_rasterization/__init__.py:97).
 */
#ifndef DGS_SURFEL_RASTERIZER_H
#define DGS_SURFEL_RASTERIZER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_ABI_VERSION 3   /* 3: + dgs_get_option / dgs_context_get_option (round 6) */

/* Replaces std::function<char*(size_t N)> (rasterizer.h:31-33, rasterize_points.cu:31-37): must return a
 * device buffer of at least `bytes` bytes, 128-byte aligned, usable on `stream`. */
typedef char* (*dgs_alloc_fn)(void* ctx, size_t bytes);

enum dgs_status {
    DGS_OK = 0,
    DGS_ERR_INVALID_ARGUMENT = -1, /* AT_ERROR shape checks, rasterize_points.cu:61-71 */
    DGS_ERR_UNSUPPORTED = -2,      /* e.g. transMat_precomp, see DESIGN.md */
    DGS_ERR_ALLOC = -3,            /* an allocator callback returned NULL */
    DGS_ERR_HIP = -4               /* a HIP call / kernel failed (CHECK_CUDA, auxiliary.h:271-278) */
};

int dgs_abi_version(void);

/* Message for the most recent error on the calling thread ("" if none). */
const char* dgs_last_error(void);

/* CudaRasterizer::Rasterizer::markVisible, rasterizer.h:24-29 / rasterizer_impl.cu:141-153.
 * present: device array of P bytes (bool). Returns DGS_OK or a negative dgs_status. */
int dgs_rasterizer_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                unsigned char* present, void* stream);

/* CudaRasterizer::Rasterizer::forward, rasterizer.h:31-57 / rasterizer_impl.cu:198-342.
 * Returns num_rendered (>= 0) or a negative dgs_status.
 *   P surfels, D active SH degree, M SH coefficients per surfel (0 with colors_precomp)
 *   background[3]; out_color[3,H,W]; out_others[8,H,W]; radii[P] (int32, may be NULL)
 *   viewmatrix/projmatrix: 16 floats each, the reference's transposed (row-vector) matrices
 *   scale_modifier is accepted and ignored, as in the reference (forward.cu:95)
 *   transMat_precomp must be NULL (DGS_ERR_UNSUPPORTED otherwise)
 *   prefiltered: accepted; a culled surfel is skipped (the reference would __trap, auxiliary.h:177-181)
 *   debug != 0: synchronise and check after every stage. */
int dgs_rasterizer_forward(dgs_alloc_fn geometry_alloc, void* geometry_ctx, dgs_alloc_fn binning_alloc, void* binning_ctx,
                           dgs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                           int height, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                           float* out_others, int* radii, int debug, void* stream);

/* CudaRasterizer::Rasterizer::backward, rasterizer.h:59-87 / rasterizer_impl.cu:346-448.
 * R = the value forward returned.  Every dL_d* row of a VISIBLE surfel (radii > 0) is written exactly once -- stored,
 * not accumulated: the per-pixel atomics of the reference are replaced by one private 80-byte accumulator row per surfel
 * that the per-surfel kernel reads back.  The eight per-surfel arrays (everything but dL_dsh) are written for EVERY row:
 * culled surfels get zeros, so the caller needs no fill in front of the call (22 MB at 200 k surfels); with caller-zeroed
 * outputs (the reference's contract, rasterize_points.cu:194-202) nothing changes.  dL_dsh keeps the reference's rule --
 * rows of visible surfels, first (D+1)^2 coefficients, everything else untouched -- because it is the array callers
 * accumulate into: pass it zeroed, or pass the buffer the visible rows are to be overwritten in
 * (diff_surfel_rasterization.set_sh_grad_sink).  Shapes: dL_dpix[3,H,W], dL_depths[8,H,W], dL_dmean2D[P,3],
 * dL_dnormal[P,3], dL_dopacity[P], dL_dcolor[P,3], dL_dmean3D[P,3], dL_dtransMat[P,9], dL_dsh[P,M,3],
 * dL_dscale[P,2], dL_drot[P,4].  Extension: dL_dnormal, dL_dcolor and dL_dtransMat may be NULL -- "not wanted".  In the reference
 * they are intermediates of the backward (per-surfel colour / normal gradients in front of the SH and transform chains) or the
 * gradient of an input this library does not take (transMat_precomp); a caller that trains SH coefficients reads none of the
 * three, and their 60 bytes per surfel are the worst-coalesced stores of the per-surfel kernel.  Returns DGS_OK or a negative dgs_status. */
int dgs_rasterizer_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                            const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                            char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                            const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                            float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                            float* dL_drot, int debug, void* stream);

/* ---- contexts -----------------------------------------------------------------------------------------------------
 * The reference's entry points carry no state (rasterizer.h:20-87) and are re-entrant across devices and threads.  What
 * this library adds -- tile-list policy, capacity mode with its overflow flag, the pinned word of the one device->host
 * read, the kernel-timing hook -- is held by a dgs_context.  dgs_rasterizer_forward/backward and the knobs further down
 * act on a default context that exists once PER DEVICE (the calling thread's current HIP device); a caller that drives
 * one device from several threads, or wants different options side by side, creates its own contexts.  Options are
 * atomics: changing one while another thread is inside a call affects that call or the next, never corrupts it. */
typedef struct dgs_context dgs_context;
dgs_context* dgs_context_create(void);          /* bound to the calling thread's current HIP device */
void dgs_context_destroy(dgs_context* ctx);
int dgs_context_set_option(dgs_context* ctx, int key, int value);               /* keys: see dgs_set_option */
int dgs_context_get_option(dgs_context* ctx, int key);                          /* current value (>= 0) or a negative status */
int dgs_context_set_overflow_flag(dgs_context* ctx, int* device_flag);         /* see dgs_set_overflow_flag */
int dgs_context_read_overflow(dgs_context* ctx, int reset);
int dgs_context_profile_enable(dgs_context* ctx, int mode);
void dgs_context_profile_reset(dgs_context* ctx);
int dgs_context_profile_read(dgs_context* ctx, double* out, int cap);
/* dgs_rasterizer_forward / dgs_rasterizer_backward with an explicit context (same arguments after it) */
int dgs_context_forward(dgs_context* ctx, dgs_alloc_fn geometry_alloc, void* geometry_ctx, dgs_alloc_fn binning_alloc,
                        void* binning_ctx, dgs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                        int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                        float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_others, int* radii,
                        int debug, void* stream);
int dgs_context_backward(dgs_context* ctx, int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                         char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix, const float* dL_depths,
                         float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                         float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream);

/* ---- introspection used by the parity tests and bench.py (not part of the reference surface) ---- */

/* Byte offsets of the private sub-arrays inside the three scratch buffers, so tests can compare each
 * stage with the oracle.  which: 0 geometry (P), 1 image (W,H), 2 binning (R).  Writes up to `cap`
 * offsets, returns the number of sub-arrays; the last entry written is the total size.
 *   geometry: rec[P*24 f32], total[3 u32: num_rendered, longest list, overflow], internal_radii[P i32], acc[P*20 f32], rects[P uint2]
 *   image   : final_T[3*T*256 f32], n_contrib[2*T*256 u32], ranges[T uint2], tile_last[T u32], order_fwd[T u32],
 *             order_bwd[T u32], tile_counts[T u32], cursor[T u32]
 *   binning : point_list[R u32], keys[R u64: depth bits << 32 | surfel, bucketed by tile], scratch */
int dgs_debug_layout(int which, int P, int width, int height, int R, size_t* offsets, int cap);

/* Tile-list policy.  on (default): every surfel is listed only in the tiles its exact, opacity-aware screen
 * bounding box can reach (alpha >= 1/255 possible); off: the reference's square 3-sigma rectangles
 * (auxiliary.h:64-74), so that the sorted lists are entry-for-entry those of the reference.  Rendered outputs and
 * gradients are the same either way; num_rendered and the private lists differ. */
void dgs_set_tight_rects(int on);

/* Knobs (default context of the current device; defaults are the tuned values): key 0 = tight rects (0/1),
 * key 1 = blend tile order (0 row-major, 1 XCD-contiguous, 2 XCD row-interleaved, 3 longest-list-first [default], 4 XCD-local
 *         groups of 4 x 4 tiles, longest first inside an XCD: half the record re-fetches of 3 for ~15 us of extra ordering work),
 * key 2 = capacity mode: value > 0 sizes the binning buffer for `value` list entries and removes the one
 *         device->host read of the forward (rasterizer_impl.cu:281-282), which makes forward + backward legal inside
 *         hipStreamBeginCapture / torch.cuda.graph; dgs_rasterizer_forward then returns `value`.  A frame whose lists do
 *         not fit renders as background and raises the overflow flag; value 0 restores the exact-size mode,
 * key 3 = per-tile sort: 2 LSD radix sort in LDS [default], 1 bitonic network with the keys in registers, 0 bitonic network in LDS,
 * key 7 = deterministic backward (0 [default] / 1 / 2).  2: the sums are added as 64-bit fixed-point numbers (2^-44) with integer
 *         atomics -- order-free, hence bit-identical from run to run, at the default kernel's speed and legal under stream capture
 *         (the context keeps [P, 20] 64-bit rows; they must exist before a capture: run one eager backward first); partial sums are
 *         quantised to 6e-14.  1: the backward blend stores its per-(list entry, wave) sums instead of adding them
 *         with float atomics and a per-surfel kernel adds them in a fixed order -- bit-identical gradients from run to run.  For
 *         tests: R x 320 bytes of scratch (R x 1280 with the row-per-block A/B kernel) from hipMallocAsync (not capturable), a linear search per (surfel, tile).
 * key 8 = dL_dsh of the backward written for EVERY row and coefficient (0 [default]: visible rows and the active bands only, as the
 *         reference does, so callers may accumulate into it; 1: zeros elsewhere, for callers whose gradient buffer is stored, not
 *         added to, and therefore never cleared).
 * key 9 = long-tile path of the blend kernels (1 [default] / 0; tile order 3 only): the tiles at the head of the longest-first dispatch
 *         order whose list (forward) / traversed length (backward) exceeds a per-launch threshold get four workgroups -- one per 8x8
 *         quadrant, four list quarters each -- instead of one; deterministic, results differ from the serial walk in rounding only.
 * key 10 / key 11 = thresholds of that path: a forward list is long from num_rendered / value entries on (default 400; never below
 *         768), a backward tile from (sum of the traversed lengths) / value on (default 512; never below 512).
 * key 12 = binning offsets in one launch (1 [default] / 0): per-tile counts -> tile ranges, bucket write cursors, num_rendered, the
 *         capacity check and the forward's dispatch order by one kernel whose workgroups exchange their aggregates (decoupled
 *         look-back); 0 = the three launches (column pass, one-workgroup scan, column pass) it replaces.  Same results.
 * key 13 = capacity mode only (1 [default] / 0): the forward blend's dispatch order and the cleared per-tile maxima are written by one more
 *         workgroup of the key-scatter launch (next to its 256 working ones) instead of by the last workgroup of the offsets kernel
 *         (a serial tail of that launch).  Same results.
 * key 14 = accumulator rider (1 [default] / 0; tile orders 3, 4): the first workgroups of the forward blend launch zero the per-surfel
 *         accumulator rows of the backward (80 B per surfel, inside the geometry buffer) and leave a flag next to num_rendered; the
 *         backward's prep launch then skips its own clear (-5 us per step at 200 k surfels: the stores ride under a VALU-bound kernel).  The
 *         backward blend resets the flag, so a second backward over the same forward state clears the rows itself.  Same results.
 * key 6 = capacity mode only: a PROMISE that no tile list is longer than `value` entries (0 = none [default]).  Without the
 *         host read the library cannot know which of its per-tile sort kernels will find work and launches all four; with the
 *         promise it launches only those for lists up to `value` (2048: one launch; 57344 = 28 segments of 2048: three).  A frame that breaks the promise is treated exactly like a capacity overflow: background, flag raised,
 * key 4 / key 5 = diagnostic: the backward / forward blend processes only the first `value` tiles of its dispatch order
 *         (0 = all); results are then incomplete -- for measuring how long the heaviest tiles run on an otherwise idle device.
 * Returns DGS_OK or an error. */
int dgs_set_option(int key, int value);
/* The current value of an option of the calling thread's device's default context (>= 0), or a negative status: a caller that
 * switches an option for a while puts back what it found (the context is shared by everything that renders on the device). */
int dgs_get_option(int key);

/* The capacity-overflow flag is one int32 in device memory, OR-ed by the forward whose lists did not fit with the reason: bit 0 the
 * lists exceed the capacity, bit 1 a list is longer than promised (option 6), bit 2 that list is also beyond the segmented sort's
 * 57 344 entries (the caller can pick its next capacity / promise from the bits; any non-zero value = the frame rendered as background).  By default
 * the library owns it (dgs_read_overflow).  A trainer that must not act on such a frame hands in its own flag here and lets
 * its optimiser kernels read it on the device (skip the update) -- no host round trip; NULL returns to the library's. */
int dgs_set_overflow_flag(int* device_flag);

/* Non-zero (the reason bits above) if a capacity overflow happened since the last reset (blocking device read; call it outside hot loops). */
int dgs_read_overflow(int reset);

/* Kernel timing hook for bench.py.  mode 1: the library brackets its kernels with HIP events on the launch stream (eager
 * launches only); mode 2: with one-thread kernels that append the device's constant-rate (100 MHz) counter to a ring --
 * legal inside a captured graph, so the kernels are timed in the launch mode the replayed step uses; mode 0: off.
 * dgs_profile_read returns accumulated milliseconds and launch counts since the last reset (up to 14 values):
 * out[0..1] fwd blend (ms, n), out[2..3] bwd blend (ms, n), out[4], out[5] = sum over the timed fwd / bwd launches of
 * S = sum_tiles(entries traversed), the unit of the blend kernels' algorithmic-bytes formula;
 * out[6..7] preprocess_fwd (ms, n), out[8..9] binning = count + scan + scatter + per-tile sort (ms, n; kernel time only in
 * capacity mode -- in exact-size mode the span also holds the host's read of num_rendered and the binning allocator), out[10..11] surfel_bwd
 * (ms, n), out[12] = sum of num_rendered, out[13] = sum of visible surfels (radii > 0) over the timed forwards. */
void dgs_profile_enable(int mode);
void dgs_profile_reset(void);
int dgs_profile_read(double* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* DGS_SURFEL_RASTERIZER_H */
