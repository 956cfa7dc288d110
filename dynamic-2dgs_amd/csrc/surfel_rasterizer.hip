// surfel_rasterizer.hip -- host orchestration + C ABI (include/dgs_surfel_rasterizer.h) of the
// MI355X surfel rasterizer.  Stage order follows CudaRasterizer::Rasterizer::forward / backward
// (rasterizer_impl.cu:198-342, :346-448 of the reference); everything runs on the caller's stream.
//
// State.  The reference's entry points are stateless and re-entrant across devices.  Everything this
// library adds on top (tile-list policy, capacity mode and its overflow flag, the pinned word of the one device->host
// read, the kernel-timing hook) lives in a dgs_context; the reference-shaped entry points use one lazily created default
// context PER DEVICE, so two devices -- or two threads with their own contexts -- never share mutable state.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dgs_surfel_rasterizer.h"
#include "kernels_blend.h"
#include "kernels_preprocess.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define DGS_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(DGS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));             \
    } while (0)

// CHECK_CUDA equivalent (auxiliary.h:271-278): with debug, synchronise after the stage
#define DGS_STAGE(name, debug, stream)                                                                 \
    do {                                                                                               \
        hipError_t e__ = hipGetLastError();                                                            \
        if (e__ == hipSuccess && (debug)) e__ = hipStreamSynchronize(stream);                          \
        if (e__ != hipSuccess) return fail(DGS_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e__)); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {  // 128-byte aligned sub-allocation, like obtain() in rasterizer_impl.h:22-27
    size_t off = 0;
    size_t take(size_t bytes)
    {
        off = align_up(off, 128);
        size_t o = off;
        off += bytes;
        return o;
    }
};

struct GeomLayout {
    size_t rec, total, internal_radii, acc, rects, bytes;
    int nblocks;
    explicit GeomLayout(int P)
    {
        Carver c;
        nblocks = (P + dgs::kSurfelBlock - 1) / dgs::kSurfelBlock;
        rec = c.take((size_t)P * dgs::kRecFloats * 4);
        total = c.take(16);  // num_rendered, longest tile list, overflow flag
        internal_radii = c.take((size_t)P * 4);
        acc = c.take((size_t)P * dgs::kAccFloats * 4);
        rects = c.take((size_t)P * 8);
        bytes = align_up(c.off, 128);
    }
};

struct ImageLayout {
    size_t final_T, n_contrib, ranges, tile_last, order_fwd, order_bwd, group_xcd, tile_counts, cursor, long_thr, bytes;
    bool lds_bins;  // per-workgroup LDS histograms fit (T * 4 bytes <= 144 KB)
    int tiles_x, tiles_y, ntiles;
    ImageLayout(int W, int H)
    {
        tiles_x = (W + dgs::kTileX - 1) / dgs::kTileX;
        tiles_y = (H + dgs::kTileY - 1) / dgs::kTileY;
        ntiles = tiles_x * tiles_y;
        Carver c;
        const size_t plane = (size_t)ntiles * dgs::kTilePix;
        final_T = c.take(3 * plane * 4);
        n_contrib = c.take(2 * plane * 4);
        ranges = c.take((size_t)ntiles * 8);
        tile_last = c.take((size_t)ntiles * 4);
        const size_t order_len = (size_t)std::max(ntiles, dgs::order_slots(tiles_x, tiles_y));   // tile order 4 has empty slots
        order_fwd = c.take(order_len * 4);
        order_bwd = c.take(order_len * 4);
        group_xcd = c.take((size_t)dgs::kOrderMaxGroups * 4);   // tile order 4: the forward's group -> XCD map, reused by the backward
        // global-atomics path: the T tile counts; LDS-histogram path: the 2 ceil(T / 64) + 1 words bin_offsets_kernel's workgroups meet in
        tile_counts = c.take((size_t)(ntiles > 2 * ((ntiles + 63) / 64) + 1 ? ntiles : 2 * ((ntiles + 63) / 64) + 1) * 4);
        lds_bins = (size_t)ntiles * 4 <= 144 * 1024;
        // LDS path: G x T matrix of per-workgroup counts / cursors; fallback: T global cursors
        cursor = c.take(lds_bins ? (size_t)dgs::kBinGroups * ntiles * 4 : (size_t)ntiles * 4);
        long_thr = c.take(sizeof(dgs::LongThr));   // thresholds of the long-tile path (forward: scan_tiles_kernel, backward: prep_bwd_kernel)
        bytes = align_up(c.off, 128);
    }
};

struct BinningLayout {
    size_t keys, point_list, scratch, bytes;
    // `with_scratch`: room for the global-memory fallback of the per-tile sort (2R keys); only allocated when a
    // tile list exceeds the LDS sorts' capacity.  point_list comes first so that its offset does not depend on it.
    BinningLayout(int R, bool with_scratch)
    {
        Carver c;
        const size_t n = (size_t)(R > 0 ? R : 1);
        point_list = c.take(n * 4);
        keys = c.take(n * 8);
        scratch = c.take(with_scratch ? 2 * n * 8 : 8);
        bytes = align_up(c.off, 128);
    }
};

dgs::Camera make_camera(const float* view_dev, const float* campos_dev, int W, int H, float tan_fovx, float tan_fovy)
{
    dgs::Camera cam;
    cam.view = view_dev;
    cam.campos = campos_dev;
    cam.focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:223-224
    cam.focal_x = W / (2.0f * tan_fovx);
    cam.tan_fovx = tan_fovx;
    cam.tan_fovy = tan_fovy;
    cam.width = W;
    cam.height = H;
    cam.tiles_x = (W + dgs::kTileX - 1) / dgs::kTileX;
    cam.tiles_y = (H + dgs::kTileY - 1) / dgs::kTileY;
    return cam;
}

// ---- optional kernel timing (bench.py roofline leg) ------------------------------------------------
// per timed forward: R = num_rendered and the number of visible surfels, the units of the preprocess / binning / per-surfel
// backward byte formulas
__global__ void sum_forward_units_kernel(const uint32_t* total, const int* radii, int P, unsigned long long* dst_R, unsigned long long* dst_Pv)
{
    unsigned long long vis = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) vis += radii[i] > 0 ? 1u : 0u;
    for (int d = 32; d >= 1; d >>= 1) vis += __shfl_xor(vis, d, 64);
    if ((threadIdx.x & 63) == 0 && vis) atomicAdd(dst_Pv, vis);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(dst_R, (unsigned long long)total[0]);
}

__global__ void sum_tile_last_kernel(const uint32_t* tile_last, int n, unsigned long long* dst)
{
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += tile_last[i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, acc);
}

// Profile mode 2: device timestamps instead of HIP events.  Events cannot be recorded inside a captured graph on ROCm
// (hipErrorInvalidHandle), a one-thread kernel can: it appends (constant-rate 100 MHz counter, tag) to a ring, so the
// blend kernels are timed INSIDE the replayed whole-step graph -- the launch mode the headline number uses.
constexpr unsigned kStampCap = 4096;
__global__ void stamp_kernel(unsigned long long* ring, unsigned* count, unsigned tag)
{
    const unsigned long long t = wall_clock64();
    const unsigned i = atomicAdd(count, 1u);
    ring[2 * (i % kStampCap)] = t;
    ring[2 * (i % kStampCap) + 1] = tag;
}

constexpr int kProfKinds = 5;   // 0 forward blend, 1 backward blend, 2 preprocess_fwd, 3 binning (count .. sort), 4 surfel_bwd
struct Prof {
    std::atomic<int> mode{0};                // 0 off, 1 HIP events (eager launches), 2 device timestamps (capturable)
    unsigned long long* counters = nullptr;  // device: [0] sum of S over timed fwd launches, [1] over bwd launches, [2] sum of
                                             // num_rendered, [3] sum of visible surfels (radii > 0) over timed forwards
    unsigned long long* ring = nullptr;      // device: kStampCap x (timestamp, tag)
    unsigned* ring_count = nullptr;          // device
    std::mutex mu;
    struct Pair { hipEvent_t a, b; int kind; };
    std::vector<Pair> pending;
    std::vector<Pair> pool;
    double ms[kProfKinds] = {0, 0, 0, 0, 0};
    long n[kProfKinds] = {0, 0, 0, 0, 0};
};

// Small pinned staging word for num_rendered (the one device->host read of the forward).
struct HostStage {
    uint32_t* u = nullptr;   // [2] num_rendered, longest tile list
    std::mutex mu;
    int ensure()
    {
        if (u) return 0;
        void* p = nullptr;
        if (hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess) return -1;
        u = (uint32_t*)p;
        return 0;
    }
};

}  // namespace

struct dgs_context {
    dgs_context()
    {
        if (const char* e = getenv("DGS_LONG_TILES")) long_tiles.store(atoi(e) != 0);   // A/B runs of whole programs (bench.py, the test suite)
        if (const char* e = getenv("DGS_MERGED_OFFSETS")) merged_offsets.store(atoi(e) != 0);
        if (const char* e = getenv("DGS_ORDER_RIDER")) order_rider.store(atoi(e) != 0);
        if (const char* e = getenv("DGS_ACC_RIDER")) acc_rider.store(atoi(e) != 0);
    }
    int device = 0;
    std::atomic<int> tight_rects{1};  // exact opacity-aware tile rectangles (surfel_math.h tight_tile_rect)
    std::atomic<int> sort_regs{2};    // per-tile sort: 2 LSD radix in LDS (default), 1 bitonic network in registers, 0 bitonic in LDS
    std::atomic<int> tile_order{3};   // kernels_blend.h tile_for_block (3 = longest tile first)
    std::atomic<int> deterministic{0};   // key 7: 1 backward blend without atomics, fixed summation order (tests); 2 fixed-point integer atomics
    unsigned long long* acc64 = nullptr; // key 7 = 2: [P, kAccFloats] fixed-point accumulator rows, zero between backward passes
    size_t acc64_rows = 0;
    std::vector<unsigned long long*> acc64_retired;   // outgrown rows: a captured graph may still point at them, freed with the context
    std::atomic<int> sh_all_rows{0};     // key 8: dL_dsh written for every row (zeros for culled surfels / unused bands)
    std::atomic<int> long_tiles{1};      // key 9: four workgroups (one per quadrant, four list quarters each) for the longest tiles
    // Measured (tools/diag/long_tune.py, blend fwd / bwd in ms; uniform 200k scene | ONE densified scene, 88 k surfels, kept as a checkpoint):
    //   path off 0.127 / 0.264 | 0.140 / 0.358;  forward divisor 150: 0.124 | 0.139, 400: .. | 0.130, 800: 0.125 | 0.131, 1600: .. | 0.131,
    //   3200: 0.133 | ..;  backward divisor 256: .. | 0.331, 512: 0.253 | 0.280, 1024: 0.259 | 0.281, 2048: .. | 0.283.
    // The backward of a densified scene is as long as its longest tiles and gains 22 %; its forward is closer to throughput-bound and
    // gains 7 %; an opaque knot (40 k of 100 k surfels on a few tiles) pays 2-4 % for the forward path (0.112 -> 0.114-0.116).
    std::atomic<int> long_div_fwd{400};  // key 10: a list is long from num_rendered / this (and 768 entries) on
    std::atomic<int> long_div_bwd{512};  // key 11: a tile is long from (sum of traversed lengths) / this (and 512 entries) on
    std::atomic<int> order_rider{1};     // key 13: capacity mode: tile_last + the forward's dispatch order by a rider workgroup of the scatter launch (0: by bin_offsets_kernel's last workgroup)
    std::atomic<int> merged_offsets{1};  // key 12: tile counts -> ranges, bucket cursors and dispatch order in one launch (bin_offsets_kernel); 0 = column pass, scan, column pass
    std::atomic<int> capacity{0};     // > 0: capacity mode (no host read of num_rendered; stream-capture safe)
    std::atomic<int> list_hint{0};    // capacity mode (key 6): promised longest tile list; 0 = no promise (every sort kernel is launched)
    std::atomic<int> grid_limit_bwd{0};   // > 0 (diagnostic, key 4): the backward blend processes only the first N tiles of its dispatch order
    std::atomic<bool> acc_rider{true};    // key 14: the forward blend zeroes the backward's accumulator rows (BlendFwdArgs::clear)
    std::atomic<int> grid_limit_fwd{0};   // > 0 (diagnostic, key 5): same for the forward blend (the other tiles' state is zero-filled)
    std::atomic<int*> overflow{nullptr};  // device flag raised by a capacity overflow (library- or caller-owned)
    int* overflow_owned = nullptr;
    std::mutex mu;                    // lazy allocations
    HostStage stage;
    Prof prof;
};

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_default_mu;
dgs_context* g_default[kMaxDevices] = {};

dgs_context* default_context()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default[dev]) {
        g_default[dev] = new dgs_context();
        g_default[dev]->device = dev;
    }
    return g_default[dev];
}

// the library-owned overflow flag, allocated on the context's device the first time capacity mode is switched on
int ensure_overflow(dgs_context* c)
{
    if (c->overflow.load()) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->overflow.load()) return 0;
    int* p = nullptr;
    if (hipMalloc((void**)&p, 4) != hipSuccess) return fail(DGS_ERR_HIP, "hipMalloc failed");
    (void)hipMemset(p, 0, 4);
    c->overflow_owned = p;
    c->overflow.store(p);
    return 0;
}

bool prof_begin(dgs_context* c, int kind, hipStream_t s, Prof::Pair& p)
{
    Prof& pr = c->prof;
    const int mode = pr.mode.load();
    if (mode == 0) return false;
    if (mode == 2) {
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, pr.ring, pr.ring_count, (unsigned)(2 * kind));
        p.kind = kind;
        return true;
    }
    std::lock_guard<std::mutex> lk(pr.mu);
    if (!pr.pool.empty()) {
        p = pr.pool.back();
        pr.pool.pop_back();
    } else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return false;
    }
    p.kind = kind;
    (void)hipEventRecord(p.a, s);
    return true;
}

void prof_mark_end(dgs_context* c, hipStream_t s, Prof::Pair& p)
{
    Prof& pr = c->prof;
    if (pr.mode.load() == 2) {
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, pr.ring, pr.ring_count, (unsigned)(2 * p.kind + 1));
    } else {
        (void)hipEventRecord(p.b, s);
        std::lock_guard<std::mutex> lk(pr.mu);
        pr.pending.push_back(p);
    }
}

void prof_end(dgs_context* c, hipStream_t s, Prof::Pair& p, const uint32_t* tile_last, int ntiles)
{
    prof_mark_end(c, s, p);
    // S = sum over tiles of the list length actually traversed, for the roofline
    Prof& pr = c->prof;
    if (pr.counters) hipLaunchKernelGGL(sum_tile_last_kernel, dim3(1), dim3(256), 0, s, tile_last, ntiles, pr.counters + p.kind);
}

int check_common(int P, int W, int H, const void* means3D)
{
    if (P < 0 || W <= 0 || H <= 0) return fail(DGS_ERR_INVALID_ARGUMENT, "P, width, height must be non-negative / positive");
    if (P > 0 && !means3D) return fail(DGS_ERR_INVALID_ARGUMENT, "means3D is NULL");
    return 0;
}

}  // namespace

extern "C" {

int dgs_abi_version(void) { return DGS_ABI_VERSION; }

const char* dgs_last_error(void) { return g_err.c_str(); }

// ---- contexts ---------------------------------------------------------------------------------------------------------
dgs_context* dgs_context_create(void)
{
    dgs_context* c = new dgs_context();
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) c->device = dev;
    return c;
}

void dgs_context_destroy(dgs_context* c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        for (auto& d : g_default)
            if (d == c) d = nullptr;
    }
    if (c->overflow_owned) (void)hipFree(c->overflow_owned);
    if (c->acc64) (void)hipFree(c->acc64);
    for (auto* p : c->acc64_retired) (void)hipFree(p);
    if (c->stage.u) (void)hipHostFree(c->stage.u);
    if (c->prof.counters) (void)hipFree(c->prof.counters);
    if (c->prof.ring) (void)hipFree(c->prof.ring);
    if (c->prof.ring_count) (void)hipFree(c->prof.ring_count);
    for (auto& p : c->prof.pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : c->prof.pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    delete c;
}

int dgs_context_set_option(dgs_context* c, int key, int value)
{
    if (!c) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    if (key == 0) { c->tight_rects.store(value != 0); return DGS_OK; }
    if (key == 1 && value >= 0 && value <= 4) { c->tile_order.store(value); return DGS_OK; }
    if (key == 7 && value >= 0 && value <= 2) { c->deterministic.store(value); return DGS_OK; }
    if (key == 8) { c->sh_all_rows.store(value != 0); return DGS_OK; }
    if (key == 9) { c->long_tiles.store(value != 0); return DGS_OK; }
    if (key == 10 && value > 0) { c->long_div_fwd.store(value); return DGS_OK; }
    if (key == 11 && value > 0) { c->long_div_bwd.store(value); return DGS_OK; }
    if (key == 12) { c->merged_offsets.store(value != 0); return DGS_OK; }
    if (key == 13) { c->order_rider.store(value != 0); return DGS_OK; }
    if (key == 14) { c->acc_rider.store(value != 0); return DGS_OK; }
    if (key == 3 && value >= 0 && value <= 2) { c->sort_regs.store(value); return DGS_OK; }
    if (key == 4 && value >= 0) { c->grid_limit_bwd.store(value); return DGS_OK; }
    if (key == 6 && value >= 0) { c->list_hint.store(value); return DGS_OK; }
    if (key == 5 && value >= 0) { c->grid_limit_fwd.store(value); return DGS_OK; }
    if (key == 2 && value >= 0) {
        if (value > 0)
            if (int e = ensure_overflow(c)) return e;
        c->capacity.store(value);
        if (value == 0) c->list_hint.store(0);   // the promise belongs to a capacity-mode session
        return DGS_OK;
    }
    return fail(DGS_ERR_INVALID_ARGUMENT, "dgs_set_option: unknown key / value");
}

int dgs_context_get_option(dgs_context* c, int key)
{
    if (!c) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    switch (key) {
    case 0: return c->tight_rects.load();
    case 1: return c->tile_order.load();
    case 2: return c->capacity.load();
    case 3: return c->sort_regs.load();
    case 4: return c->grid_limit_bwd.load();
    case 5: return c->grid_limit_fwd.load();
    case 6: return c->list_hint.load();
    case 7: return c->deterministic.load();
    case 8: return c->sh_all_rows.load();
    case 9: return c->long_tiles.load();
    case 10: return c->long_div_fwd.load();
    case 11: return c->long_div_bwd.load();
    case 12: return c->merged_offsets.load();
    case 13: return c->order_rider.load();
    case 14: return c->acc_rider.load() ? 1 : 0;
    }
    return fail(DGS_ERR_INVALID_ARGUMENT, "dgs_get_option: unknown key");
}

int dgs_context_set_overflow_flag(dgs_context* c, int* device_flag)
{
    if (!c) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    if (device_flag) {
        c->overflow.store(device_flag);
        return DGS_OK;
    }
    c->overflow.store(c->overflow_owned);   // back to the library-owned flag (allocated on demand)
    if (!c->overflow_owned && c->capacity.load() > 0) return ensure_overflow(c);
    return DGS_OK;
}

int dgs_context_read_overflow(dgs_context* c, int reset)
{
    if (!c) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    int* flag = c->overflow.load();
    if (!flag) return 0;
    int v = 0;
    if (hipMemcpy(&v, flag, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(DGS_ERR_HIP, "hipMemcpy failed");
    if (reset && v) (void)hipMemset(flag, 0, 4);
    return v;
}

int dgs_context_profile_enable(dgs_context* c, int mode)
{
    if (!c || mode < 0 || mode > 2) return fail(DGS_ERR_INVALID_ARGUMENT, "dgs_profile_enable: mode must be 0, 1 or 2");
    Prof& pr = c->prof;
    if (mode != 0) {   // allocate now: nothing may be allocated later, while a stream capture is in progress
        std::lock_guard<std::mutex> lk(pr.mu);
        if (!pr.counters) {
            DGS_HIP(hipMalloc((void**)&pr.counters, 32));
            DGS_HIP(hipMemset(pr.counters, 0, 32));
        }
        if (mode == 2 && !pr.ring) {
            DGS_HIP(hipMalloc((void**)&pr.ring, (size_t)kStampCap * 16));
            DGS_HIP(hipMalloc((void**)&pr.ring_count, 4));
            DGS_HIP(hipMemset(pr.ring_count, 0, 4));
        }
    }
    pr.mode.store(mode);
    return DGS_OK;
}

void dgs_context_profile_reset(dgs_context* c)
{
    if (!c) return;
    Prof& pr = c->prof;
    std::lock_guard<std::mutex> lk(pr.mu);
    for (auto& p : pr.pending) pr.pool.push_back(p);
    pr.pending.clear();
    for (int k = 0; k < kProfKinds; k++) { pr.ms[k] = 0; pr.n[k] = 0; }
    if (pr.counters) (void)hipMemset(pr.counters, 0, 32);
    if (pr.ring_count) (void)hipMemset(pr.ring_count, 0, 4);
}

int dgs_context_profile_read(dgs_context* c, double* out, int cap)
{
    if (!c || !out) return fail(DGS_ERR_INVALID_ARGUMENT, "NULL pointer");
    Prof& pr = c->prof;
    std::lock_guard<std::mutex> lk(pr.mu);
    for (auto& p : pr.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            pr.ms[p.kind] += ms;
            pr.n[p.kind] += 1;
        }
        pr.pool.push_back(p);
    }
    pr.pending.clear();
    if (pr.ring && pr.ring_count) {
        // device timestamps (mode 2): begin/end stamps come in stream order; pair them per kind
        (void)hipDeviceSynchronize();
        unsigned cnt = 0;
        (void)hipMemcpy(&cnt, pr.ring_count, 4, hipMemcpyDeviceToHost);
        if (cnt > 0) {
            const unsigned n = cnt < kStampCap ? cnt : kStampCap;
            std::vector<unsigned long long> h(2 * (size_t)kStampCap);
            (void)hipMemcpy(h.data(), pr.ring, (size_t)kStampCap * 16, hipMemcpyDeviceToHost);
            unsigned long long open_t[kProfKinds] = {0, 0, 0, 0, 0};
            bool open[kProfKinds] = {false, false, false, false, false};
            const unsigned first = cnt <= kStampCap ? 0u : cnt % kStampCap;   // oldest entry still in the ring
            for (unsigned k = 0; k < n; k++) {
                const unsigned i = (first + k) % kStampCap;
                const unsigned tag = (unsigned)h[2 * i + 1];
                const int kind = (int)(tag >> 1);
                if (kind >= kProfKinds) continue;
                if ((tag & 1u) == 0) { open_t[kind] = h[2 * i]; open[kind] = true; }
                else if (open[kind]) {
                    pr.ms[kind] += (double)(h[2 * i] - open_t[kind]) * 1e-5;   // 100 MHz ticks -> ms
                    pr.n[kind] += 1;
                    open[kind] = false;
                }
            }
            (void)hipMemset(pr.ring_count, 0, 4);
        }
    }
    unsigned long long cnt[4] = {0, 0, 0, 0};
    if (pr.counters) (void)hipMemcpy(cnt, pr.counters, 32, hipMemcpyDeviceToHost);
    const double v[14] = {pr.ms[0], (double)pr.n[0], pr.ms[1], (double)pr.n[1], (double)cnt[0], (double)cnt[1],
                          pr.ms[2], (double)pr.n[2], pr.ms[3], (double)pr.n[3], pr.ms[4], (double)pr.n[4], (double)cnt[2], (double)cnt[3]};
    int k = cap < 14 ? cap : 14;
    for (int i = 0; i < k; i++) out[i] = v[i];
    return k;
}

// ---- the same knobs on the default context of the calling thread's current device --------------------------------------
void dgs_set_tight_rects(int on) { (void)dgs_context_set_option(default_context(), 0, on); }
int dgs_set_option(int key, int value) { return dgs_context_set_option(default_context(), key, value); }
int dgs_get_option(int key) { return dgs_context_get_option(default_context(), key); }
int dgs_set_overflow_flag(int* device_flag) { return dgs_context_set_overflow_flag(default_context(), device_flag); }
int dgs_read_overflow(int reset) { return dgs_context_read_overflow(default_context(), reset); }
void dgs_profile_enable(int mode) { (void)dgs_context_profile_enable(default_context(), mode); }
void dgs_profile_reset(void) { dgs_context_profile_reset(default_context()); }
int dgs_profile_read(double* out, int cap) { return dgs_context_profile_read(default_context(), out, cap); }

#ifdef DGS_COUNT_VISITS
// development build only (-DDGS_COUNT_VISITS, tools/diag/visit_counts.py): the backward blend's visit counters, optionally cleared
int dgs_debug_visit_counts(unsigned long long* out4, int reset)
{
    if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(dgs::g_visit_counts), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dgs::g_visit_counts), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int dgs_debug_layout(int which, int P, int width, int height, int R, size_t* offsets, int cap)
{
    std::vector<size_t> v;
    if (which == 0) {
        GeomLayout g(P);
        v = {g.rec, g.total, g.internal_radii, g.acc, g.rects, g.bytes};
    } else if (which == 1) {
        ImageLayout m(width, height);
        v = {m.final_T, m.n_contrib, m.ranges, m.tile_last, m.order_fwd, m.order_bwd, m.tile_counts, m.cursor, m.bytes};
    } else if (which == 2) {
        ImageLayout m(width, height);
        BinningLayout b(R, false);
        v = {b.point_list, b.keys, b.scratch, b.bytes};
    } else {
        return fail(DGS_ERR_INVALID_ARGUMENT, "dgs_debug_layout: which must be 0, 1 or 2");
    }
    int n = (int)v.size() < cap ? (int)v.size() : cap;
    for (int i = 0; i < n; i++) offsets[i] = v[i];
    return n;
}

int dgs_rasterizer_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                unsigned char* present, void* stream_)
{
    (void)projmatrix;  // only feeds a dead expression in the reference (auxiliary.h:170-172)
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0) return fail(DGS_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return DGS_OK;
    if (!means3D || !viewmatrix || !present) return fail(DGS_ERR_INVALID_ARGUMENT, "NULL pointer");
    hipLaunchKernelGGL(dgs::mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
    DGS_STAGE("mark_visible", 0, stream);
    return DGS_OK;
}

int dgs_context_forward(dgs_context* ctx, dgs_alloc_fn geometry_alloc, void* geometry_ctx, dgs_alloc_fn binning_alloc, void* binning_ctx,
                           dgs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                           int height, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                           float* out_others, int* radii, int debug, void* stream_)
{
    (void)scale_modifier; (void)projmatrix; (void)prefiltered;
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    if (int e = check_common(P, width, height, means3D)) return e;
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(DGS_ERR_INVALID_ARGUMENT, "allocator callback is NULL");
    if (!out_color || !out_others || !background || !viewmatrix || !cam_pos) return fail(DGS_ERR_INVALID_ARGUMENT, "NULL pointer");
    if (transMat_precomp) return fail(DGS_ERR_UNSUPPORTED, "transMat_precomp (cov3D_precomp) is not supported; pass scales and rotations");
    if (P == 0) return 0;  // rasterize_points.cu:106
    if (!opacities || !scales || !rotations) return fail(DGS_ERR_INVALID_ARGUMENT, "opacities/scales/rotations NULL");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(DGS_ERR_INVALID_ARGUMENT, "provide exactly one of shs / colors_precomp");
    if (shs && (M <= 0 || D < 0 || D > 3 || (D + 1) * (D + 1) > M))
        return fail(DGS_ERR_INVALID_ARGUMENT, "SH degree / coefficient count mismatch");
    if (D < 0) D = 0;

    GeomLayout gl(P);
    ImageLayout il(width, height);
    char* geom = geometry_alloc(geometry_ctx, gl.bytes);
    char* img = image_alloc(image_ctx, il.bytes);
    if (!geom || !img) return fail(DGS_ERR_ALLOC, "geometry/image allocator returned NULL");
    if (!radii) radii = (int*)(geom + gl.internal_radii);  // rasterizer_impl.cu:230-233

    const dgs::Camera cam = make_camera(viewmatrix, cam_pos, width, height, tan_fovx, tan_fovy);

    // ---- K2 preprocess
    uint32_t* tile_counts = (uint32_t*)(img + il.tile_counts);
    uint32_t* cursor = (uint32_t*)(img + il.cursor);
    uint2* ranges = (uint2*)(img + il.ranges);
    dgs::PreprocessArgs pa;
    pa.P = P; pa.D = D; pa.M = M;
    pa.means3D = means3D; pa.scales = scales; pa.rotations = rotations; pa.opacities = opacities;
    pa.shs = shs; pa.colors_precomp = colors_precomp; pa.cam = cam;
    pa.radii = radii;
    pa.rec = (float4*)(geom + gl.rec);
    pa.rects = (uint2*)(geom + gl.rects);
    pa.tight = ctx->tight_rects.load() ? 1 : 0;
    pa.row_inv = 0;
    size_t sh_lds = 0;
    if (!colors_precomp && shs) {
        if (M * 3 > 48 || M <= 0) return fail(DGS_ERR_INVALID_ARGUMENT, "forward: 1..16 SH coefficients per channel expected");
        pa.row_inv = (unsigned)(0xFFFFFFFFu / (unsigned)(M * 3)) + 1u;
        sh_lds = (size_t)dgs::kSurfelBlock * (M * 3 + 1) * sizeof(float);
    }
    Prof::Pair pp2;
    const bool timed2 = prof_begin(ctx, 2, stream, pp2);
    hipLaunchKernelGGL(dgs::preprocess_fwd_kernel, dim3(gl.nblocks), dim3(dgs::kSurfelBlock), sh_lds, stream, pa);
    if (timed2) prof_mark_end(ctx, stream, pp2);
    DGS_STAGE("preprocess_fwd", debug, stream);
    Prof::Pair pp3;
    // binning: count, column pass, scan, column pass, scatter, sort.  (Exact-size mode: the span also contains the D2H copy of
    // num_rendered, the stream synchronisation and the caller's allocator below -- its time is kernel time only in capacity mode,
    // which is where bench.py reads it.)
    const bool timed3 = prof_begin(ctx, 3, stream, pp3);
    // ---- K3 per-tile entry counts
    dgs::BinArgs ba_;
    ba_.P = P; ba_.ntiles = il.ntiles; ba_.tiles_x = il.tiles_x; ba_.chunk = (P + dgs::kBinGroups - 1) / dgs::kBinGroups;
    ba_.radii = radii; ba_.rects = pa.rects; ba_.rec = pa.rec; ba_.M = cursor; ba_.keys = nullptr;
    ba_.state = (const uint32_t*)(geom + gl.total);
    const size_t hist_bytes = (size_t)il.ntiles * 4;
    const int off_blocks = (il.ntiles + 63) / 64;
    const bool merged_offsets = il.lds_bins && ctx->merged_offsets.load();
    ba_.sync = merged_offsets ? tile_counts : nullptr;
    ba_.nsync = 2 * off_blocks + 1;
    ba_.ranges = nullptr; ba_.order = nullptr; ba_.group_xcd = nullptr; ba_.tile_last = nullptr; ba_.tiles_y = il.tiles_y; ba_.order_mode = 0;
    if (il.lds_bins) {
        hipLaunchKernelGGL(dgs::count_tiles_lds_kernel, dim3(dgs::kBinGroups), dim3(dgs::kBinThreads), hist_bytes, stream, ba_);
        if (!merged_offsets)
            hipLaunchKernelGGL(dgs::column_pass_kernel, dim3(off_blocks), dim3(64 * dgs::kColGroups), 0, stream, cursor, il.ntiles,
                               (const uint2*)nullptr, tile_counts);
    } else {
        DGS_HIP(hipMemsetAsync(tile_counts, 0, hist_bytes, stream));
        hipLaunchKernelGGL(dgs::count_tiles_global_kernel, dim3(gl.nblocks), dim3(dgs::kSurfelBlock), 0, stream, P, (const int*)radii,
                           (const uint2*)pa.rects, il.tiles_x, tile_counts);
    }
    DGS_STAGE("count_tiles", debug, stream);
    const int capacity = ctx->capacity.load();
    int* overflow = ctx->overflow.load();
    // ---- K3/K6 scan of the T tile counts -> tile ranges, num_rendered, longest list
    int tile_order = ctx->tile_order.load();
    if (tile_order == 4 && dgs::order_groups(il.tiles_x, il.tiles_y) > dgs::kOrderMaxGroups) tile_order = 3;   // > 128 x 128 tiles
    bool order_later = false;
    if (merged_offsets) {
        // column sums + scan + bucket cursors + dispatch order in one launch (kernels_preprocess.h: bin_offsets_kernel)
        dgs::OffsetsArgs oa;
        oa.M = cursor; oa.ntiles = il.ntiles; oa.ranges = ranges; oa.state = (uint32_t*)(geom + gl.total);
        oa.cap = (uint32_t)capacity; oa.overflow = overflow; oa.list_hint = (uint32_t)(capacity > 0 ? ctx->list_hint.load() : 0);
        oa.order = tile_order >= 3 ? (uint32_t*)(img + il.order_fwd) : (uint32_t*)nullptr;
        oa.tiles_x = il.tiles_x; oa.tiles_y = il.tiles_y; oa.order_mode = tile_order;
        oa.group_xcd = (uint32_t*)(img + il.group_xcd); oa.tile_last = (uint32_t*)(img + il.tile_last);
        oa.long_thr = (uint32_t*)(img + il.long_thr); oa.long_div = (uint32_t)ctx->long_div_fwd.load();
        oa.sync = tile_counts;
        // capacity mode: the scatter launch always follows (R = capacity > 0) and carries the order as a rider workgroup
        oa.order_later = (capacity > 0 && oa.order != nullptr && ctx->order_rider.load()) ? 1 : 0;
        order_later = oa.order_later != 0;
        hipLaunchKernelGGL(dgs::bin_offsets_kernel, dim3(off_blocks), dim3(64 * dgs::kColGroups), 0, stream, oa);
    } else
    hipLaunchKernelGGL(dgs::scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)tile_counts, il.ntiles, ranges,
                       il.lds_bins ? (uint32_t*)nullptr : cursor, (uint32_t*)(geom + gl.total), (uint32_t)capacity, overflow,
                       tile_order >= 3 ? (uint32_t*)(img + il.order_fwd) : (uint32_t*)nullptr,   // + the forward's dispatch order
                       (uint32_t)(capacity > 0 ? ctx->list_hint.load() : 0), il.tiles_x, il.tiles_y, tile_order,
                       (uint32_t*)(img + il.group_xcd), (uint32_t*)(img + il.tile_last), (uint32_t*)(img + il.long_thr), (uint32_t)ctx->long_div_fwd.load());
    DGS_STAGE("scan_tiles", debug, stream);

    // ---- num_rendered to the host: the binning buffer is sized from it (rasterizer_impl.cu:281-285).
    // Capacity mode skips the round trip: the buffer is sized for g_capacity entries, every sort variant is
    // launched (each tile picks its own), and the value returned to the caller is the capacity.
    uint32_t R_u = 0, longest = 0;
    const bool capacity_mode = capacity > 0;
    if (capacity_mode) {
        R_u = (uint32_t)capacity;
        const int hint = ctx->list_hint.load();
        longest = hint > 0 ? (uint32_t)hint : 0xffffffffu;   // a longer list than promised raises the overflow flag (scan_tiles_kernel)
    } else {
        // one pinned word per context: concurrent forwards of the same context take turns here (other contexts do not wait)
        std::lock_guard<std::mutex> lk(ctx->stage.mu);
        if (ctx->stage.ensure()) return fail(DGS_ERR_HIP, "hipHostMalloc failed");
        DGS_HIP(hipMemcpyAsync(ctx->stage.u, geom + gl.total, 8, hipMemcpyDeviceToHost, stream));
        DGS_HIP(hipStreamSynchronize(stream));
        R_u = ctx->stage.u[0];
        longest = ctx->stage.u[1];
    }
    if (R_u > 0x7fffffffu) return fail(DGS_ERR_INVALID_ARGUMENT, "num_rendered overflows int32");
    const int R = (int)R_u;

    const bool need_global_sort = longest > (ctx->sort_regs.load() == 2 ? (uint32_t)dgs::kSegCap : 16384u);   // (radix mode: scratch for the sorted segments)
    BinningLayout bl(R, need_global_sort);
    char* bin = binning_alloc(binning_ctx, bl.bytes);
    if (!bin) return fail(DGS_ERR_ALLOC, "binning allocator returned NULL");
    if (R > 0) {
        // ---- K4 scatter (depth, index) keys into the tile buckets
        uint64_t* keys = (uint64_t*)(bin + bl.keys);
        if (il.lds_bins) {
            if (!merged_offsets)
                hipLaunchKernelGGL(dgs::column_pass_kernel, dim3(off_blocks), dim3(64 * dgs::kColGroups), 0, stream, cursor, il.ntiles,
                                   (const uint2*)ranges, (uint32_t*)nullptr);
            ba_.keys = keys;
            if (order_later) {
                ba_.ranges = ranges; ba_.order = (uint32_t*)(img + il.order_fwd); ba_.group_xcd = (uint32_t*)(img + il.group_xcd);
                ba_.tile_last = (uint32_t*)(img + il.tile_last); ba_.order_mode = tile_order;
            }
            const size_t scatter_lds = order_later && hist_bytes < dgs::kScatterRiderLds ? dgs::kScatterRiderLds : hist_bytes;
            hipLaunchKernelGGL(dgs::scatter_keys_lds_kernel, dim3(dgs::kBinGroups + (order_later ? 1 : 0)), dim3(dgs::kBinThreads), scatter_lds, stream, ba_);
        } else {
            dgs::ScatterArgs sa;
            sa.P = P; sa.radii = radii; sa.rec = pa.rec; sa.rects = pa.rects; sa.cursor = cursor;
            sa.state = (const uint32_t*)(geom + gl.total);
            sa.keys = keys;
            sa.tiles_x = il.tiles_x;
            hipLaunchKernelGGL(dgs::scatter_keys_kernel, dim3(gl.nblocks), dim3(dgs::kSurfelBlock), 0, stream, sa);
        }
        DGS_STAGE("scatter_keys", debug, stream);
        // ---- K5 per-tile sort (stable radix order of rasterizer_impl.cu:304-309 = (tile, depth bits, surfel index))
        uint32_t* plist = (uint32_t*)(bin + bl.point_list);
        const int sort_mode = ctx->sort_regs.load();
        if (sort_mode == 2) {
            // per-tile LSD radix sort (default): lists up to 2048 entries in 36 KB of LDS (one workgroup per tile); longer ones as
            // segments of 2048 sorted side by side by the same kernel + a rank / merge step, up to 28 segments; beyond 57 344 entries
            // the global-memory network.  Every tile picks its kernel on the device; when the longest list is known on the host
            // (exact-size mode, or promised) the launches that cannot have work are skipped.
            const int big_grid = il.ntiles < 256 ? il.ntiles : 256;
            hipLaunchKernelGGL((dgs::sort_tiles_radix_kernel<2048>), dim3(il.ntiles), dim3(256), 0, stream, (const uint2*)ranges, il.ntiles,
                               (const uint64_t*)keys, plist, 0);
            if (longest > (uint32_t)dgs::kSegCap) {
                // (a 30 000-entry list on ONE workgroup's global-memory network took 0.45 ms; a crowded tile is exactly what
                // densification produces)
                // 256 columns of workgroups walk the tiles (64 were enough while segments were rare: at 1 M surfels / 1600 x 1600 thousands of
                // tiles hold 2-4 k entries and a column handled 156 of them one after the other); rows that no list reaches exit at once
                const dim3 seg_grid(il.ntiles < 256 ? il.ntiles : 256, dgs::kMaxSegs);
                const uint32_t* state = (const uint32_t*)(geom + gl.total);
                hipLaunchKernelGGL((dgs::sort_tiles_radix_kernel<dgs::kSegCap, true>), seg_grid, dim3(256), 0, stream, (const uint2*)ranges, il.ntiles,
                                   (const uint64_t*)keys, plist, 0, (uint64_t*)(bin + bl.scratch), state);
                hipLaunchKernelGGL(dgs::merge_segments_kernel, seg_grid, dim3(256), 0, stream, (const uint2*)ranges, il.ntiles,
                                   (const uint64_t*)(bin + bl.scratch), plist, state);
            }
            if (longest > (uint32_t)(dgs::kSegCap * dgs::kMaxSegs))
                hipLaunchKernelGGL(dgs::sort_tiles_global_kernel, dim3(big_grid), dim3(256), 0, stream, (const uint2*)ranges, il.ntiles,
                                   (const uint64_t*)keys, (uint64_t*)(bin + bl.scratch), plist, dgs::kSegCap * dgs::kMaxSegs);
        } else {
            if (sort_mode == 1)
                hipLaunchKernelGGL(dgs::sort_tiles_reg_kernel, dim3(il.ntiles), dim3(256), 0, stream, (const uint2*)ranges,
                                   (const uint64_t*)keys, plist);
            else
                hipLaunchKernelGGL((dgs::sort_tiles_lds_kernel<2048>), dim3(il.ntiles), dim3(256), 0, stream, (const uint2*)ranges,
                                   (const uint64_t*)keys, plist, 0);
            // capacity mode does not know the longest list on the host: ONE fallback launch (global scratch) covers every tile
            // above 2048 entries instead of two mostly empty ones
            if (longest > 2048u && !capacity_mode)
                hipLaunchKernelGGL((dgs::sort_tiles_lds_kernel<16384>), dim3(il.ntiles), dim3(256), 0, stream, (const uint2*)ranges,
                                   (const uint64_t*)keys, plist, 2048);
            if (need_global_sort)
                hipLaunchKernelGGL(dgs::sort_tiles_global_kernel, dim3(il.ntiles < 256 ? il.ntiles : 256), dim3(256), 0, stream, (const uint2*)ranges, il.ntiles,
                                   (const uint64_t*)keys, (uint64_t*)(bin + bl.scratch), plist, capacity_mode ? 2048 : 16384);
        }
        DGS_STAGE("sort_tiles", debug, stream);
    }
    if (timed3) {
        prof_mark_end(ctx, stream, pp3);
        if (ctx->prof.counters)
            hipLaunchKernelGGL(sum_forward_units_kernel, dim3(64), dim3(256), 0, stream, (const uint32_t*)(geom + gl.total), (const int*)radii, P,
                               ctx->prof.counters + 2, ctx->prof.counters + 3);
    }

    // ---- K7 forward blend
    dgs::BlendFwdArgs fa;
    fa.ranges = ranges;
    fa.point_list = (const uint32_t*)(bin + bl.point_list);
    fa.rec = pa.rec;
    fa.W = width; fa.H = height; fa.tiles_x = il.tiles_x; fa.tiles_y = il.tiles_y; fa.mode = tile_order;
    fa.order = (const uint32_t*)(img + il.order_fwd);   // written by scan_tiles_kernel
    fa.bg = background;
    fa.final_T = (float*)(img + il.final_T);
    fa.n_contrib = (uint32_t*)(img + il.n_contrib);
    fa.tile_last = (uint32_t*)(img + il.tile_last);
    fa.out_color = out_color;
    fa.out_others = out_others;
    int grid = dgs::blend_grid_size(il.tiles_x, il.tiles_y, fa.mode);
    fa.long_thr = nullptr;
#if DGS_FWD_ROWS
    if (fa.mode == 3 && ctx->long_tiles.load() && !ctx->grid_limit_fwd.load()) {   // long-tile path (kernels_blend.h): four workgroups for the longest lists
        fa.long_thr = (const dgs::LongThr*)(img + il.long_thr);
        grid = dgs::long_grid_size(il.ntiles);
    }
#endif
    if (const int lim = ctx->grid_limit_fwd.load()) {
        grid = lim < grid ? lim : grid;
        DGS_HIP(hipMemsetAsync(img + il.final_T, 0, il.ranges - il.final_T, stream));   // final_T, n_contrib of the skipped tiles
        DGS_HIP(hipMemsetAsync(img + il.tile_last, 0, (size_t)il.ntiles * 4, stream));
    }
    // accumulator rider: 256 workgroups in FRONT of the tile workgroups (they hold a slot for a few microseconds each while the tile
    // workgroups fill the rest of the chip; at the end of the grid they would lengthen the drain)
    fa.clear = nullptr; fa.clear_n4 = 0; fa.clear_blocks = 0;
    fa.clean_flag = (uint32_t*)(geom + gl.total) + 3;
    {
        const size_t acc_bytes = (size_t)P * dgs::kAccFloats * 4;
        if (ctx->acc_rider.load() && !ctx->grid_limit_fwd.load() && tile_order >= 3 && acc_bytes % 16 == 0 && acc_bytes / 16 < 0xf0000000ull) {
            fa.clear = (float4*)(geom + gl.acc);
            fa.clear_n4 = (uint32_t)(acc_bytes / 16);
            fa.clear_blocks = (int)std::min<size_t>(256, ((fa.clear_n4 + 16 * dgs::kTilePix - 1) / (16 * dgs::kTilePix) + 7) / 8 * 8);   // ~16 stores per thread
            grid += fa.clear_blocks;
        }
    }
    Prof::Pair pp;
    const bool timed = prof_begin(ctx, 0, stream, pp);
#if DGS_FWD_ROWS
    hipLaunchKernelGGL(dgs::blend_fwd_rows_kernel, dim3(grid), dim3(dgs::kTilePix), 0, stream, fa);
#else
    hipLaunchKernelGGL(dgs::blend_fwd_kernel, dim3(grid), dim3(dgs::kTilePix), 0, stream, fa);
#endif
    if (timed) prof_end(ctx, stream, pp, fa.tile_last, il.ntiles);
    DGS_STAGE("blend_fwd", debug, stream);
    return R;
}

int dgs_context_backward(dgs_context* ctx, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                            const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                            char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                            const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                            float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                            float* dL_drot, int debug, void* stream_)
{
    (void)scale_modifier; (void)projmatrix;
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx) return fail(DGS_ERR_INVALID_ARGUMENT, "context is NULL");
    if (int e = check_common(P, width, height, means3D)) return e;
    if (P == 0) return DGS_OK;  // rasterize_points.cu:204
    if (transMat_precomp) return fail(DGS_ERR_UNSUPPORTED, "transMat_precomp (cov3D_precomp) is not supported");
    if ((unsigned long long)P * dgs::kAccFloats * 4ull >= 0xffffffffull) return fail(DGS_ERR_UNSUPPORTED, "backward: more than 53 million surfels (32-bit accumulator offsets)");
    if (!geom_buffer || !img_buffer || (R > 0 && !binning_buffer)) return fail(DGS_ERR_INVALID_ARGUMENT, "scratch buffer is NULL");
    if (!dL_dpix || !dL_depths || !dL_dmean2D || !dL_dopacity || !dL_dmean3D ||   // (dL_dnormal, dL_dcolor, dL_dtransMat: NULL = not wanted)
        !dL_dscale || !dL_drot || !scales || !rotations || !viewmatrix || !campos || !background)
        return fail(DGS_ERR_INVALID_ARGUMENT, "NULL pointer");
    if (shs && colors_precomp == nullptr && !dL_dsh) return fail(DGS_ERR_INVALID_ARGUMENT, "dL_dsh is NULL");
    if (D < 0) D = 0;

    GeomLayout gl(P);
    ImageLayout il(width, height);
    BinningLayout bl(R, false);  // only point_list is needed; its offset does not depend on the scratch
    if (!radii) radii = (const int*)(geom_buffer + gl.internal_radii);

    const dgs::Camera cam = make_camera(viewmatrix, campos, width, height, tan_fovx, tan_fovy);

    float* acc = (float*)(geom_buffer + gl.acc);
    const size_t acc_bytes = (size_t)P * dgs::kAccFloats * 4;
    int bwd_mode = ctx->tile_order.load();
    if (bwd_mode == 4 && dgs::order_groups(il.tiles_x, il.tiles_y) > dgs::kOrderMaxGroups) bwd_mode = 3;
    // accumulator rows cleared and (modes 3, 4) the tiles ordered by the traversed length the forward measured: one launch
    const bool prep_fused = R > 0 && bwd_mode >= 3 && acc_bytes % 16 == 0 && (reinterpret_cast<size_t>(acc) & 15) == 0;
    if (prep_fused) {
        const size_t n4 = acc_bytes / 16;
        const int fill_blocks = (int)std::min<size_t>(1024, (n4 + 4 * 1024 - 1) / (4 * 1024));   // ~4 stores of 16 B per thread
        hipLaunchKernelGGL(dgs::prep_bwd_kernel, dim3(fill_blocks + 1), dim3(1024), 0, stream, (float4*)acc, n4,
                           (const uint32_t*)(img_buffer + il.tile_last), il.tiles_x, il.tiles_y, bwd_mode,
                           (uint32_t*)(img_buffer + il.order_bwd), (uint32_t*)(img_buffer + il.group_xcd), (uint32_t*)(img_buffer + il.long_thr), (uint32_t)ctx->long_div_bwd.load(),
                           DGS_BWD_ROWS ? (const uint32_t*)nullptr : (const uint32_t*)(geom_buffer + gl.total) + 3);   // (the A/B rows kernel does not reset the flag)
        DGS_STAGE("prep_bwd", debug, stream);
    } else {
        DGS_HIP(hipMemsetAsync(acc, 0, acc_bytes, stream));
    }

    // ---- K8 backward blend
    if (R > 0) {
        dgs::BlendBwdArgs ba;
        ba.ranges = (const uint2*)(img_buffer + il.ranges);
        ba.point_list = (const uint32_t*)(binning_buffer + bl.point_list);
        ba.rec = (const float4*)(geom_buffer + gl.rec);
        ba.W = width; ba.H = height; ba.tiles_x = il.tiles_x; ba.tiles_y = il.tiles_y; ba.mode = ctx->tile_order.load();
        ba.order = (const uint32_t*)(img_buffer + il.order_bwd);
        if (ba.mode == 4 && dgs::order_groups(il.tiles_x, il.tiles_y) > dgs::kOrderMaxGroups) ba.mode = 3;
        if (ba.mode >= 3 && !prep_fused) {  // by the traversed length the forward measured
            hipLaunchKernelGGL(dgs::tile_order_kernel, dim3(1), dim3(1024), 0, stream, (const uint2*)nullptr,
                               (const uint32_t*)(img_buffer + il.tile_last), il.tiles_x, il.tiles_y, ba.mode,
                               (uint32_t*)(img_buffer + il.order_bwd), (uint32_t*)(img_buffer + il.group_xcd));
            DGS_STAGE("tile_order_bwd", debug, stream);
        }
        ba.bg = background;
        ba.final_T = (const float*)(img_buffer + il.final_T);
        ba.n_contrib = (const uint32_t*)(img_buffer + il.n_contrib);
        ba.tile_last = (const uint32_t*)(img_buffer + il.tile_last);
        ba.dL_dpix = dL_dpix;
        ba.dL_dothers = dL_depths;
        ba.acc = acc;
        ba.clean_flag = (uint32_t*)(geom_buffer + gl.total) + 3;
        int grid = dgs::blend_grid_size(il.tiles_x, il.tiles_y, ba.mode);
        ba.long_thr = nullptr;
#if !DGS_BWD_ROWS && DGS_BWD_REDUCE == 4
        if (ba.mode == 3 && prep_fused && ctx->long_tiles.load() && !ctx->grid_limit_bwd.load()) {   // long-tile path (kernels_blend.h)
            ba.long_thr = (const dgs::LongThr*)(img_buffer + il.long_thr);
            grid = dgs::long_grid_size(il.ntiles);
        }
#endif
        if (const int lim = ctx->grid_limit_bwd.load()) grid = lim < grid ? lim : grid;
        Prof::Pair pp;
        const bool timed = prof_begin(ctx, 1, stream, pp);
        const int det = ctx->deterministic.load();
        if (det == 2) {
            // fixed-point sums with integer atomics: order-free, capturable.  The rows live in the context (zero between backward
            // passes: fixed_to_acc_kernel clears what it converts); they are (re)allocated outside a capture only
            {
                std::lock_guard<std::mutex> lk(ctx->mu);
                if (ctx->acc64_rows < (size_t)P) {
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    (void)hipStreamIsCapturing(stream, &cs);
                    if (cs != hipStreamCaptureStatusNone)
                        return fail(DGS_ERR_INVALID_ARGUMENT, "deterministic backward (option 7 = 2): the fixed-point rows must exist before a capture (run one eager backward at this size first)");
                    DGS_HIP(hipStreamSynchronize(stream));
                    //  the outgrown rows are RETIRED, not freed: a graph captured at the smaller size has their address baked
                    // in and may still be replayed (its sums then land in rows nobody converts -- a stale graph, but not a write into
                    // freed memory).  Two backward passes on different streams of ONE context would still share the live rows: callers
                    // that run views concurrently give each its own context (diff_surfel_rasterization.Lane).
                    if (ctx->acc64) ctx->acc64_retired.push_back(ctx->acc64);
                    ctx->acc64 = nullptr; ctx->acc64_rows = 0;
                    const size_t rows = (size_t)P + (size_t)P / 4 + 1024;
                    DGS_HIP(hipMalloc((void**)&ctx->acc64, rows * dgs::kAccFloats * sizeof(unsigned long long)));
                    DGS_HIP(hipMemset(ctx->acc64, 0, rows * dgs::kAccFloats * sizeof(unsigned long long)));
                    ctx->acc64_rows = rows;
                }
            }
            ba.det_part = nullptr;
            ba.acc64 = ctx->acc64;
            hipLaunchKernelGGL(dgs::blend_bwd_kernel<2>, dim3(grid), dim3(dgs::kTilePix), 0, stream, ba);
            const size_t n = (size_t)P * dgs::kAccFloats;
            hipLaunchKernelGGL(dgs::fixed_to_acc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ctx->acc64, acc, n);
        } else if (det == 1) {
            // test option: every (list entry, wave) stores its sums in a row of its own, a per-surfel kernel adds them in a fixed
            // order.  R x 320 bytes of scratch from the stream-ordered allocator (not capturable: tests run eagerly)
            float* part = nullptr;
            const size_t bytes = (size_t)R * dgs::kDetRows * dgs::kAccFloats * sizeof(float);
            DGS_HIP(hipMallocAsync((void**)&part, bytes, stream));
            DGS_HIP(hipMemsetAsync(part, 0, bytes, stream));
            ba.det_part = part;
#if DGS_BWD_ROWS
            hipLaunchKernelGGL(dgs::blend_bwd_rows_kernel<true>, dim3(grid), dim3(dgs::kTilePix), 0, stream, ba);
#else
            hipLaunchKernelGGL(dgs::blend_bwd_kernel<1>, dim3(grid), dim3(dgs::kTilePix), 0, stream, ba);
#endif
            hipLaunchKernelGGL(dgs::det_reduce_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, radii,
                               (const uint2*)(geom_buffer + gl.rects), il.tiles_x, ba.ranges, ba.point_list, (const float*)part, acc);
            DGS_HIP(hipFreeAsync(part, stream));
        } else {
            ba.det_part = nullptr;
            ba.acc64 = nullptr;
#if DGS_BWD_ROWS
            hipLaunchKernelGGL(dgs::blend_bwd_rows_kernel<false>, dim3(grid), dim3(dgs::kTilePix), 0, stream, ba);
#else
            hipLaunchKernelGGL(dgs::blend_bwd_kernel<0>, dim3(grid), dim3(dgs::kTilePix), 0, stream, ba);
#endif
        }
        if (timed) prof_end(ctx, stream, pp, ba.tile_last, il.ntiles);
        DGS_STAGE("blend_bwd", debug, stream);
    }

    // ---- K9 + K10 fused per-surfel backward
    dgs::SurfelBwdArgs sa;
    sa.P = P; sa.D = D; sa.M = M;
    sa.means3D = means3D; sa.scales = scales; sa.rotations = rotations;
    sa.shs = colors_precomp ? nullptr : shs;
    sa.cam = cam;
    sa.radii = radii;
    sa.rec = (const float4*)(geom_buffer + gl.rec);
    sa.acc = acc;
    sa.dL_dmean2D = dL_dmean2D; sa.dL_dnormal = dL_dnormal; sa.dL_dopacity = dL_dopacity; sa.dL_dcolor = dL_dcolor;
    sa.dL_dmean3D = dL_dmean3D; sa.dL_dtransMat = dL_dtransMat; sa.dL_dsh = dL_dsh; sa.sh_all_rows = ctx->sh_all_rows.load(); sa.dL_dscale = dL_dscale; sa.dL_drot = dL_drot;
    size_t sh_lds = 0;
    if (sa.shs) {
        if (M * 3 > 48) return fail(DGS_ERR_INVALID_ARGUMENT, "backward: more than 16 SH coefficients per channel");
        sa.row_inv = (unsigned)(0xFFFFFFFFu / (unsigned)(M * 3)) + 1u;
        sh_lds = (size_t)dgs::kSurfelBlock * (M * 3 + 1) * sizeof(float);
    }
    Prof::Pair pp4;
    const bool timed4 = prof_begin(ctx, 4, stream, pp4);
    hipLaunchKernelGGL(dgs::surfel_bwd_kernel, dim3(gl.nblocks), dim3(dgs::kSurfelBlock), sh_lds, stream, sa);
    if (timed4) prof_mark_end(ctx, stream, pp4);
    DGS_STAGE("surfel_bwd", debug, stream);
    return DGS_OK;
}


// ---- the reference-shaped entry points (rasterizer.h:20-87): default context of the current device ------------------
int dgs_rasterizer_forward(dgs_alloc_fn geometry_alloc, void* geometry_ctx, dgs_alloc_fn binning_alloc, void* binning_ctx,
                           dgs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                           int height, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                           float* out_others, int* radii, int debug, void* stream)
{
    return dgs_context_forward(default_context(), geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
                               background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                               transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
                               out_others, radii, debug, stream);
}

int dgs_rasterizer_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                            const float* rotations, const float* transMat_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                            char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
                            const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                            float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale,
                            float* dL_drot, int debug, void* stream)
{
    return dgs_context_backward(default_context(), P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales,
                                scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                                geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_depths, dL_dmean2D, dL_dnormal, dL_dopacity,
                                dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, debug, stream);
}

}  // extern "C"
