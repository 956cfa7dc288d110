// Forward / backward chain kernels of csrc/node_mlp.h alone, random weights: time per launch (HIP events) and, built with
// -DDGS_MLP_TRACE, the 100 MHz device clock of workgroup 0 / thread 0 after every stage of the forward chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDGS_MLP_TRACE tools/micro/mlp_chain_bench.hip -o tools/micro/mlp_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#ifdef DGS_MLP_TRACE
__device__ long long g_trace[64];
__device__ int g_trace_n;
#define MLP_TRACE_POINT() do { if (threadIdx.x == 0 && blockIdx.x == 0 && g_trace_n < 64) g_trace[g_trace_n++] = wall_clock64(); } while (0)
#endif
#include "../../dynamic-2dgs_amd/csrc/node_mlp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 1024, iters = 50;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 0.05f);
    auto dev = [&](size_t n, bool rnd) {
        std::vector<float> h(n, 0.f);
        if (rnd) for (auto& v : h) v = nd(rng);
        float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        return d;
    };
    mlp::Weights w{};
    for (int l = 0; l < 10; l++) { w.W[l] = dev((size_t)mlp::l_out(l) * mlp::l_in(l), true); w.b[l] = dev(mlp::l_out(l), true); }
    float* hw = dev(16 * 256, true); float* hb = dev(16, true);
    for (int r = 0; r < 16; r++) { w.hw[r] = hw + 256 * (r < 13 ? r : 0); w.hb[r] = hb + (r < 13 ? r : 0); }
    float* x = dev((size_t)M * 3, true); float* t = dev(M, true);
    float* packed = dev(mlp::kPackedFloats, false);
    float* saved = dev(mlp::sv_total(M), false);
    float* scratch = dev(mlp::sc_total(M), false);
    float* attrs = dev((size_t)M * 13, false);
    float* g_attrs = dev((size_t)M * 13, true);
    int nthreads = mlp::kFwdVecs + mlp::kBwdVecs;
    hipLaunchKernelGGL(mlp::mlp_pack_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, 0, w, (float4*)packed);
    mlp::FwdArgs a{}; a.M = M; a.x = x; a.x_stride = 3; a.t = t; a.t_stride = 1; a.wp = (const float4*)packed;
    a.bias = packed + mlp::kBiasOff; a.saved = saved; a.attrs = attrs;
    mlp::BwdArgs b{}; b.M = M; b.g_attrs = g_attrs; b.saved = saved; b.scratch = scratch; b.wq = (const float4*)packed + mlp::kFwdVecs;
    // weight-gradient descriptors as dgs_mlp_backward builds them (csrc/train_ops.hip)
    mlp::WgArgs g{};
    g.M = M; g.accumulate = 1;
    float* gw[10]; float* gb[10];
    for (int l = 0; l < 10; l++) { gw[l] = dev((size_t)mlp::l_out(l) * mlp::l_in(l), false); gb[l] = dev(mlp::l_out(l), false); }
    float* ghw = dev(16 * 256, false); float* ghb = dev(16, false);
    for (int r = 0; r < 16; r++) { g.hw[r] = ghw + 256 * (r < 13 ? r : 0); g.hb[r] = ghb + (r < 13 ? r : 0); }
    int ndesc = 0, block = 0;
    auto add = [&](const float* dz, int dzs, int out, const float* xx, int xs, int in, float* dw, int dws, float* db) {
        mlp::WgDesc& d = g.d[ndesc++];
        d.dz = dz; d.dz_stride = dzs; d.out = out; d.x = xx; d.x_stride = xs; d.in = in; d.dw = dw; d.dw_stride = dws; d.db = db;
        d.iblocks = (in + mlp::kWgTileI - 1) / mlp::kWgTileI;
        d.ntiles = ((out + mlp::kWgTileJ - 1) / mlp::kWgTileJ) * d.iblocks;
    };
    const int W = mlp::kW;
    auto H = [&](int l) { return saved + mlp::sv_h(M, l); };
    auto dZ = [&](int l) { return scratch + mlp::sc_dz(M, l); };
    add(g_attrs, mlp::kHeads, mlp::kHeads, H(7), W, W, nullptr, W, nullptr);
    for (int l = 7; l >= 1; l--) {
        if (l == 5) {
            add(dZ(5), W, W, saved + mlp::sv_inp(M), mlp::kInPad, mlp::kIn, gw[7], mlp::kIn + W, gb[7]);
            add(dZ(5), W, W, H(4), W, W, gw[7] + mlp::kIn, mlp::kIn + W, nullptr);
        } else {
            add(dZ(l), W, W, H(l - 1), W, W, gw[l + 2], W, gb[l + 2]);
        }
    }
    add(dZ(0), W, W, saved + mlp::sv_inp(M), mlp::kInPad, mlp::kIn, gw[2], mlp::kIn, gb[2]);
    add(scratch + mlp::sc_dt2(M), 32, mlp::kTOut, saved + mlp::sv_t1(M), W, W, gw[1], W, gb[1]);
    add(scratch + mlp::sc_dt1(M), W, W, saved + mlp::sv_et(M), mlp::kTPad, mlp::kTCh, gw[0], mlp::kTCh, gb[0]);
    g.ndesc = ndesc;
    block = mlp::wg_place(g);
    if (getenv("WG_GRID")) block = atoi(getenv("WG_GRID"));   // fewer workgroups: throughput- or latency-bound?
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int pass = 0; pass < 3; pass++) {
        for (int i = 0; i < 5; i++) {
            if (pass == 0) hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, a);
            else if (pass == 1) hipLaunchKernelGGL(mlp::mlp_bwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, b);
            else if (M == 1024) hipLaunchKernelGGL(mlp::mlp_wgrad_kernel, dim3(block), dim3(mlp::kWgThreads), 0, 0, g);
            else hipLaunchKernelGGL(mlp::mlp_wgrad_kernel, dim3(block), dim3(mlp::kWgThreads), 0, 0, g);
        }
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; i++) {
            if (pass == 0) hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, a);
            else if (pass == 1) hipLaunchKernelGGL(mlp::mlp_bwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, b);
            else if (M == 1024) hipLaunchKernelGGL(mlp::mlp_wgrad_kernel, dim3(block), dim3(mlp::kWgThreads), 0, 0, g);
            else hipLaunchKernelGGL(mlp::mlp_wgrad_kernel, dim3(block), dim3(mlp::kWgThreads), 0, 0, g);
        }
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch (M = %d, %d workgroups)\n", pass == 0 ? "forward chain" : pass == 1 ? "backward chain" : "weight gradients", ms * 1e3 / iters, M,
               pass < 2 ? M / mlp::kRows : block);
    }
#ifdef DGS_MLP_TRACE
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, 4);
    hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, a);
    CK(hipDeviceSynchronize());
    long long tr[64]; int n;
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr)); hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), 4);
    printf("forward stages of workgroup 0 (us since its first stamp):");
    for (int i = 0; i < n; i++) printf(" %.2f", (tr[i] - tr[0]) * 0.01);
    printf("\n");
#endif
    return 0;
}
