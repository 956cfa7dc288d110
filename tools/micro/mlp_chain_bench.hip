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
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 5; i++) {
            if (pass == 0) hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, a);
            else hipLaunchKernelGGL(mlp::mlp_bwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, b);
        }
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; i++) {
            if (pass == 0) hipLaunchKernelGGL(mlp::mlp_fwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, a);
            else hipLaunchKernelGGL(mlp::mlp_bwd_kernel, dim3(M / mlp::kRows), dim3(mlp::kThreads), 0, 0, b);
        }
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s chain: %.1f us per launch (M = %d, %d workgroups)\n", pass ? "backward" : "forward", ms * 1e3 / iters, M, M / mlp::kRows);
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
