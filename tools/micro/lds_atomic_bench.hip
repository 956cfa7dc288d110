// micro-benchmark: throughput of LDS atomics (ds_add_f32 / ds_add_u32 / plain RMW) under the access pattern of lbs_bwd_kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kM = 1024, kG = 23, kThreads = 512, kBlocks = 256, kPerThread = 2, kVals = 69;
template <int MODE>
__global__ void __launch_bounds__(kThreads) bench(const int* __restrict__ idx, float* out)
{
    extern __shared__ float s_tab[];
    for (int i = threadIdx.x; i < kM * kG; i += kThreads) s_tab[i] = 0.f;
    __syncthreads();
    for (int it = 0; it < kPerThread; it++) {
        const int n = (blockIdx.x * kPerThread + it) * kThreads + threadIdx.x;
        const int j0 = idx[3 * n], j1 = idx[3 * n + 1], j2 = idx[3 * n + 2];
        const int js[3] = {j0, j1, j2};
        float v = (float)n * 1e-6f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* acc = s_tab + js[k] * kG;
#pragma unroll
            for (int c = 0; c < kG; c++) {
                if (MODE == 0) atomicAdd(acc + c, v);
                if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(acc + c), (unsigned)__float_as_uint(v) & 0xffu);
                if (MODE == 2) acc[c] += v;   // racy plain RMW: lower bound (ds_read + ds_write)
                if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(s_tab) + ((js[k] * kG + c) >> 1), 1ull);
                v += 1e-7f;
            }
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < kM * kG; i += kThreads) s += s_tab[i];
    if (s == 123456.f) out[blockIdx.x] = s;
}
template <int MODE>
float run(const int* idx, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(bench<MODE>, dim3(kBlocks), dim3(kThreads), kM * kG * 4, 0, idx, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(bench<MODE>, dim3(kBlocks), dim3(kThreads), kM * kG * 4, 0, idx, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 20;
}
int main()
{
    const int N = kBlocks * kPerThread * kThreads;
    std::vector<int> h(3 * N), hs(3 * N);
    unsigned s = 12345;
    for (int i = 0; i < 3 * N; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 10) % kM; }
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) hs[3 * i + k] = (i / 64 * 3 + k) % kM;   // whole wave on one node (sorted surfels)
    int *d, *ds; float* out;
    hipMalloc(&d, h.size() * 4); hipMalloc(&ds, h.size() * 4); hipMalloc(&out, kBlocks * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    printf("%d blocks x %d threads x %d points x %d atomics\n", kBlocks, kThreads, kPerThread, kVals);
    printf("random nodes : ds_add_f32 %.1f us | ds_add_u32 %.1f us | plain rmw %.1f us | ds_add_u64 %.1f us\n", run<0>(d, out), run<1>(d, out), run<2>(d, out), run<3>(d, out));
    printf("same node/wave: ds_add_f32 %.1f us | ds_add_u32 %.1f us | plain rmw %.1f us\n", run<0>(ds, out), run<1>(ds, out), run<2>(ds, out));
    return 0;
}
