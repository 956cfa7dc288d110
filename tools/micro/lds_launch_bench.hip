// micro-benchmark: what does a workgroup's LDS allocation cost at launch on gfx950?  (Nothing: 2.5 us per launch from 16 KB to 160 KB --
// written to test whether the ~73 us floor of round 2's 132-KB sort kernel came from its allocation; it does not.)  Kernels that do nothing but touch one LDS word,
// static allocations from 16 KB to 160 KB, grids of 36 / 288 / 2048 workgroups of 256 threads; time per launch from HIP events
// over 50 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_launch_bench.hip -o tools/micro/lds_launch_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KB>
__global__ void __launch_bounds__(256) touch(int* out, int flag)
{
    __shared__ int s[KB * 256];
    if (flag) { s[threadIdx.x * KB] = flag; __syncthreads(); out[blockIdx.x] = s[(threadIdx.x * KB + 7) % (KB * 256)]; }
}

template <int KB>
static void run(int* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%4d KB:", KB);
    const int grids[3] = {36, 288, 2048};
    for (int g : grids) {
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(touch<KB>, dim3(g), dim3(256), 0, 0, out, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(touch<KB>, dim3(g), dim3(256), 0, 0, out, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  grid %4d: %7.1f us", g, ms * 1e3 / 50);
    }
    printf("\n");
}

int main()
{
    int* out;
    hipMalloc(&out, 4096 * 4);
    run<16>(out); run<32>(out); run<48>(out); run<60>(out); run<64>(out); run<65>(out); run<72>(out); run<88>(out); run<96>(out);
    run<128>(out); run<132>(out); run<144>(out); run<160>(out);
    return 0;
}
