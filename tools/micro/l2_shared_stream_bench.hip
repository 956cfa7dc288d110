// Every workgroup streams the SAME buffer (the situation of a kernel whose workgroups all need all weights): achieved
// bytes/s per CU and in total, by workgroup count and buffer size.  1024 threads, 16-byte loads, `depth` loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH>
__global__ void __launch_bounds__(1024) stream(const float4* __restrict__ buf, int nvec, int reps, float* out)
{
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < reps; r++)
        for (int i = threadIdx.x; i < nvec; i += 1024 * DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) v[d] = buf[i + d * 1024 < nvec ? i + d * 1024 : i];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) { s.x += v[d].x; s.y += v[d].y; s.z += v[d].z; s.w += v[d].w; }
        }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = s.x;
}
int main()
{
    float4* buf; float* out;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes_kb[] = {256, 2048, 8192};
    const int wgs[] = {8, 32, 64, 128, 256, 512};
    for (int skb : sizes_kb)
        for (int g : wgs) {
            int nvec = skb * 1024 / 16, reps = skb <= 256 ? 16 : 2;
            hipLaunchKernelGGL(stream<8>, dim3(g), dim3(1024), 0, 0, buf, nvec, reps, out);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(stream<8>, dim3(g), dim3(1024), 0, 0, buf, nvec, reps, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 100.0, bytes = (double)skb * 1024 * reps;
            printf("buffer %5d KB x %2d passes, %3d workgroups: %7.1f us/launch  %6.1f GB/s per workgroup  %6.2f TB/s total\n", skb, reps, g, us,
                   bytes / us * 1e-3, bytes * g / us * 1e-6);
        }
    return 0;
}
