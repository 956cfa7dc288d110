// micro-benchmark: issue rate of the VALU / cross-lane / matrix instructions the blend kernels are built from, on gfx950.
// Every kernel runs kIters x 16 independent instances of one instruction per wave; the table printed is
// SIMD cycles per wave64 instruction at 1, 2 and 4 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate_bench.hip -o gpurun_out/valu_rate_bench && gpurun_out/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kIters = 16384;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Mode { FMA, PKFMA, PKFMA_BCAST, PKMUL, PKADD, EXP, RCP, CNDMASK, BPERMUTE, DPP_ADD, PERMLANE32, MFMA16, MFMA16_PLUS_FMA, FMA_SGPR, MUL, LDS_B128_BCAST, FMA_DEP4, MFMA4x4, CND_E64, CMP_CND, CMP_VCC, CMP_E64, VMAX, VAND, VMOV, READLANE, READFIRST, CND_VCC_INIT, NMODES };
static const char* kNames[NMODES] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel bcast", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
                                     "v_cndmask_b32", "ds_bpermute_b32", "v_add_f32 dpp row_shr:1", "v_permlane32_swap", "v_mfma_f32_16x16x4_f32",
                                     "mfma16x16x4 + 4 v_fma (per pair)", "v_fma_f32 sgpr src", "v_mul_f32", "ds_read_b128 uniform addr", "v_fma_f32 4 chains (dep)", "v_mfma_f32_4x4x1_16B_f32", "v_cndmask_b32_e64 sgpr mask", "v_cmp_gt_f32 vcc + v_cndmask vcc (pair)", "v_cmp_gt_f32 -> vcc", "v_cmp_gt_f32_e64 -> sgpr", "v_max_f32", "v_and_b32", "v_mov_b32", "v_readlane_b32", "v_readfirstlane_b32", "v_cndmask_b32 vcc (vcc set once)"};

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, float seed, int iters)
{
    __shared__ float4v s_buf[256];
    const int tid = threadIdx.x;
    s_buf[tid] = float4v{seed, seed, seed, seed};
    __syncthreads();
    float a[16];
    float2v p[16];
    float4v m4[4];
    float x = seed + tid * 1e-9f, y = 1.0f + seed;
    float2v x2 = {x, y}, y2 = {y, x};
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed * i; p[i] = float2v{seed * i, seed}; }
#pragma unroll
    for (int i = 0; i < 4; i++) m4[i] = float4v{seed, seed, seed, seed};
    const float sg = __builtin_amdgcn_readfirstlane(seed);
    int addr = (tid ^ 1) * 4;
    unsigned long long smask;
    asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(smask) : "v"(x), "v"(y));
    unsigned long long sm[4] = {0, 0, 0, 0};
    int sr[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if (MODE == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == FMA_DEP4) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(x), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
            REP16(X)
#undef X
        } else if (MODE == FMA_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sg), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(y2));
            REP16(X)
#undef X
        } else if (MODE == PKFMA_BCAST) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(x2), "v"(y2));
            REP16(X)
#undef X
        } else if (MODE == PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(y2));
            REP16(X)
#undef X
        } else if (MODE == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(y2));
            REP16(X)
#undef X
        } else if (MODE == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : );
            REP16(X)
#undef X
        } else if (MODE == BPERMUTE) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(addr));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (MODE == DPP_ADD) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == PERMLANE32) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (MODE == MFMA16) {
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m4[i & 3]) : "v"(x), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == MFMA4x4) {
#define X(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m4[i & 3]) : "v"(x), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == MFMA16_PLUS_FMA) {
            // 8 pairs of (1 MFMA + 4 FMA): does the matrix pipe run under the VALU stream for free?
#define X(i)                                                                                               \
    if ((i) < 8) {                                                                                         \
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m4[i & 3]) : "v"(x), "v"(y));           \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(2 * (i)) & 15]) : "v"(x), "v"(y));                \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(2 * (i) + 1) & 15]) : "v"(x), "v"(y));            \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(2 * (i) + 8) & 15]) : "v"(x), "v"(y));            \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(2 * (i) + 9) & 15]) : "v"(x), "v"(y));            \
    }
            REP16(X)
#undef X

        } else if (MODE == CND_E64) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "s"(smask));
            REP16(X)
#undef X
        } else if (MODE == CND_VCC_INIT) {
            if (it == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : );
            REP16(X)
#undef X
        } else if (MODE == CMP_CND) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
            REP16(X)
#undef X
        } else if (MODE == CMP_VCC) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(y) : "vcc");
            REP16(X)
#undef X
        } else if (MODE == CMP_E64) {
#define X(i) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(a[i]), "v"(y));
            REP16(X)
#undef X
        } else if (MODE == VMAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            REP16(X)
#undef X
        } else if (MODE == VAND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            REP16(X)
#undef X
        } else if (MODE == VMOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(y));
            REP16(X)
#undef X
        } else if (MODE == READLANE) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr[i & 3]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == READFIRST) {
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr[i & 3]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == LDS_B128_BCAST) {
            const int ua = __builtin_amdgcn_readfirstlane(it & 255) * 16;
#define X(i)                                                                                   \
    if ((i) < 4) { asm volatile("ds_read_b128 %0, %1" : "=v"(m4[i]) : "v"(ua + (i) * 16)); }
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    float s = x + (float)(sm[0] + sm[1] + sm[2] + sm[3]) + (float)(sr[0] + sr[1] + sr[2] + sr[3]);
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 4; i++) s += m4[i].x + m4[i].y + m4[i].z + m4[i].w;
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

// wave instructions issued per loop iteration
static int insts_per_iter(int mode)
{
    if (mode == MFMA16_PLUS_FMA) return 40;
    if (mode == LDS_B128_BCAST) return 4;
    if (mode == CMP_CND) return 32;
    return 16;
}

template <int MODE>
static void run(float* out, int n_cu, double clock_hz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-36s", kNames[MODE]);
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int grid = n_cu * wps;
        hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, out, 0.f, 64);
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, out, 0.f, kIters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double sec = ms * 1e-3 / 3;
        const double inst_per_simd = (double)kIters * insts_per_iter(MODE) * wps;  // each SIMD runs wps waves
        printf("  %d w/SIMD: %6.2f cyc/inst", wps, sec * clock_hz / inst_per_simd);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clock_hz = prop.clockRate * 1e3;
    printf("%s: %d CUs, %.0f MHz (cycles below assume this clock)\n", prop.gcnArchName, n_cu, clock_hz * 1e-6);
    float* out;
    hipMalloc(&out, (size_t)n_cu * 4 * 256 * 4);
    run<FMA>(out, n_cu, clock_hz);
    run<FMA_DEP4>(out, n_cu, clock_hz);
    run<MUL>(out, n_cu, clock_hz);
    run<FMA_SGPR>(out, n_cu, clock_hz);
    run<PKFMA>(out, n_cu, clock_hz);
    run<PKFMA_BCAST>(out, n_cu, clock_hz);
    run<PKMUL>(out, n_cu, clock_hz);
    run<PKADD>(out, n_cu, clock_hz);
    run<EXP>(out, n_cu, clock_hz);
    run<RCP>(out, n_cu, clock_hz);
    run<CNDMASK>(out, n_cu, clock_hz);
    run<BPERMUTE>(out, n_cu, clock_hz);
    run<DPP_ADD>(out, n_cu, clock_hz);
    run<PERMLANE32>(out, n_cu, clock_hz);
    run<MFMA16>(out, n_cu, clock_hz);
    run<MFMA4x4>(out, n_cu, clock_hz);
    run<MFMA16_PLUS_FMA>(out, n_cu, clock_hz);
    run<LDS_B128_BCAST>(out, n_cu, clock_hz);
    run<CND_E64>(out, n_cu, clock_hz);
    run<CND_VCC_INIT>(out, n_cu, clock_hz);
    run<CMP_CND>(out, n_cu, clock_hz);
    run<CMP_VCC>(out, n_cu, clock_hz);
    run<CMP_E64>(out, n_cu, clock_hz);
    run<VMAX>(out, n_cu, clock_hz);
    run<VAND>(out, n_cu, clock_hz);
    run<VMOV>(out, n_cu, clock_hz);
    run<READLANE>(out, n_cu, clock_hz);
    run<READFIRST>(out, n_cu, clock_hz);
    return 0;
}
