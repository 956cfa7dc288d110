// Lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950, with and without the A-block broadcast (cbsz = 4, abid = g), checked against
// the map node_mlp.h relies on:  A[blk][i] in lane 4*blk + i,  B[blk][j] in lane 4*blk + j,  D[blk][i][j] in VGPR i of lane 4*blk + j;
// with cbsz = 4 every block uses the A of block abid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void probe(const float* a, const float* b, float* d)
{
    int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
    for (int v = 0; v < 4; v++) d[v * 64 + l] = acc[v];
}
int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; l++) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    int bad_total = 0;
    auto check = [&](const char* name, int abid) {   // abid < 0: no broadcast
        hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++) {
                int blk = l >> 2, j = l & 3, ablk = abid < 0 ? blk : abid;
                float want = ha[4 * ablk + v] * hb[4 * blk + j];
                if (hd[v * 64 + l] != want) bad++;
            }
        printf("%s: %d mismatches\n", name, bad);
        bad_total += bad;
    };
    hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, a, b, d); check("no broadcast", -1);
    hipLaunchKernelGGL((probe<4, 0>), dim3(1), dim3(64), 0, 0, a, b, d); check("cbsz=4 abid=0", 0);
    hipLaunchKernelGGL((probe<4, 1>), dim3(1), dim3(64), 0, 0, a, b, d); check("cbsz=4 abid=1", 1);
    hipLaunchKernelGGL((probe<4, 3>), dim3(1), dim3(64), 0, 0, a, b, d); check("cbsz=4 abid=3", 3);
    return bad_total != 0;
}
