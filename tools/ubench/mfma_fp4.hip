// Micro-benchmark: issue interval of v_mfma_f32_32x32x64_f8f6f4 on MI355X with FP4 / FP8 operands, one chain and
// four independent chains, 1 and 2 waves per SIMD; and the same with 32 VALU min/med3 per MFMA group mixed in.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int FMT, int CHAINS, int VALU>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, b = {0x11111111, 0x19191919, 0x11991199, 0x11119999, 0, 0, 0, 0};
    if (FMT == 0) { a[4] = a[0]; a[5] = a[1]; a[6] = a[2]; a[7] = a[3]; b[4] = b[0]; b[5] = b[1]; b[6] = b[2]; b[7] = b[3]; }
    a[0] += threadIdx.x & 1;
    v16f acc[CHAINS];
    unsigned k1 = 0x7f000000u, k2 = 0x7f000000u;
    for (int c = 0; c < CHAINS; ++c)
        for (int g = 0; g < 16; ++g) acc[c][g] = (float)(c + g);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], FMT, FMT, 0, 0, 0, 0);
            if (VALU) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float x = acc[c][g];
                    k2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(k1), x, __uint_as_float(k2)));
                    const unsigned u = __float_as_uint(x);
                    k1 = k1 < u ? k1 : u;
                }
            }
        }
    }
    float s = __uint_as_float(k1) + __uint_as_float(k2);
    for (int c = 0; c < CHAINS; ++c)
        for (int g = 0; g < 16; ++g) s += acc[c][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FMT, int CHAINS, int VALU> void run(const char *name)
{
    float *d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocksPerCU : {1, 2}) {
        int grid = 256 * blocksPerCU, iters = 2000;
        hipLaunchKernelGGL((k<FMT, CHAINS, VALU>), dim3(grid), dim3(256), 0, 0, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FMT, CHAINS, VALU>), dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = (double)grid * 4 * iters * CHAINS * 4; // MFMAs
        printf("%-34s waves/SIMD=%d  %.3f ms  %.1f clk per MFMA per SIMD @2.4GHz  (%.2f PFLOP/s)\n", name, blocksPerCU, ms,
               1024.0 * 2.4e9 * ms * 1e-3 / n, n * 131072.0 / ms / 1e12);
    }
    hipFree(d);
}
int main()
{
    run<4, 1, 0>("fp4 1 chain");
    run<4, 4, 0>("fp4 4 chains");
    run<0, 4, 0>("fp8 4 chains");
    run<4, 4, 1>("fp4 4 chains + 32 VALU per 4 MFMA");
    run<4, 1, 1>("fp4 1 chain + 32 VALU per 4 MFMA");
    return 0;
}
