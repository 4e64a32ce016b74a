// Micro-benchmark: how far do v_mfma_f32_32x32x64_f8f6f4 (FP4 operands) and VALU instructions of the SAME SIMD overlap on MI355X?
// The all-pairs matcher folds every 32 x 32 tile of MFMA results with 2 VALU instructions per element; its PMC pass shows neither
// pipe saturated (VALU issue 0.45, MFMA busy 0.26).  Cases, each at 1 / 2 / 3 waves per SIMD, in shader clocks (s_memtime) per
// loop iteration of one wave (an iteration = 4 chained MFMAs + NV VALU instructions):
//   mfma only | valu only | mfma + VALU that does not touch the accumulators | mfma + the matcher's fold of the 16 results
// Build twice: plain (accumulators in AGPRs when the compiler likes) and with -mllvm -amdgpu-mfma-vgpr-form=1.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_agpr mfma_overlap.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_overlap_vgpr mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only, 1: VALU only, 2: MFMA + independent VALU, 3: MFMA + fold of the results (2 per element), 4: as 3 with two
// chains in flight (the fold of one runs under the MFMAs of the other)
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k(float *out, long long *clk, int iters)
{
    v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, b = {0x11111111, 0x19191919, 0x11991199, 0x11119999, 0, 0, 0, 0};
    a[0] += threadIdx.x & 1;
    v16f acc, acc2, c;
    for (int g = 0; g < 16; ++g) { c[g] = 128.0f + (float)g / 32768.0f; acc[g] = c[g]; acc2[g] = c[g]; }
    unsigned k1 = 0x7f000000u, k2 = 0x7f000000u, j1 = 0x7f000000u + threadIdx.x, j2 = 0x7e000000u;
    float w = (float)threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE != 1) {
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < 4; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int v = 0; v < NV / 2; ++v) {
                j2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(j1), w, __uint_as_float(j2)));
                j1 = j1 < j2 ? j1 : j2;
                w += 1.0f;
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                k2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(k1), acc[g], __uint_as_float(k2)));
                const unsigned u = __float_as_uint(acc[g]);
                k1 = k1 < u ? k1 : u;
            }
        }
        if (MODE == 4) {
            acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if ((g & 3) == 3 && g < 15) acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc2, 4, 4, 0, 0, 0, 0);
                k2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(k1), acc[g], __uint_as_float(k2)));
                const unsigned u = __float_as_uint(acc[g]);
                k1 = k1 < u ? k1 : u;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                k2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(k1), acc2[g], __uint_as_float(k2)));
                const unsigned u = __float_as_uint(acc2[g]);
                k1 = k1 < u ? k1 : u;
            }
        }
        c[0] += 1.0f;
    }
    const long long t1 = clock64();
    float s = __uint_as_float(k1) + __uint_as_float(k2) + __uint_as_float(j1) + __uint_as_float(j2);
    for (int g = 0; g < 16; ++g) s += acc[g] + acc2[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV> void run(const char *name)
{
    float *d; long long *c; hipMalloc(&d, 64 * 4096 * 4); hipMalloc(&c, 4096 * 8);
    static long long h[4096];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-46s", name);
    for (int wps : {1, 2, 3}) {
        const int grid = 1024 * wps, iters = 4000;
        hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, d, c, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, d, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, grid * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < grid; ++i) sum += (double)h[i];
        const double per_iter = sum / grid / iters;                      // clocks of one wave per iteration
        printf("  w/SIMD=%d: %7.1f clk/iter/wave = %6.1f per SIMD-iteration (%.3f ms)", wps, per_iter, per_iter / wps, ms);
    }
    printf("\n");
    hipFree(d); hipFree(c);
}
int main()
{
    run<0, 0>("4 MFMA (one chain)");
    run<1, 32>("32 VALU");
    run<1, 48>("48 VALU");
    run<2, 32>("4 MFMA + 32 VALU not touching the results");
    run<2, 48>("4 MFMA + 48 VALU not touching the results");
    run<3, 32>("4 MFMA, then fold of the 16 results (32 VALU)");
    run<4, 64>("2 x (4 MFMA + fold), fold under the other chain");
    return 0;
}
