// Micro-benchmark: issue cost of packed-f32 vector instructions (v_pk_mul_f32 / v_pk_add_f32, as k_describe's rotation uses them)
// against their scalar forms, 3 waves per SIMD on the whole chip: ns per wave64 instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    f2 a = {1.0001f + threadIdx.x * 1e-6f, 0.9999f}, b = {1.00001f, 0.99999f}, c = {1e-7f, -1e-7f};
    float s0 = a.x, s1 = a.y;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int v = 0; v < 32; ++v) {
            if (MODE == 0) { a = a * b; a = a + c; }                                  // 2 packed instructions
            if (MODE == 1) { s0 = s0 * b.x; s1 = s1 * b.y; s0 = s0 + c.x; s1 = s1 + c.y; }   // the same work, 4 scalar ones
            if (MODE == 2) { s0 = s0 * b.x; s0 = s0 + c.x; }                           // 2 scalar instructions
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y + s0 + s1;
}
template <int MODE> void run(const char *name, int per_iter)
{
    float *d; (void)hipMalloc(&d, 64 * 3072 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(3072), dim3(64), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    const int iters = 2000;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(3072), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.2f ns per instruction and SIMD\n", name, ms, ms * 1e6 / ((double)iters * per_iter * 3));
    (void)hipFree(d);
}
int main()
{
    run<0>("v_pk_mul_f32 + v_pk_add_f32 (64 per iteration)", 64);
    run<1>("4 scalar mul / add for the same work (128)", 128);
    run<2>("v_mul_f32 + v_add_f32 (64)", 64);
    return 0;
}
