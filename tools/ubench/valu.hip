// Micro-benchmark: sustained integer VALU rate on MI355X (xor+bcnt chain, min/max, mul_i24)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3u, a3 = a0 + 77u;
    unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { // xor + bcnt
                acc0 = __popc(a0 ^ (acc1 + u)) + acc0; acc1 = __popc(a1 ^ acc2) + acc1;
                acc2 = __popc(a2 ^ acc3) + acc2; acc3 = __popc(a3 ^ acc0) + acc3;
            } else if (MODE == 1) { // min/max
                acc0 = min(acc0 + 3u, a0 ^ acc1); acc1 = max(acc1, a1 + acc2); acc2 = min(acc2 + 5u, a2 ^ acc3); acc3 = max(acc3, a3 + acc0);
            } else { // float fma
                float f0 = __uint_as_float(acc0), f1 = __uint_as_float(acc1), f2 = __uint_as_float(acc2), f3 = __uint_as_float(acc3);
                f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + f2; f2 = f2 * 1.0001f + f3; f3 = f3 * 0.9999f + f0;
                acc0 = __float_as_uint(f0); acc1 = __float_as_uint(f1); acc2 = __float_as_uint(f2); acc3 = __float_as_uint(f3);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
template <int MODE> void run(const char *name, int ops_per_inner)
{
    unsigned *d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocksPerCU : {1, 2, 4, 8}) {
        int grid = 256 * blocksPerCU, iters = 4000;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)grid * 4 * iters * 16 * ops_per_inner; // wave-instructions
        printf("%s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s  => %.2f clk/instr/SIMD @2.4GHz\n", name, blocksPerCU, ms,
               winstr / ms / 1e6, 1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
    }
}
int main() { run<0>("xor+bcnt", 8 + 1); run<1>("minmax+add/xor", 8); run<2>("fma_f32", 4); return 0; }
