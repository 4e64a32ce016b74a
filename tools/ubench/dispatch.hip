// Micro-benchmark: workgroup dispatch rate on MI355X for small workgroups.
// Grid = 89,600 one-wave workgroups (the FAST launch of a 64-frame 640x480 batch) with
// ~6 KB of LDS, against the same waves packed 4 or 16 per workgroup, empty bodies and
// bodies with N dependent VALU instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__global__ __launch_bounds__(T) void k(unsigned *out, int work)
{
    extern __shared__ unsigned sm[];
    unsigned a = threadIdx.x + blockIdx.x;
    for (int i = 0; i < work; ++i) a = a * 5u + 1u;
    if (work < 0) sm[threadIdx.x] = a;
    if (a == 0x12345u) out[0] = a + sm[0];
}
template <int T> void run(int waves, int lds_per_wave, int work)
{
    unsigned *d; hipMalloc(&d, 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wg = waves / (T / 64);
    const int lds = lds_per_wave * (T / 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<T>, dim3(wg), dim3(T), lds, 0, d, work);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<T>, dim3(wg), dim3(T), lds, 0, d, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("T=%4d wgs=%6d lds/wg=%6d work=%5d : %8.2f us/launch  %7.1f waves/us\n", T, wg, lds, work, ms / reps * 1e3, waves / (ms / reps * 1e3));
    hipFree(d);
}
int main()
{
    const int waves = 89600;
    for (int work : {0, 250, 1000}) {
        run<64>(waves, 0, work);
        run<64>(waves, 6144, work);
        run<256>(waves, 6144, work);
        run<1024>(waves, 6144, work);
    }
    return 0;
}
