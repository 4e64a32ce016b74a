// Micro-benchmark: cost of cooperative_groups grid.sync() on MI355X (hipLaunchCooperativeKernel), 32 / 64 / 256 workgroups of 256 threads.
// Measured in round 4: 7.5-10 us per sync at 32-64 workgroups, 24 us at 256 -- no cheaper than a kernel launch, so a one-mesh CG spread over
// several compute units with two grid syncs per iteration cannot beat the 11-13 us per iteration of the launch-per-phase path.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k(int *out, int iters)
{
    cg::grid_group g = cg::this_grid();
    int v = 0;
    for (int i = 0; i < iters; ++i) { if (threadIdx.x == 0) atomicAdd(out, 1); g.sync(); v += *out; g.sync(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = v;
}
int main()
{
    int dev = 0, coop = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    printf("cooperative launch attribute: %d\n", coop);
    int *d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    int iters = 1000; void *args[] = {&d, &iters};
    for (int grid : {32, 64, 256}) {
        hipMemset(d, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((void *)k, dim3(grid), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("grid %d: launch %s, %.3f ms for %d iterations x 2 grid syncs = %.2f us per sync, counter %d (expect %d)\n", grid, hipGetErrorString(e), ms, iters, ms * 1e3 / (2 * iters), h[0], grid * iters);
    }
    return 0;
}
