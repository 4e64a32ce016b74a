// Issue cost of the double-precision instructions k_fem_cg_xcd is made of, in shader clocks per wave instruction: 8 independent chains,
// 256-thread workgroup (one wave per SIMD) and two such workgroups on one compute unit (two waves per SIMD).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_rates tools/ubench/f64_rates.hip && /tmp/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 64, REP = 64;
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *clk, float seed)
{
    double a[8]; float f[8];
    __shared__ double sh[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sh[i] = i * 0.5;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; f[i] = seed * i + threadIdx.x; }
    int idx = (threadIdx.x * 3) & 2047;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int j = 0; j < N / 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == 4) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
                if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 6) { asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(idx * 8 + 64 * i)); }
                if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
        if (OP == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, double *out, unsigned long long *clk)
{
    for (int nb : {1, 2}) {                          // blocks: both land on one compute unit? not promised -- so launch 2 x 256 CUs' worth and read block 0
        const int grid = nb == 1 ? 1 : 512;
        k<OP><<<grid, 256>>>(out, clk, 1.5f);
        k<OP><<<grid, 256>>>(out, clk, 1.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        printf("%-16s %s: %.2f clocks per wave instruction (block 0), %.2f (slowest block)\n", name, nb == 1 ? "1 wave/SIMD " : "2 waves/SIMD", (double)h[0] / (N * REP), (double)mx / (N * REP));
    }
}
int main()
{
    double *out; unsigned long long *clk;
    hipMalloc(&out, sizeof(double) * 512 * 256); hipMalloc(&clk, sizeof(unsigned long long) * 512);
    run<0>("v_add_f64", out, clk); run<1>("v_mul_f64", out, clk); run<2>("v_fma_f64", out, clk); run<3>("v_cvt_f64_f32", out, clk);
    run<4>("v_mov_b32_dpp", out, clk); run<5>("v_add_f32", out, clk); run<6>("ds_read_b64", out, clk); run<7>("v_xor_b32", out, clk);
    return 0;
}
