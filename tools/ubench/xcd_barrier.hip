// Micro-benchmark for a one-mesh CG spread over the compute units of ONE XCD: P workgroups of a 256-workgroup launch
// (blockIdx % 8 == 0: dealt round-robin, they share an XCD -- for speed only, the protocol does not rely on it) run a loop of
//   each: sc1-store its 2-KB slice of a 53-KB vector; every wave s_waitcnt vmcnt(0); __syncthreads; lane 0: agent-scope atomic add
//   each: poll the counter with sc1 loads (one wave), __syncthreads; sc1-load the WHOLE vector into LDS; check every value
// which is MI355X_MICROARCH.md's measured-valid hand-off (one lane signals for all the workgroup's stores, sc1 both sides, one
// workgroup per CU).  Reports us per barrier + staging, stale values seen, the participants' XCC ids.  Every spin is bounded.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_barrier.hip -o tools/ubench/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int T = 256, N = 6656, P_MAX = 32;          // N doubles = 53 KB, a multiple of 2 x T
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_for(unsigned *ctr, unsigned target, unsigned *abort_flag)
{
    bool ok = true;
    if (threadIdx.x < 64) {
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        }
        if (!ok) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __syncthreads_and(ok);
}
__global__ __launch_bounds__(T) void k(double *vec, unsigned *ctr, unsigned *abort_flag, int P, int stride, int iters, int stage, unsigned *out, int plain)
{
    extern __shared__ double lds[];
    if (blockIdx.x % stride != 0 || (int)(blockIdx.x / stride) >= P) return;
    const int r = blockIdx.x / stride, tid = threadIdx.x;
    if (tid == 0) out[2 + r] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID (id 20), bits 0-3
    const int per = N / P, lo = r * per;
    unsigned stale = 0, bar = 0;
    for (int it = 0; it < iters; ++it) {
        if (plain) { for (int i = tid; i < per; i += T) vec[lo + i] = (double)(it * 7 + 1) + (double)(lo + i) * 1e-6; }   // plain stores: the line stays in this XCD's L2
        else for (int i = tid; i < per; i += T) st_sc1(vec + lo + i, (double)(it * 7 + 1) + (double)(lo + i) * 1e-6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_for(ctr, (unsigned)P * ++bar, abort_flag)) return;
        if (stage == 1) {         // 8-byte sc1 loads, 13 in flight per lane (a plain `lds[i] = ld_sc1(..)` loop waits for every load: 7.7 us)
            constexpr int U = 13;
            for (int b = 0; b < N; b += U * T) {
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld_sc1(vec + min(b + u * T + tid, N - 1));
#pragma unroll
                for (int u = 0; u < U; ++u) if (b + u * T + tid < N) lds[b + u * T + tid] = v[u];
            }
            __syncthreads();
        } else if (stage == 2) {  // 16-byte sc1 buffer loads, 13 in flight per lane
            constexpr int U = 13;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(vec, 0, N * 8, 0x00020000);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * min(u * T + tid, N / 2 - 1), 0, 1 << 4);   // aux bit 4: sc1
#pragma unroll
            for (int u = 0; u < U; ++u) if (u * T + tid < N / 2) reinterpret_cast<u32x4 *>(lds)[u * T + tid] = v[u];
            __syncthreads();
        }
        if (stage) {
            for (int i = tid; i < per * P; i += T) stale += lds[i] != (double)(it * 7 + 1) + (double)i * 1e-6;
        }
        // second barrier of the iteration: nobody overwrites the vector before everybody has read it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_for(ctr, (unsigned)P * ++bar, abort_flag)) return;
    }
    atomicAdd(out, stale);
    if (tid == 0 && r == 0) out[1] = 1;
}
int main()
{
    double *vec; unsigned *ctr, *out;
    hipMalloc(&vec, sizeof(double) * N); hipMalloc(&ctr, 64); hipMalloc(&out, 4 * (2 + P_MAX));
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(double) * N);
    const int iters = 20000;
    for (int plain : {0, 1}) for (int stride : {8, 1}) for (int P : {32}) for (int stage : {0, 2}) {
        hipMemsetAsync(ctr, 0, 64, st); hipMemsetAsync(out, 0, 4 * (2 + P_MAX), st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(k, dim3(256), dim3(T), sizeof(double) * N, st, vec, ctr, ctr + 8, P, stride, iters, stage, out, plain);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h(2 + P_MAX);
        hipMemcpyAsync(h.data(), out, 4 * (2 + P_MAX), hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        printf("plain-stores %d stride %d P %2d stage %d: %.3f ms / %d iterations = %.2f us per iteration (2 barriers%s), finished %u, stale %u, xcc ids:", plain, stride, P, stage, ms,
               iters, ms * 1e3 / iters, stage ? " + 53 KB staged" : "", h[1], h[0]);
        for (int i = 0; i < P; ++i) printf(" %u", h[2 + i]);
        printf("\n");
    }
    return 0;
}
