// common.h -- host-side helpers shared by the C-ABI translation units.
#ifndef ORBX_COMMON_H
#define ORBX_COMMON_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/orbslam_hip.h"

namespace orbx {

// Last error text, per thread (the ABI never throws).
inline std::string &last_error()
{
    static thread_local std::string s;
    return s;
}

inline int set_error(int code, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
    last_error() = buf;
    return code;
}

// True when a HIP device is usable.  Probed once; there is no CPU fallback.
inline bool device_ok()
{
    static int state = -1;
    if (state < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        state = (e == hipSuccess && n > 0) ? 1 : 0;
        (void)hipGetLastError();
    }
    return state == 1;
}

// Optional per-kernel timing with HIP events recorded on the launch stream
// (bench.py's roofline leg).  start(kind)/stop(kind) bracket one launch and only
// record when `kind` is in `mask`: every recorded event costs a few microseconds
// of dispatch gap, so the timed region enables just the kernel it reports.
// flush() synchronises and accumulates.
struct KernelProfiler {
    static constexpr int MAXK = 16, MAXEV = 4096;
    unsigned mask = 0;
    const char *names[MAXK] = {nullptr};
    double ms[MAXK] = {0};
    long launches[MAXK] = {0};
    hipEvent_t ev[MAXEV];
    int kind[MAXEV]; // start: kind, stop: kind | 0x100
    int n = 0, created = 0;

    void rec(int tag, hipStream_t st)
    {
        if (n == created) { (void)hipEventCreate(&ev[created]); ++created; }
        kind[n] = tag;
        (void)hipEventRecord(ev[n++], st);
    }
    void start(int k, hipStream_t st)
    {
        if (!((mask >> k) & 1u)) return;
        if (n + 4 >= MAXEV) flush();
        rec(k, st);
    }
    void stop(int k, hipStream_t st)
    {
        if ((mask >> k) & 1u) rec(k | 0x100, st);
    }
    // A pair of events for hipExtLaunchKernelGGL to fill -- the kernel's own start and end, as rocprofv3 sees them -- instead of two
    // hipEventRecord commands around the launch: those are queue packets of their own, and between them and a 70-us kernel of a
    // three-stream mix lay 40 us that were not the kernel (bench.py's roofline: 0.113 ms per launch against the trace's 0.071).
    bool pair(int k, hipEvent_t *a, hipEvent_t *b)
    {
        *a = *b = nullptr;
        if (!((mask >> k) & 1u)) return false;
        if (n + 4 >= MAXEV) flush();
        for (int j = 0; j < 2; ++j) {
            if (n == created) { (void)hipEventCreate(&ev[created]); ++created; }
            kind[n] = j ? (k | 0x100) : k;
            (j ? *b : *a) = ev[n++];
        }
        return true;
    }
    void flush()
    {
        if (n == 0) return;
        (void)hipEventSynchronize(ev[n - 1]);
        for (int i = 1; i < n; ++i) {
            if (!(kind[i] & 0x100) || kind[i - 1] != (kind[i] & 0xff)) continue;
            float t = 0.f;
            const int k = kind[i] & 0xff;
            if (hipEventElapsedTime(&t, ev[i - 1], ev[i]) == hipSuccess) { ms[k] += t; launches[k]++; }
        }
        n = 0;
    }
    void reset()
    {
        flush();
        for (int i = 0; i < MAXK; ++i) { ms[i] = 0; launches[i] = 0; }
    }
    ~KernelProfiler()
    {
        for (int i = 0; i < created; ++i) (void)hipEventDestroy(ev[i]);
    }
};

// Small host <-> device transfers of the per-frame host-array calls, done by the compute queue itself: pinned host memory is
// mapped into the device's address space, so a kernel can read the staged inputs from it (16 bytes per lane) and write the
// result block into it (posted writes).  A copy-engine transfer costs its own time (11 us for 250 KB) PLUS ~8 us of
// hand-over between the engine and the compute queue on either side of the kernels -- measured on a 0.1-ms call: 0.108 ->
// 0.092 ms with both directions moved to the queue.  Above STAGE_MAX bytes the copy engine's bandwidth wins and is used.
constexpr size_t STAGE_MAX = (size_t)2 << 20;
static __global__ __launch_bounds__(256) void k_stage_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
inline bool stage_ok(const void *a, const void *b, size_t bytes)
{
    return bytes <= STAGE_MAX && ((((uintptr_t)a) | ((uintptr_t)b) | bytes) & 15) == 0;
}
// pinned -> device / device -> pinned on `st`; both asynchronous, ordered with the kernels of the stream
inline hipError_t stage_in(void *dev, const void *pin, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    if (!stage_ok(dev, pin, bytes)) return hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, st, static_cast<const uint4 *>(pin), static_cast<uint4 *>(dev), bytes / 16);
    return hipGetLastError();
}
inline hipError_t stage_out(void *pin, const void *dev, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    if (!stage_ok(dev, pin, bytes)) return hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st);
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, st, static_cast<const uint4 *>(dev), static_cast<uint4 *>(pin), bytes / 16);
    return hipGetLastError();
}

// Inclusive scan of one int per lane over the wave: DPP row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes (lanes without a source add
// 0), then the last lane of row 0 (2) broadcast into row 1 (3) and lane 31 into rows 2 and 3: six register-to-register steps instead of
// six ds_bpermute round trips through LDS (__shfl_up), each of which the next step waits for.
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15, rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31, rows 2 and 3
    return v;
}

// Minimum over each row of 16 lanes, in every lane of the row: four DPP rotations (row_ror 8 / 4 / 2 / 1) instead of four
// ds_bpermute round trips (__shfl_xor).  wave_min_u32: the four row minima joined through scalars, the same value in all 64 lanes.
__device__ __forceinline__ unsigned row_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = row_min_u32(v);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

} // namespace orbx

#define ORBX_FAIL(code, msg) return orbx::set_error((code), (msg), __FILE__, __LINE__)

#define ORBX_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return orbx::set_error(ORBX_ERR_HIP, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ORBX_NEED_DEVICE()                                                              \
    do {                                                                                \
        if (!orbx::device_ok()) ORBX_FAIL(ORBX_ERR_NO_DEVICE, "no usable HIP device (no CPU fallback exists)"); \
    } while (0)

#endif // ORBX_COMMON_H
