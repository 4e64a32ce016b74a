// orbm_match.hip -- Hamming matching data plane of ORBmatcher for gfx950.
//
// Replaces the inner loops of ORB_SLAM2::ORBmatcher (src/ORBmatcher.cc):
// DescriptorDistance (:1848-1864) as XOR + v_bcnt popcount on 256-bit rows, and
// the best / second-best / arg-best selection every Search* function shares
// (strict '<': first-seen candidate wins ties; second updated with 'else if').
// This is bit-twiddling, not a contraction: no MFMA.  One query row per lane;
// the B-side tile is staged in LDS and read as wave-uniform (broadcast) 128-bit
// words, so LDS traffic is conflict-free and HBM sees each descriptor once per
// workgroup.
#include <limits.h>
#include <stdint.h>

#include "common.h"

namespace {

constexpr int MT = 256;   // threads per block = query rows per block
constexpr int TILE = 256; // B descriptors staged per step (8 KiB)

__device__ __forceinline__ int hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ void select_update(int dist, int j, int &best, int &second, int &idx)
{
    // if(dist<bestDist){bestDist2=bestDist;bestDist=dist;bestIdx=j;} else if(dist<bestDist2) bestDist2=dist;
    const bool lt = dist < best;
    second = lt ? best : (dist < second ? dist : second);
    idx = lt ? j : idx;
    best = lt ? dist : best;
}

// Pair p: queries = set qa[p], candidates = all of set qb[p] in index order.
__global__ __launch_bounds__(MT) void k_match_sets(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                   int cap, const int *__restrict__ qa, const int *__restrict__ qb,
                                                   int th, float nnratio, int *__restrict__ best_o,
                                                   int *__restrict__ second_o, int *__restrict__ idx_o,
                                                   int *__restrict__ match12, int *__restrict__ nmatch)
{
    __shared__ uint4 tile[TILE * 2];
    const int p = blockIdx.y, tid = threadIdx.x;
    const int sa = qa ? qa[p] : 0, sb = qb ? qb[p] : 1;
    const int nA = counts[sa] < cap ? counts[sa] : cap, nB = counts[sb] < cap ? counts[sb] : cap;
    if (blockIdx.x * MT >= nA) return;
    const int i = blockIdx.x * MT + tid;
    const uint4 *A = reinterpret_cast<const uint4 *>(desc + (size_t)sa * cap * 32);
    const uint4 *B = reinterpret_cast<const uint4 *>(desc + (size_t)sb * cap * 32);
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (i < nA) { a0 = A[2 * i]; a1 = A[2 * i + 1]; }
    int best = INT_MAX, second = INT_MAX, idx = -1;
    for (int j0 = 0; j0 < nB; j0 += TILE) {
        const int nt = nB - j0 < TILE ? nB - j0 : TILE;
        __syncthreads();
        for (int k = tid; k < nt * 2; k += MT) tile[k] = B[2 * j0 + k];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < nt; ++j) {
            const int dist = hamming256(a0, a1, tile[2 * j], tile[2 * j + 1]);
            select_update(dist, j0 + j, best, second, idx);
        }
    }
    if (i < nA) {
        const size_t o = (size_t)p * cap + i;
        if (best_o) best_o[o] = best;
        if (second_o) second_o[o] = second;
        if (idx_o) idx_o[o] = idx;
        // ORBmatcher.cc:674-676: bestDist<=TH && bestDist<(float)bestDist2*mfNNratio
        const bool ok = idx >= 0 && best <= th && (float)best < (float)second * nnratio;
        if (match12) match12[o] = ok ? idx : -1;
        if (nmatch) {
            const unsigned long long b = __ballot(ok);
            if ((tid & 63) == 0 && b) atomicAdd(&nmatch[p], __popcll(b));
        }
    }
}

// Gated variant: per-query candidate list (CSR), candidate order preserved.
__global__ __launch_bounds__(MT) void k_match_cands(const uint4 *__restrict__ A, int nA, const uint4 *__restrict__ B,
                                                    const int *__restrict__ off, const int *__restrict__ cidx,
                                                    int *__restrict__ best_o, int *__restrict__ second_o,
                                                    int *__restrict__ idx_o)
{
    const int i = blockIdx.x * MT + threadIdx.x;
    if (i >= nA) return;
    const uint4 a0 = A[2 * i], a1 = A[2 * i + 1];
    int best = INT_MAX, second = INT_MAX, idx = -1;
    for (int k = off[i]; k < off[i + 1]; ++k) {
        const int j = cidx[k];
        select_update(hamming256(a0, a1, B[2 * j], B[2 * j + 1]), j, best, second, idx);
    }
    best_o[i] = best; second_o[i] = second; idx_o[i] = idx;
}

__global__ __launch_bounds__(MT) void k_hamming_matrix(const uint4 *__restrict__ A, int nA, const uint4 *__restrict__ B,
                                                       int nB, unsigned short *__restrict__ out)
{
    const int j = blockIdx.x * MT + threadIdx.x, i = blockIdx.y;
    if (j >= nB) return;
    out[(size_t)i * nB + j] = (unsigned short)hamming256(A[2 * i], A[2 * i + 1], B[2 * j], B[2 * j + 1]);
}

orbx::KernelProfiler g_prof;

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? 0 : -1; }
};

} // namespace

extern "C" {

int orbm_match_batch_dev(const uint8_t *desc_dev, const int32_t *counts_dev, int cap, const int32_t *pair_a_dev,
                         const int32_t *pair_b_dev, int npairs, int th, float nnratio, int32_t *best_dev,
                         int32_t *second_dev, int32_t *idx_dev, int32_t *match12_dev, int32_t *nmatch_dev, void *stream)
{
    if (!desc_dev || !counts_dev || cap <= 0 || npairs <= 0) ORBX_FAIL(ORBX_ERR_ARG, "bad match arguments");
    ORBX_NEED_DEVICE();
    hipStream_t st = (hipStream_t)stream;
    if (nmatch_dev) ORBX_HIP(hipMemsetAsync(nmatch_dev, 0, sizeof(int) * npairs, st));
    g_prof.begin(st);
    hipLaunchKernelGGL(k_match_sets, dim3((cap + MT - 1) / MT, npairs), dim3(MT), 0, st, desc_dev, counts_dev, cap,
                       pair_a_dev, pair_b_dev, th, nnratio, best_dev, second_dev, idx_dev, match12_dev, nmatch_dev);
    g_prof.mark(0, st);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbm_match_bruteforce(const uint8_t *A, int nA, const uint8_t *B, int nB, int32_t *best, int32_t *second, int32_t *idx)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || !best || !second || !idx) ORBX_FAIL(ORBX_ERR_ARG, "bad match arguments");
    ORBX_NEED_DEVICE();
    if (nA == 0) return ORBX_OK;
    const int cap = nA > nB ? nA : nB;
    DevBuf d, c, o;
    if (d.alloc((size_t)2 * cap * 32) || c.alloc(2 * sizeof(int)) || o.alloc((size_t)3 * cap * sizeof(int)))
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    const int cnt[2] = {nA, nB};
    ORBX_HIP(hipMemcpy(d.p, A, (size_t)nA * 32, hipMemcpyHostToDevice));
    if (nB) ORBX_HIP(hipMemcpy((uint8_t *)d.p + (size_t)cap * 32, B, (size_t)nB * 32, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(c.p, cnt, sizeof(cnt), hipMemcpyHostToDevice));
    int *ob = (int *)o.p;
    hipLaunchKernelGGL(k_match_sets, dim3((cap + MT - 1) / MT, 1), dim3(MT), 0, 0, (const uint8_t *)d.p, (const int *)c.p,
                       cap, (const int *)nullptr, (const int *)nullptr, 0, 0.f, ob, ob + cap, ob + 2 * cap,
                       (int *)nullptr, (int *)nullptr);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(best, ob, sizeof(int) * nA, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(second, ob + cap, sizeof(int) * nA, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(idx, ob + 2 * cap, sizeof(int) * nA, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

int orbm_match_candidates(const uint8_t *A, int nA, const uint8_t *B, int nB, const int32_t *cand_off,
                          const int32_t *cand_idx, int32_t *best, int32_t *second, int32_t *idx)
{
    if (nA < 0 || nB < 0 || (nA && !A) || !cand_off || !best || !second || !idx) ORBX_FAIL(ORBX_ERR_ARG, "bad match arguments");
    ORBX_NEED_DEVICE();
    if (nA == 0) return ORBX_OK;
    const int nc = cand_off[nA];
    if (nc < 0 || (nc && (!cand_idx || !B))) ORBX_FAIL(ORBX_ERR_ARG, "bad candidate lists");
    for (int k = 0; k < nc; ++k)
        if (cand_idx[k] < 0 || cand_idx[k] >= nB) ORBX_FAIL(ORBX_ERR_ARG, "candidate index out of range");
    for (int i = 0; i < nA; ++i)
        if (cand_off[i] > cand_off[i + 1] || cand_off[i] < 0) ORBX_FAIL(ORBX_ERR_ARG, "candidate offsets not monotone");
    DevBuf da, db, doff, dci, o;
    if (da.alloc((size_t)nA * 32) || db.alloc((size_t)nB * 32) || doff.alloc(sizeof(int) * (nA + 1)) ||
        dci.alloc(sizeof(int) * nc) || o.alloc(sizeof(int) * 3 * (size_t)nA))
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    ORBX_HIP(hipMemcpy(da.p, A, (size_t)nA * 32, hipMemcpyHostToDevice));
    if (nB) ORBX_HIP(hipMemcpy(db.p, B, (size_t)nB * 32, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(doff.p, cand_off, sizeof(int) * (nA + 1), hipMemcpyHostToDevice));
    if (nc) ORBX_HIP(hipMemcpy(dci.p, cand_idx, sizeof(int) * nc, hipMemcpyHostToDevice));
    int *ob = (int *)o.p;
    hipLaunchKernelGGL(k_match_cands, dim3((nA + MT - 1) / MT), dim3(MT), 0, 0, (const uint4 *)da.p, nA,
                       (const uint4 *)db.p, (const int *)doff.p, (const int *)dci.p, ob, ob + nA, ob + 2 * nA);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(best, ob, sizeof(int) * nA, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(second, ob + nA, sizeof(int) * nA, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(idx, ob + 2 * nA, sizeof(int) * nA, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

int orbm_hamming_matrix(const uint8_t *A, int nA, const uint8_t *B, int nB, uint16_t *out)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || ((nA && nB) && !out)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (nA == 0 || nB == 0) return ORBX_OK;
    DevBuf da, db, o;
    if (da.alloc((size_t)nA * 32) || db.alloc((size_t)nB * 32) || o.alloc(sizeof(uint16_t) * (size_t)nA * nB))
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    ORBX_HIP(hipMemcpy(da.p, A, (size_t)nA * 32, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(db.p, B, (size_t)nB * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nB + MT - 1) / MT, nA), dim3(MT), 0, 0, (const uint4 *)da.p, nA,
                       (const uint4 *)db.p, nB, (unsigned short *)o.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(out, o.p, sizeof(uint16_t) * (size_t)nA * nB, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

int orbm_profile_enable(int on)
{
    g_prof.names[0] = "k_match_sets";
    g_prof.reset();
    g_prof.on = on != 0;
    return ORBX_OK;
}

int orbm_profile_read(double *total_ms, int64_t *launches)
{
    g_prof.flush();
    if (total_ms) *total_ms = g_prof.ms[0];
    if (launches) *launches = g_prof.launches[0];
    return ORBX_OK;
}

// Host-side acceptance filter; pure integer/float compares on caller arrays.
int orbm_match_filter(int nA, const int32_t *best, const int32_t *second, const int32_t *idx, int th, float nnratio,
                      int32_t *match12, int *nmatches)
{
    if (nA < 0 || (nA && (!best || !second || !idx || !match12))) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    int n = 0;
    for (int i = 0; i < nA; ++i) {
        const bool ok = idx[i] >= 0 && best[i] <= th && (float)best[i] < (float)second[i] * nnratio;
        match12[i] = ok ? idx[i] : -1;
        n += ok;
    }
    if (nmatches) *nmatches = n;
    return ORBX_OK;
}

} // extern "C"
