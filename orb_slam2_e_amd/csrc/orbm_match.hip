// orbm_match.hip -- Hamming matching data plane of ORBmatcher for gfx950.
//
// Replaces the inner loops of ORB_SLAM2::ORBmatcher (src/ORBmatcher.cc):
// DescriptorDistance (:1848-1864) as XOR + v_bcnt popcount on 256-bit rows, and
// the best / second-best / arg-best selection every Search* function shares
// (strict '<': first-seen candidate wins ties; second updated with 'else if').
// The gated searches are bit-twiddling on short candidate lists (no MFMA); the all-pairs
// matcher is a contraction over the descriptor bits and runs on the matrix cores (k_match_sets_mfma).
#include <limits.h>
#include <stdint.h>

#include <math.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "orbm_internal.h"
#include "orbx_internal.h"

using namespace orbm_detail;

namespace {

// v_bcnt_u32_b32 d, s0, s1 = popcount(s0) + s1: one accumulating chain, 8 xor + 8 bcnt
// per 256-bit pair (left to the compiler the sum becomes 8 bcnt + 3-4 v_add3).
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ int hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    unsigned d = bcnt_acc(a0.x ^ b0.x, 0u);
    d = bcnt_acc(a0.y ^ b0.y, d);
    d = bcnt_acc(a0.z ^ b0.z, d);
    d = bcnt_acc(a0.w ^ b0.w, d);
    d = bcnt_acc(a1.x ^ b1.x, d);
    d = bcnt_acc(a1.y ^ b1.y, d);
    d = bcnt_acc(a1.z ^ b1.z, d);
    d = bcnt_acc(a1.w ^ b1.w, d);
    return (int)d;
}

// Sequential form used by the gated variant (candidate lists are short).
__device__ __forceinline__ void select_update(int dist, int j, int &best, int &second, int &idx)
{
    // if(dist<bestDist){bestDist2=bestDist;bestDist=dist;bestIdx=j;} else if(dist<bestDist2) bestDist2=dist;
    const bool lt = dist < best;
    second = lt ? best : (dist < second ? dist : second);
    idx = lt ? j : idx;
    best = lt ? dist : best;
}

// Pair p: queries = set qa[p], candidates = all of set qb[p] in index order.
// Workgroup = 4 waves x 128 query rows: every lane keeps TWO query rows in VGPRs
// (rows i and i+64 of the block), every wave scans one quarter of the candidates.
// A wave stages its candidates 64 at a time in a private LDS buffer (one 32-byte
// row per lane, double buffered, no workgroup barrier) and reads them back as
// wave-uniform 128-bit broadcasts, each feeding two Hamming distances.  Per pair:
// 8 v_xor + 8 v_bcnt + key pack + min/max.  key = dist<<16 | j keeps "strict <,
// first index wins": the smallest key is the best match, the second smallest key
// carries the second-best distance (= what the if / else-if chain of
// ORBmatcher.cc:664-672 leaves in bestDist2).  The four partial (smallest, second
// smallest) pairs are merged through LDS at the end.
constexpr int MSEG = 4;   // waves per workgroup = candidate segments
constexpr int MQ = 2;     // query rows per lane

__device__ __forceinline__ void key_update(unsigned key, unsigned &k1, unsigned &k2)
{
    // k1 <= k2 always: median(k1, key, k2) = new second smallest (one v_med3_u32 instead of max + min)
    unsigned m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k1), "v"(key), "v"(k2));
    k2 = m;
    k1 = k1 < key ? k1 : key;
}

__global__ __launch_bounds__(64 * MSEG) void k_match_sets(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                          int cap, const int *__restrict__ qa, const int *__restrict__ qb,
                                                          int th, float nnratio, int *__restrict__ best_o,
                                                          int *__restrict__ second_o, int *__restrict__ idx_o,
                                                          int *__restrict__ match12, int *__restrict__ nmatch)
{
    __shared__ uint4 stage[MSEG][2][64 * 2];
    __shared__ unsigned sk[MSEG][MQ][2][64];
    const int p = blockIdx.y, lane = threadIdx.x, seg = threadIdx.y;
    const int sa = qa ? qa[p] : 0, sb = qb ? qb[p] : 1;
    const int nA = min(max(counts[sa], 0), cap), nB = min(max(counts[sb], 0), cap); // a count outside [0, cap] is a caller's bug; never index with it
    const int row0 = blockIdx.x * 64 * MQ;
    if (row0 >= nA) return;
    const uint4 *A = reinterpret_cast<const uint4 *>(desc + (size_t)sa * cap * 32);
    const uint4 *B = reinterpret_cast<const uint4 *>(desc + (size_t)sb * cap * 32);
    uint4 a[MQ][2];
    unsigned k1[MQ], k2[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const int i = row0 + q * 64 + lane;
        a[q][0] = a[q][1] = make_uint4(0, 0, 0, 0);
        if (i < nA) { a[q][0] = A[2 * i]; a[q][1] = A[2 * i + 1]; }
        k1[q] = k2[q] = 0xffffffffu;
    }
    const int per = (nB + MSEG - 1) / MSEG;
    const int j0 = seg * per, j1 = j0 + per < nB ? j0 + per : nB;
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
    if (j0 + lane < j1) { n0 = B[2 * (j0 + lane)]; n1 = B[2 * (j0 + lane) + 1]; }
    stage[seg][0][2 * lane] = n0;
    stage[seg][0][2 * lane + 1] = n1;
    int buf = 0;
    for (int c0 = j0; c0 < j1; c0 += 64, buf ^= 1) {
        const int nxt = c0 + 64 + lane;
        if (nxt < j1) { n0 = B[2 * nxt]; n1 = B[2 * nxt + 1]; } // in flight during the scan below
        const int nc = j1 - c0 < 64 ? j1 - c0 : 64;
        const uint4 *t = stage[seg][buf];
        int j = 0;
        for (; j + 4 <= nc; j += 4) { // manual x4: the asm-based popcount chain defeats #pragma unroll
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 b0 = t[2 * (j + u)], b1 = t[2 * (j + u) + 1];
#pragma unroll
                for (int q = 0; q < MQ; ++q)
                    key_update(((unsigned)hamming256(a[q][0], a[q][1], b0, b1) << 16) | (unsigned)(c0 + j + u), k1[q], k2[q]);
            }
        }
        for (; j < nc; ++j) {
            const uint4 b0 = t[2 * j], b1 = t[2 * j + 1];
#pragma unroll
            for (int q = 0; q < MQ; ++q)
                key_update(((unsigned)hamming256(a[q][0], a[q][1], b0, b1) << 16) | (unsigned)(c0 + j), k1[q], k2[q]);
        }
        stage[seg][buf ^ 1][2 * lane] = n0;
        stage[seg][buf ^ 1][2 * lane + 1] = n1;
    }
    // merge the per-segment (smallest, second smallest) pairs; keys are unique
#pragma unroll
    for (int q = 0; q < MQ; ++q) { sk[seg][q][0][lane] = k1[q]; sk[seg][q][1][lane] = k2[q]; }
    __syncthreads();
    if (seg != 0) return;
    int nok = 0;
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        unsigned m1 = k1[q], m2 = k2[q];
#pragma unroll
        for (int g = 1; g < MSEG; ++g) {
            const unsigned o1 = sk[g][q][0][lane], o2 = sk[g][q][1][lane];
            const unsigned hi = m1 > o1 ? m1 : o1, lo2 = m2 < o2 ? m2 : o2;
            m1 = m1 < o1 ? m1 : o1;
            m2 = hi < lo2 ? hi : lo2;
        }
        const int i = row0 + q * 64 + lane;
        bool ok = false;
        if (i < nA) {
            const int best = nB > 0 ? (int)(m1 >> 16) : INT_MAX, idx = nB > 0 ? (int)(m1 & 0xffffu) : -1;
            const int second = nB > 1 ? (int)(m2 >> 16) : INT_MAX;
            const size_t o = (size_t)p * cap + i;
            if (best_o) best_o[o] = best;
            if (second_o) second_o[o] = second;
            if (idx_o) idx_o[o] = idx;
            // ORBmatcher.cc:674-676: bestDist<=TH && bestDist<(float)bestDist2*mfNNratio
            ok = idx >= 0 && best <= th && (float)best < (float)second * nnratio;
            if (match12) match12[o] = ok ? idx : -1;
        }
        nok += __popcll(__ballot(ok));
    }
    if (nmatch && lane == 0 && nok) atomicAdd(&nmatch[p], nok);
}

// ---- The all-pairs matcher on the matrix cores ---------------------------------------------------------
// A 2000 x 2000 Hamming matrix is a contraction over the 256 descriptor bits: with train bit t sent to
// (1 - 2t) and query bit q to -(1 - 2q) / 2, sum_k a_k b_k = hamming - 128, and +-1 / +-0.5 are exact FP4
// (E2M1) values, so v_mfma_f32_32x32x64_f8f6f4 computes 32 x 32 distances from four K = 64 steps at the FP4
// rate (8 x the K of the bf16 form per cycle) with exact small-integer f32 sums.  The accumulator is seeded
// with 128 + r * 2^-15 (r = the row inside its 32-row tile), and the tile's first row * 2^-15 is added to what
// leaves the tile, so the values that compete ARE the selection keys dist + j / 32768 (j = train index): at
// most 9 + 15 significant bits, exact in f32, positive, and ordered like their bit patterns.  The K order inside
// a step is whatever the hardware uses: both operands are expanded by the same function of (lane half,
// step), and a dot product does not care.  Train rows sit on M (the 16 accumulator registers of a lane),
// queries on N (the lane), so a lane folds its registers into the running pair of its own query with no
// cross-lane traffic -- and it folds ONE key per MFMA result, the smallest of its 16 (five v_min3_u32 and
// three more): the pair (smallest key, second smallest GROUP minimum) differs from (smallest, second smallest)
// only by what the winner's own group hides, 15 keys that the finishing lane recomputes with popcounts at the
// very end (group_rest_min).  40 fold instructions per train tile and wave instead of 128.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int XQ = 4;                          // 32-query tiles per wave
// Waves per workgroup.  The kernel holds 3 waves per SIMD (156 VGPRs: the four query tiles' FP4 operands alone are 128), 12
// per CU: workgroups of 3 waves make that 4 workgroups per CU = 1024 slots, exactly the 16 query blocks x 64 pairs of the
// bench -- one full round.  With 4 waves per workgroup (3 per CU, 768 slots) the same grid took a full round and a third of one.
constexpr int XW = 3;
constexpr unsigned XKEY_INF = 0x7f000000u;     // above every key (keys < 257)
constexpr float XIDX = 1.0f / 32768.0f;
constexpr int XMAXN = 32768;                   // j * 2^-15 < 1
constexpr unsigned FP4_TRAIN = 0xAAA22A22u;    // byte v = two E2M1 codes for bits (v & 1, v >> 1): 0 -> +1 (0x2), 1 -> -1 (0xA)
constexpr unsigned FP4_QUERY = 0x11199199u;    // 0 -> -0.5 (0x9), 1 -> +0.5 (0x1)

// 32 descriptor bits -> 32 FP4 codes: the 2-bit fields of the four bytes select table bytes (v_perm_b32).
template <unsigned TBL> __device__ __forceinline__ v8i expand_fp4(unsigned x)
{
    v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    r[0] = (int)__builtin_amdgcn_perm(0u, TBL, x & 0x03030303u);
    r[1] = (int)__builtin_amdgcn_perm(0u, TBL, (x >> 2) & 0x03030303u);
    r[2] = (int)__builtin_amdgcn_perm(0u, TBL, (x >> 4) & 0x03030303u);
    r[3] = (int)__builtin_amdgcn_perm(0u, TBL, (x >> 6) & 0x03030303u);
    return r;
}

// key_update for keys that come straight out of an MFMA: no inline asm here -- the compiler's hazard recogniser
// does not see asm operands, and a VALU read of a fresh MFMA result needs software wait states.  The keys are
// positive finite floats, so v_med3_f32 orders them like v_med3_u32.
__device__ __forceinline__ void key_update_f(float key, unsigned &k1, unsigned &k2)
{
    k2 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(k1), key, __uint_as_float(k2)));
    const unsigned u = __float_as_uint(key);
    k1 = k1 < u ? k1 : u;
}

__device__ __forceinline__ void merge_pairs(unsigned &m1, unsigned &m2, unsigned o1, unsigned o2)
{
    const unsigned hi = m1 > o1 ? m1 : o1, lo2 = m2 < o2 ? m2 : o2;
    m1 = m1 < o1 ? m1 : o1;
    m2 = hi < lo2 ? hi : lo2;
}

// One 32-row train tile against the XQ query tiles of the wave: expand, 4 x XQ MFMAs, fold the keys.
__device__ __forceinline__ void mfma_tile(const uint4 &x, const v16f &c, float tilebase, const v8i (&bq)[XQ][4], unsigned (&k1)[XQ], unsigned (&k2)[XQ])
{
    const v8i a0 = expand_fp4<FP4_TRAIN>(x.x), a1 = expand_fp4<FP4_TRAIN>(x.y), a2 = expand_fp4<FP4_TRAIN>(x.z),
              a3 = expand_fp4<FP4_TRAIN>(x.w);
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bq[q][0], c, 4, 4, 0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bq[q][1], acc, 4, 4, 0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, bq[q][2], acc, 4, 4, 0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a3, bq[q][3], acc, 4, 4, 0, 0, 0, 0);
        // ONE key per MFMA result enters the running pair: the smallest of the lane's 16 (v_min3_u32 on the bit patterns: positive
        // floats order like integers) -- 10 instructions instead of 32.  The pair then holds the smallest key and the second smallest
        // GROUP minimum; what the groups hide is only the rest of the winner's own group, and that is looked at once, at the very end
        // (group_rest_min).
        unsigned u[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) u[g] = __float_as_uint(acc[g]);
        const unsigned t0 = min(min(u[0], u[1]), u[2]), t1 = min(min(u[3], u[4]), u[5]), t2 = min(min(u[6], u[7]), u[8]),
                       t3 = min(min(u[9], u[10]), u[11]), t4 = min(min(u[12], u[13]), u[14]);
        const unsigned m = min(min(min(t0, t1), t2), min(min(t3, t4), u[15]));
        key_update_f(__uint_as_float(m) + tilebase, k1[q], k2[q]);   // the tile's first row joins the index part here: once per result, not per accumulator
    }
}

// The keys of the other rows of the winner's group -- the 16 rows of tile idx / 32 that share its lane half: 8 a + 4 h + b, a, b = 0..3,
// h = bit 2 of the row -- against query descriptor (a0, a1): the smallest of them (XKEY_INF if there is none below nB).
__device__ __forceinline__ unsigned group_rest_min(const uint4 &a0, const uint4 &a1, const uint4 *__restrict__ B, int idx, int nB)
{
    const int base = (idx & ~31) + (idx & 4);
    unsigned m = XKEY_INF;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = base + 8 * a + b, jc = min(j, nB - 1);
            const int dist = popc256(a0, a1, B[2 * jc], B[2 * jc + 1]);
            const unsigned key = __float_as_uint((float)dist + (float)j * XIDX);
            m = (j != idx && j < nB && key < m) ? key : m;
        }
    return m;
}

// Workgroup = XW waves x (XQ * 32 = 128 queries); wave `seg` scans the train tiles seg, seg + XW, ... .
__global__ __launch_bounds__(64 * XW) void k_match_sets_mfma(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                          int cap, const int *__restrict__ qa, const int *__restrict__ qb,
                                                          int th, float nnratio, int *__restrict__ best_o,
                                                          int *__restrict__ second_o, int *__restrict__ idx_o,
                                                          int *__restrict__ match12, int *__restrict__ nmatch)
{
    __shared__ unsigned sk[XW][XQ][2][32];
    const int p = blockIdx.y, lane = threadIdx.x & 63, seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sa = qa ? qa[p] : 0, sb = qb ? qb[p] : 1;
    const int nA = min(max(counts[sa], 0), cap), nB = min(max(counts[sb], 0), cap); // a count outside [0, cap] is a caller's bug; never index with it
    const int row0 = blockIdx.x * 32 * XQ;
    if (row0 >= nA) return;
    const uint4 *A = reinterpret_cast<const uint4 *>(desc + (size_t)sa * cap * 32);
    const uint4 *B = reinterpret_cast<const uint4 *>(desc + (size_t)sb * cap * 32);
    const int r = lane & 31, h = lane >> 5;            // operand row / column, and which 128-bit half of it this lane feeds
    v8i bq[XQ][4];
    unsigned k1[XQ], k2[XQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        const int i = min(row0 + q * 32 + r, nA - 1);
        const uint4 x = A[2 * i + h];
        bq[q][0] = expand_fp4<FP4_QUERY>(x.x); bq[q][1] = expand_fp4<FP4_QUERY>(x.y);
        bq[q][2] = expand_fp4<FP4_QUERY>(x.z); bq[q][3] = expand_fp4<FP4_QUERY>(x.w);
        k1[q] = k2[q] = XKEY_INF;
    }
    // accumulator register g of this lane is train row (g & 3) + 8 * (g >> 2) + 4 * h of the tile
    v16f c;
#pragma unroll
    for (int g = 0; g < 16; ++g) c[g] = 128.0f + (float)((g & 3) + 8 * (g >> 2) + 4 * h) * XIDX;     // the row inside its tile
    // full tiles first (no row test anywhere in that loop), then the one ragged tile, which belongs to one wave
    const int nfull = nB >> 5;
    uint4 nx = make_uint4(0, 0, 0, 0);
    if (seg < nfull) nx = B[2 * (seg * 32 + r) + h];
    for (int t = seg; t < nfull; t += XW) {
        const uint4 x = nx;
        if (t + XW < nfull) nx = B[2 * ((t + XW) * 32 + r) + h];                // in flight during this tile
        mfma_tile(x, c, (float)(32 * t) * XIDX, bq, k1, k2);
    }
    if ((nB & 31) && (nfull % XW) == seg) {
        const uint4 x = B[2 * min(nfull * 32 + r, nB - 1) + h];
#pragma unroll
        for (int g = 0; g < 16; ++g)
            if (nfull * 32 + (g & 3) + 8 * (g >> 2) + 4 * h >= nB) c[g] = __uint_as_float(XKEY_INF);   // rows past the set never win
        mfma_tile(x, c, (float)(32 * nfull) * XIDX, bq, k1, k2);
    }
    // fold the two lane halves (rows 4h.. of every tile), then the four waves
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        merge_pairs(k1[q], k2[q], (unsigned)__shfl_xor((int)k1[q], 32), (unsigned)__shfl_xor((int)k2[q], 32));
        if (h == 0) { sk[seg][q][0][r] = k1[q]; sk[seg][q][1][r] = k2[q]; }
    }
    __syncthreads();
    // the first waves finish the workgroup's queries, 64 each: lane -> query tile 2 seg + h, row r
    bool ok = false;
    const int qt = 2 * seg + h, i = row0 + qt * 32 + r;
    if (qt < XQ && i < nA) {
        unsigned m1 = sk[0][qt][0][r], m2 = sk[0][qt][1][r];
#pragma unroll
        for (int g = 1; g < XW; ++g) merge_pairs(m1, m2, sk[g][qt][0][r], sk[g][qt][1][r]);
        const float f1 = __uint_as_float(m1);
        const int d1 = (int)f1;
        const int best = nB > 0 ? d1 : INT_MAX, idx = nB > 0 ? (int)((f1 - (float)d1) * 32768.0f) : -1;
        if (nB > 1) {   // m2 = the second smallest group minimum so far: the rest of the winner's group may hold something smaller
            const unsigned gm = group_rest_min(A[2 * i], A[2 * i + 1], B, idx, nB);
            m2 = gm < m2 ? gm : m2;
        }
        const float f2 = __uint_as_float(m2);
        const int second = nB > 1 ? (int)f2 : INT_MAX;
        const size_t o = (size_t)p * cap + i;
        if (best_o) best_o[o] = best;
        if (second_o) second_o[o] = second;
        if (idx_o) idx_o[o] = idx;
        ok = idx >= 0 && best <= th && (float)best < (float)second * nnratio;     // ORBmatcher.cc:674-676
        if (match12) match12[o] = ok ? idx : -1;
    }
    const int nok = __popcll(__ballot(ok));
    if (nmatch && lane == 0 && nok) atomicAdd(&nmatch[p], nok);
}

// The same search with the train tiles SHARED by a workgroup: every wave owns YQ query tiles and scans ALL train tiles; a tile is
// loaded and expanded to FP4 once per workgroup (wave w of a batch of YW tiles takes tile w) and set down in LDS in the operand layout
// -- lane L of the producer holds exactly what lane L of every consumer feeds its MFMAs -- so the 44-instruction expansion is paid once
// per YW waves, and a wave's running pairs are complete at the end (no merge across waves).  Two sets of YW slots: one barrier per batch.
constexpr int YQ = 2, YW = 4;
__global__ __launch_bounds__(64 * YW) void k_match_sets_mfma_shared(const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                                 int cap, const int *__restrict__ qa, const int *__restrict__ qb,
                                                                 int th, float nnratio, int *__restrict__ best_o,
                                                                 int *__restrict__ second_o, int *__restrict__ idx_o,
                                                                 int *__restrict__ match12, int *__restrict__ nmatch)
{
    __shared__ __align__(16) uint4 s_tile[2][YW][4][64];      // [set][slot][K step][lane]: 32 KB
    const int p = blockIdx.y, lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sa = qa ? qa[p] : 0, sb = qb ? qb[p] : 1;
    const int nA = min(max(counts[sa], 0), cap), nB = min(max(counts[sb], 0), cap);
    const int row0 = blockIdx.x * 32 * YQ * YW;
    if (row0 >= nA) return;
    const uint4 *A = reinterpret_cast<const uint4 *>(desc + (size_t)sa * cap * 32);
    const uint4 *B = reinterpret_cast<const uint4 *>(desc + (size_t)sb * cap * 32);
    const int r = lane & 31, h = lane >> 5;
    const int qrow0 = row0 + wv * 32 * YQ;                    // this wave's queries
    v8i bq[YQ][4];
    unsigned k1[YQ], k2[YQ];
#pragma unroll
    for (int q = 0; q < YQ; ++q) {
        const int i = min(qrow0 + q * 32 + r, nA - 1);
        const uint4 x = A[2 * i + h];
        bq[q][0] = expand_fp4<FP4_QUERY>(x.x); bq[q][1] = expand_fp4<FP4_QUERY>(x.y);
        bq[q][2] = expand_fp4<FP4_QUERY>(x.z); bq[q][3] = expand_fp4<FP4_QUERY>(x.w);
        k1[q] = k2[q] = XKEY_INF;
    }
    v16f c;
#pragma unroll
    for (int g = 0; g < 16; ++g) c[g] = 128.0f + (float)((g & 3) + 8 * (g >> 2) + 4 * h) * XIDX;
    const int ntiles = (nB + 31) >> 5, nfull = nB >> 5, nbatch = (ntiles + YW - 1) / YW;
    // one tile from LDS slot (set, sl) against the wave's query tiles, seed `cs` (rows past the set carry XKEY_INF in the last tile's)
    auto consume = [&](int set, int sl, int t, const v16f &cs) {
        v8i a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = s_tile[set][sl][k][lane];
            a[k] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
        const float tilebase = (float)(32 * t) * XIDX;
        v16f accs[YQ];      // the query tiles' chains interleaved: an MFMA's accumulator is two instructions old, not one
#pragma unroll
        for (int q = 0; q < YQ; ++q) accs[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], bq[q][0], cs, 4, 4, 0, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < YQ; ++q) accs[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], bq[q][k], accs[q], 4, 4, 0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const v16f acc = accs[q];
            unsigned u[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) u[g] = __float_as_uint(acc[g]);
            const unsigned t0 = min(min(u[0], u[1]), u[2]), t1 = min(min(u[3], u[4]), u[5]), t2 = min(min(u[6], u[7]), u[8]),
                           t3 = min(min(u[9], u[10]), u[11]), t4 = min(min(u[12], u[13]), u[14]);
            const unsigned m = min(min(min(t0, t1), t2), min(min(t3, t4), u[15]));
            key_update_f(__uint_as_float(m) + tilebase, k1[q], k2[q]);
        }
    };
    // this wave's tile of batch 0, requested now; of batch b + 1 while batch b is being worked on
    uint4 nx = B[2 * min(wv * 32 + r, max(nB - 1, 0)) + h];
    for (int b = 0; b < nbatch; ++b) {
        const int set = b & 1;
        {   // produce: tile YW b + wv (a tile past the last one is never consumed)
            const uint4 x = nx;
            const int tn = YW * (b + 1) + wv;
            if (b + 1 < nbatch) nx = B[2 * min(tn * 32 + r, nB - 1) + h];
            const v8i a0 = expand_fp4<FP4_TRAIN>(x.x), a1 = expand_fp4<FP4_TRAIN>(x.y), a2 = expand_fp4<FP4_TRAIN>(x.z), a3 = expand_fp4<FP4_TRAIN>(x.w);
            s_tile[set][wv][0][lane] = make_uint4((unsigned)a0[0], (unsigned)a0[1], (unsigned)a0[2], (unsigned)a0[3]);
            s_tile[set][wv][1][lane] = make_uint4((unsigned)a1[0], (unsigned)a1[1], (unsigned)a1[2], (unsigned)a1[3]);
            s_tile[set][wv][2][lane] = make_uint4((unsigned)a2[0], (unsigned)a2[1], (unsigned)a2[2], (unsigned)a2[3]);
            s_tile[set][wv][3][lane] = make_uint4((unsigned)a3[0], (unsigned)a3[1], (unsigned)a3[2], (unsigned)a3[3]);
        }
        __syncthreads();                    // the batch's tiles are down; the other set is free again (its readers passed the last barrier)
#pragma unroll
        for (int sl = 0; sl < YW; ++sl) {
            const int t = YW * b + sl;
            if (t >= nfull) break;          // (the ragged tile, if any, is the last one: after the loop)
            consume(set, sl, t, c);
        }
    }
    if (nB & 31) {                          // the ragged tile sits in the last batch's set
        v16f cr;
#pragma unroll
        for (int g = 0; g < 16; ++g) cr[g] = nfull * 32 + (g & 3) + 8 * (g >> 2) + 4 * h >= nB ? __uint_as_float(XKEY_INF) : c[g];
        consume((nbatch - 1) & 1, nfull - YW * (nbatch - 1), nfull, cr);
    }
    // the two lane halves saw rows 4h.. of every tile: merge them; then lane (h, r) finishes query tile h, row r
#pragma unroll
    for (int q = 0; q < YQ; ++q)
        merge_pairs(k1[q], k2[q], (unsigned)__shfl_xor((int)k1[q], 32), (unsigned)__shfl_xor((int)k2[q], 32));
    static_assert(YQ == 2, "the finishing step maps a lane half to a query tile");
    unsigned m1 = h ? k1[1] : k1[0], m2 = h ? k2[1] : k2[0];
    bool ok = false;
    const int i = qrow0 + h * 32 + r;
    if (i < nA) {
        const float f1 = __uint_as_float(m1);
        const int d1 = (int)f1;
        const int best = nB > 0 ? d1 : INT_MAX, idx = nB > 0 ? (int)((f1 - (float)d1) * 32768.0f) : -1;
        if (nB > 1) {
            const unsigned gm = group_rest_min(A[2 * i], A[2 * i + 1], B, idx, nB);
            m2 = gm < m2 ? gm : m2;
        }
        const float f2 = __uint_as_float(m2);
        const int second = nB > 1 ? (int)f2 : INT_MAX;
        const size_t o = (size_t)p * cap + i;
        if (best_o) best_o[o] = best;
        if (second_o) second_o[o] = second;
        if (idx_o) idx_o[o] = idx;
        ok = idx >= 0 && best <= th && (float)best < (float)second * nnratio;     // ORBmatcher.cc:674-676
        if (match12) match12[o] = ok ? idx : -1;
    }
    const int nok = __popcll(__ballot(ok));
    if (nmatch && lane == 0 && nok) atomicAdd(&nmatch[p], nok);
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

// SearchForTriangulation inner loop (ORBmatcher.cc:892-990) + CheckDistEpipolarLine
// (:341-358): every query over its BoW-node candidate list (16 lanes per query, see the kernel).
// `dist>bestDist` is non-strict in the reference, so a later candidate with an equal
// distance replaces the earlier one; vbMatched2 is never set there, so queries are
// independent.
struct TriParams { float F12[9]; float ex, ey; int only_stereo; };
// 16 lanes per query, four queries per wave: the candidates of a query (its BoW node's members in frame 2, ~30) are taken 16 at a
// time.  The reference keeps the LAST candidate among those of smallest distance that pass the gates (`dist > bestDist` is
// non-strict, and the gates do not depend on bestDist), i.e. the minimum of dist << 16 | (0xffff - position in the list).
// (One lane per query walking its list alone was a 68-us chain of dependent loads for 2,000 queries on 32 waves.)
__global__ __launch_bounds__(MT) void k_match_triang(const orbx_keypoint *__restrict__ kps1, const uint4 *__restrict__ A, int nA,
                                                     const orbx_keypoint *__restrict__ kps2, const uint4 *__restrict__ B,
                                                     const int *__restrict__ off, const int *__restrict__ cidx,
                                                     const uint8_t *__restrict__ hasmp1, const uint8_t *__restrict__ hasmp2,
                                                     const uint8_t *__restrict__ stereo1, const uint8_t *__restrict__ stereo2,
                                                     TriParams tp, const float *__restrict__ scale2, const float *__restrict__ sigma2,
                                                     int *__restrict__ match12, int *__restrict__ bestdist)
{
    const int i = (blockIdx.x * MT + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (i >= nA) return;                       // (whole 16-lane groups leave together)
    unsigned key = 0xffffffffu;
    int k0 = 0;
    if (!hasmp1[i] && !(tp.only_stereo && !stereo1[i])) {
        const uint4 a0 = A[2 * i], a1 = A[2 * i + 1];
        const orbx_keypoint kp1 = kps1[i];
        // epipolar line in the second image l = x1' F12 = [a b c]
        const float a = kp1.x * tp.F12[0] + kp1.y * tp.F12[3] + tp.F12[6];
        const float b = kp1.x * tp.F12[1] + kp1.y * tp.F12[4] + tp.F12[7];
        const float c = kp1.x * tp.F12[2] + kp1.y * tp.F12[5] + tp.F12[8];
        const float den = a * a + b * b;
        const bool st1 = stereo1[i] != 0;
        k0 = off[i];
        const int k1 = off[i + 1];
        for (int k = k0 + sub; k < k1; k += 16) {
            const int j = cidx[k];
            if (hasmp2[j]) continue;
            const bool st2 = stereo2[j] != 0;
            if (tp.only_stereo && !st2) continue;
            const int dist = popc256(a0, a1, B[2 * j], B[2 * j + 1]);
            if (dist > 45) continue;           // TH_LOW
            const orbx_keypoint kp2 = kps2[j];
            if (!st1 && !st2) {
                const float distex = tp.ex - kp2.x, distey = tp.ey - kp2.y;
                if (distex * distex + distey * distey < 100 * scale2[kp2.octave]) continue;
            }
            const float num = a * kp2.x + b * kp2.y + c;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if ((double)dsqr < 3.84 * (double)sigma2[kp2.octave]) {
                const unsigned k2 = ((unsigned)dist << 16) | (0xffffu - (unsigned)min(k - k0, 0xffff));
                key = k2 < key ? k2 : key;
            }
        }
    }
    key = orbx::row_min_u32(key);
    if (sub == 0) {
        const bool hit = key != 0xffffffffu;
        match12[i] = hit ? cidx[k0 + (int)(0xffffu - (key & 0xffffu))] : -1;
        bestdist[i] = hit ? (int)(key >> 16) : 45;
    }
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:305-370), one wave per map
// point: N x N Hamming matrix in LDS, per-row median (element int(0.5*(N-1)) of the
// sorted row) by bisection on the value, arg-min with first-wins through a
// (median<<16 | row) key.
constexpr int DD_MAXN = 128;
__global__ __launch_bounds__(64) void k_distinctive(const uint4 *__restrict__ desc, const int *__restrict__ off,
                                                    int *__restrict__ best)
{
    __shared__ uint4 s_d[DD_MAXN * 2];
    __shared__ unsigned short s_m[DD_MAXN * DD_MAXN];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int o = off[i], N = off[i + 1] - o;
    if (N <= 0) { if (lane == 0) best[i] = -1; return; }
    if (N > DD_MAXN) {
        // A map point with more observations than the LDS matrix holds (long sessions: hundreds of keyframes see one point): row by
        // row, the distances of row a from the descriptors in L2 (lane = column, 64 at a time) into a 257-bin histogram, its
        // median = the first bin at which the running count reaches kth + 1.  N^2 / 64 wave iterations: rare, and bounded.
        int *hist = reinterpret_cast<int *>(s_m);
        const int kth = (int)(0.5 * (N - 1));
        unsigned key = 0xffffffffu;
        for (int a = 0; a < N; ++a) {
            for (int k = lane; k < 320; k += 64) hist[k] = 0;
            __syncthreads();
            const uint4 a0 = desc[2 * (size_t)(o + a)], a1 = desc[2 * (size_t)(o + a) + 1];
            for (int b = lane; b < N; b += 64) {
                const int d = a == b ? 0 : popc256(a0, a1, desc[2 * (size_t)(o + b)], desc[2 * (size_t)(o + b) + 1]);
                atomicAdd(&hist[d], 1);
            }
            __syncthreads();
            int c[5], run = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) { c[k] = hist[5 * lane + k]; run += c[k]; }
            const int incl = orbx::wave_incl_scan(run);     // inclusive scan over the lanes
            const unsigned long long reach = __ballot(incl >= kth + 1);
            const int owner = __ffsll((long long)reach) - 1;     // (the total is N >= kth + 1: some lane reaches it)
            int med = 0;
            if (lane == owner) {
                int cum = incl - run;
#pragma unroll
                for (int k = 0; k < 5; ++k) { cum += c[k]; if (cum >= kth + 1) { med = 5 * lane + k; break; } }
            }
            med = __shfl(med, owner);
            const unsigned k2 = ((unsigned)med << 16) | (unsigned)a;
            key = k2 < key ? k2 : key;
            __syncthreads();
        }
        if (lane == 0) best[i] = (int)(key & 0xffffu);
        return;
    }
    for (int k = lane; k < 2 * N; k += 64) s_d[k] = desc[2 * (size_t)o + k];
    __syncthreads();
    for (int p = lane; p < N * N; p += 64) {
        const int a = p / N, b = p - a * N;
        s_m[p] = a == b ? 0 : (unsigned short)popc256(s_d[2 * a], s_d[2 * a + 1], s_d[2 * b], s_d[2 * b + 1]);
    }
    __syncthreads();
    const int kth = (int)(0.5 * (N - 1));
    unsigned key = 0xffffffffu;
    for (int a = lane; a < N; a += 64) {
        int lo = 0, hi = 256; // smallest v with #(row <= v) >= kth+1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int b = 0; b < N; ++b) c += s_m[a * N + b] <= mid;
            if (c >= kth + 1) hi = mid; else lo = mid + 1;
        }
        const unsigned k2 = ((unsigned)lo << 16) | (unsigned)a;
        key = k2 < key ? k2 : key;
    }
    key = orbx::wave_min_u32(key);
    if (lane == 0) best[i] = (int)(key & 0xffffu);
}

// DBoW2 vocabulary-tree descent (TemplatedVocabulary.h:1218-1262): at every level the Hamming distance to all children of
// the current node, strict '<' so the first child wins ties; records the node at level L - levelsup (the FeatureVector key)
// and the leaf's word id.
// The tree is stored by CHILD SLOT, not by node: slot s = the s-th entry of the reference's concatenated children lists, so the
// children of any node are one contiguous run [c0, c1) of 32-byte descriptors (slot_desc) and of 16-byte records
// {c0, c1 of the child's own children, node id, word id} (slot_rec).  A level is then ONE memory round trip: every lane asks for
// its child's descriptor and record together, the row minimum picks the child, the winner's record comes from its lane.  (Indexed
// by node -- child_off[id] -> child_ids[k] -> node_desc[id], three dependent gathers per level -- the 35-MB tree of ORBvoc
// (k = 10, L = 6) cost 18 round trips per descriptor instead of 6.)
__global__ __launch_bounds__(MT) void k_bow_transform(const uint4 *__restrict__ slot_desc, const int4 *__restrict__ slot_rec, int root_c1,
                                                      int nid_level, const uint4 *__restrict__ feat, const int *__restrict__ counts,
                                                      int cap, int n_single, int *__restrict__ word_id, int *__restrict__ node_id)
{
    // 16 lanes (one DPP row) per feature: the children of the current node are scored 16 at a time, one per lane, and the row
    // minimum of dist << 16 | position picks the first closest child (strict d < best_d, :1246-1254)
    const int set = blockIdx.y, i = blockIdx.x * (MT / 16) + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int n = counts ? min(max(counts[set], 0), cap) : n_single;
    if (i >= n) return;
    const size_t o = (size_t)set * cap + i;
    const uint4 a0 = feat[2 * o], a1 = feat[2 * o + 1];
    const int row_base = (int)(threadIdx.x & 63u & ~15u);
    int c0 = 0, c1 = root_c1, level = 0, nid = 0, word = -1;
    do {
        ++level;
        unsigned key = 0xffffffffu;
        int4 mine = make_int4(0, 0, 0, -1);
        for (int kb = c0; kb < c1; kb += 16) {
            const int k = kb + sub;
            if (k < c1) {
                const int4 r = slot_rec[k];
                const unsigned d = (unsigned)popc256(a0, a1, slot_desc[2 * (size_t)k], slot_desc[2 * (size_t)k + 1]);
                const unsigned key2 = (d << 16) | (unsigned)(k - c0);
                if (key2 < key) { key = key2; mine = r; }
            }
        }
        // the winner's position is kb - c0 + sub with kb - c0 a multiple of 16: its lane of the row is position & 15
        const int src = (row_base + (int)(orbx::row_min_u32(key) & 15u)) << 2;
        c0 = __builtin_amdgcn_ds_bpermute(src, mine.x);
        c1 = __builtin_amdgcn_ds_bpermute(src, mine.y);
        const int id = __builtin_amdgcn_ds_bpermute(src, mine.z);
        word = __builtin_amdgcn_ds_bpermute(src, mine.w);
        if (level == nid_level) nid = id;
    } while (c1 > c0);
    if (sub == 0) {
        word_id[o] = word;
        node_id[o] = nid;
    }
}

__global__ __launch_bounds__(MT) void k_hamming_matrix(const uint4 *__restrict__ A, int nA, const uint4 *__restrict__ B,
                                                       int nB, unsigned short *__restrict__ out)
{
    const int j = blockIdx.x * MT + threadIdx.x, i = blockIdx.y;
    if (j >= nB) return;
    out[(size_t)i * nB + j] = (unsigned short)hamming256(A[2 * i], A[2 * i + 1], B[2 * j], B[2 * j + 1]);
}

orbx::KernelProfiler g_prof;

} // namespace

namespace orbm_detail {

static std::mutex g_ws_mutex;
static std::vector<Workspace *> g_ws_free;

int Workspace::reserve(size_t dev_bytes, size_t pin_bytes)
{
    used = 0;
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return -1; }
    if (dev_bytes > dev_cap) {
        if (dev) (void)hipFree(dev);
        dev = nullptr; dev_cap = 0;
        const size_t want = std::max(dev_bytes + dev_bytes / 2, (size_t)1 << 20);
        if (hipMalloc((void **)&dev, want) != hipSuccess) { dev = nullptr; return -1; }
        if (hipMemsetAsync(dev, 0, want, st) != hipSuccess) return -1;    // flags are generation numbers >= 1: stale memory must not look like one
        dev_cap = want;
    }
    if (pin_bytes > pin_cap) {
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; pin_cap = 0;
        const size_t want = std::max(pin_bytes + pin_bytes / 2, (size_t)1 << 20);
        if (hipHostMalloc((void **)&pin, want, hipHostMallocDefault) != hipSuccess) { pin = nullptr; return -1; }
        pin_cap = want;
    }
    return 0;
}

int Workspace::reserve_entries(size_t n)
{
    if (n <= ent_cap) return 0;
    if (ent) (void)hipFree(ent);
    ent = nullptr; ent_cap = 0;
    const size_t want = std::max(n + n / 2, (size_t)1 << 18);
    if (hipMalloc((void **)&ent, want * sizeof(unsigned)) != hipSuccess) { ent = nullptr; return -1; }
    ent_cap = want;
    return 0;
}

Workspace *workspace_acquire()
{
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    if (!g_ws_free.empty()) { Workspace *w = g_ws_free.back(); g_ws_free.pop_back(); return w; }
    return new Workspace();
}

void workspace_release(Workspace *w)
{
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    g_ws_free.push_back(w);
}

} // namespace orbm_detail

static std::atomic<int> g_allpairs_kind{ORBM_ALLPAIRS_AUTO};

// All-pairs launch: matrix-core kernel while the train index fits the key's 15 fraction bits (unless the caller
// asked for the popcount kernel, orbm_set_allpairs_kernel).
static void launch_match_sets(hipStream_t st, const uint8_t *desc, const int *counts, int cap, const int *qa, const int *qb, int npairs,
                              int th, float nnratio, int *best, int *second, int *idx, int *match12, int *nmatch)
{
    // Two matrix-core kernels, bit-identical results.  AUTO: train tiles shared by a workgroup through LDS (fewer vector instructions:
    // 7.5 M against 9.4 M per 64-pair launch; 39 us alone against 37, but the pipelined step is 0.9 % faster with it and a 20-step
    // region 3 %: what a kernel issues is what the other contexts' kernels wait behind).  ORBM_ALLPAIRS_MFMA: the kernel that splits
    // the train tiles over a workgroup's waves.
    const int kind = g_allpairs_kind.load(std::memory_order_relaxed);
    if (cap <= XMAXN && kind == ORBM_ALLPAIRS_AUTO)
        hipLaunchKernelGGL(k_match_sets_mfma_shared, dim3((cap + 32 * YQ * YW - 1) / (32 * YQ * YW), npairs), dim3(64 * YW), 0, st, desc, counts, cap,
                           qa, qb, th, nnratio, best, second, idx, match12, nmatch);
    else if (cap <= XMAXN && kind == ORBM_ALLPAIRS_MFMA)
        hipLaunchKernelGGL(k_match_sets_mfma, dim3((cap + 32 * XQ - 1) / (32 * XQ), npairs), dim3(64 * XW), 0, st, desc, counts, cap, qa, qb,
                           th, nnratio, best, second, idx, match12, nmatch);
    else
        hipLaunchKernelGGL(k_match_sets, dim3((cap + 64 * MQ - 1) / (64 * MQ), npairs), dim3(64, MSEG), 0, st, desc, counts, cap, qa, qb,
                           th, nnratio, best, second, idx, match12, nmatch);
}

extern "C" {

int orbm_match_batch_dev(const uint8_t *desc_dev, const int32_t *counts_dev, int cap, const int32_t *pair_a_dev,
                         const int32_t *pair_b_dev, int npairs, int th, float nnratio, int32_t *best_dev,
                         int32_t *second_dev, int32_t *idx_dev, int32_t *match12_dev, int32_t *nmatch_dev, void *stream)
{
    if (!desc_dev || !counts_dev || cap <= 0 || npairs <= 0) ORBX_FAIL(ORBX_ERR_ARG, "bad match arguments");
    ORBX_NEED_DEVICE();
    hipStream_t st = (hipStream_t)stream;
    // sets that are an extractor's resident results, produced on another stream (both calls handed NULL: the extractor then
    // works on its handle's stream, this call on the default stream): ordered both ways by events; nothing on a common stream
    orbx_extractor *producer = orbx_detail::order_after_producer(desc_dev, st);
    if (nmatch_dev) ORBX_HIP(hipMemsetAsync(nmatch_dev, 0, sizeof(int) * npairs, st));
    g_prof.start(0, st);
    launch_match_sets(st, desc_dev, counts_dev, cap, pair_a_dev, pair_b_dev, npairs, th, nnratio, best_dev, second_dev, idx_dev,
                      match12_dev, nmatch_dev);
    g_prof.stop(0, st);
    orbx_detail::reader_done(producer, st);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbm_match_bruteforce(const uint8_t *A, int nA, const uint8_t *B, int nB, int32_t *best, int32_t *second, int32_t *idx)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || !best || !second || !idx) ORBX_FAIL(ORBX_ERR_ARG, "bad match arguments");
    ORBX_NEED_DEVICE();
    if (nA == 0) return ORBX_OK;
    const int cap = nA > nB ? nA : nB;
    const int cnt[2] = {nA, nB};
    StagedCall sc;
    const size_t o_a = sc.in(nullptr, (size_t)2 * cap * 32); // two sets of `cap` rows
    sc.in_at(o_a, A, (size_t)nA * 32);
    sc.in_at(o_a + (size_t)cap * 32, B, (size_t)nB * 32);
    const size_t o_c = sc.in(cnt, sizeof(cnt));
    const size_t o_o = sc.out(sizeof(int) * 3 * (size_t)cap);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    int *ob = sc.d<int>(o_o);
    launch_match_sets(sc.stream(), sc.d<const uint8_t>(o_a), sc.d<const int>(o_c), cap, nullptr, nullptr, 1, 0, 0.f, ob, ob + cap,
                      ob + 2 * cap, nullptr, nullptr);
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    const int *r = sc.r<int>(o_o);
    memcpy(best, r, sizeof(int) * nA); memcpy(second, r + cap, sizeof(int) * nA); memcpy(idx, r + 2 * cap, sizeof(int) * nA);
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
    StagedCall sc;
    const size_t o_a = sc.in(A, (size_t)nA * 32), o_b = sc.in(B, (size_t)nB * 32), o_off = sc.in(cand_off, sizeof(int) * (nA + 1)),
                 o_ci = sc.in(cand_idx, sizeof(int) * (size_t)nc), o_o = sc.out(sizeof(int) * 3 * (size_t)nA);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    int *ob = sc.d<int>(o_o);
    hipLaunchKernelGGL(k_match_cands, dim3((nA + MT - 1) / MT), dim3(MT), 0, sc.stream(), sc.d<const uint4>(o_a), nA,
                       sc.d<const uint4>(o_b), sc.d<const int>(o_off), sc.d<const int>(o_ci), ob, ob + nA, ob + 2 * nA);
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    const int *r = sc.r<int>(o_o);
    memcpy(best, r, sizeof(int) * nA); memcpy(second, r + nA, sizeof(int) * nA); memcpy(idx, r + 2 * nA, sizeof(int) * nA);
    return ORBX_OK;
}

int orbm_distinctive_descriptors(const uint8_t *desc, const int32_t *off, int m, int32_t *best)
{
    if (m < 0 || (m && (!off || !best))) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (m == 0) return ORBX_OK;
    const int total = off[m];
    if (off[0] != 0 || total < 0 || (total && !desc)) ORBX_FAIL(ORBX_ERR_ARG, "bad offsets");
    for (int i = 0; i < m; ++i) {
        if (off[i] > off[i + 1]) ORBX_FAIL(ORBX_ERR_ARG, "offsets not monotone");
        if (off[i + 1] - off[i] > 65535) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "more than 65,535 observations of one map point");
    }
    StagedCall sc;
    const size_t o_d = sc.in(desc, (size_t)32 * total), o_off = sc.in(off, sizeof(int) * (m + 1)), o_o = sc.out(sizeof(int) * m);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    hipLaunchKernelGGL(k_distinctive, dim3(m), dim3(64), 0, sc.stream(), sc.d<const uint4>(o_d), sc.d<const int>(o_off), sc.d<int>(o_o));
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(best, sc.r<int>(o_o), sizeof(int) * m);
    return ORBX_OK;
}

struct orbm_vocabulary {
    int nnodes = 0, L = 0;
    int root_c1 = 0;            // the root's children are slots [0, root_c1)
    uint8_t *d_slot_desc = nullptr;   // [nnodes - 1][32]: descriptor of the node in child slot s (k_bow_transform)
    int4 *d_slot_rec = nullptr;       // [nnodes - 1]: {c0, c1 (that node's own children), node id, word id}
    std::vector<double> weight; // per word id (host: the BowVector is assembled by the caller)
    std::vector<int> word_of_node;
};

int orbm_vocab_create(const int32_t *child_off, const int32_t *child_ids, const uint8_t *node_desc, const int32_t *node_word,
                      const double *node_weight, int nnodes, int L, orbm_vocabulary **out)
{
    if (!child_off || !child_ids || !node_desc || !node_word || !node_weight || nnodes < 2 || L < 1 || !out)
        ORBX_FAIL(ORBX_ERR_ARG, "bad vocabulary");
    const int nch = child_off[nnodes];
    if (child_off[0] != 0 || nch != nnodes - 1) ORBX_FAIL(ORBX_ERR_ARG, "vocabulary is not a tree rooted at node 0");
    for (int i = 0; i < nnodes; ++i)
        if (child_off[i] > child_off[i + 1]) ORBX_FAIL(ORBX_ERR_ARG, "child offsets not monotone");
    for (int k = 0; k < nch; ++k)
        if (child_ids[k] <= 0 || child_ids[k] >= nnodes) ORBX_FAIL(ORBX_ERR_ARG, "child id out of range");
    if (child_off[1] == 0) ORBX_FAIL(ORBX_ERR_ARG, "root has no children");
    // depth check: every descent must end (children have larger depth); guards the device loop
    std::vector<int> depth(nnodes, -1);
    depth[0] = 0;
    std::vector<int> stack(1, 0);
    int visited = 0;
    while (!stack.empty()) {
        const int u = stack.back(); stack.pop_back(); ++visited;
        for (int k = child_off[u]; k < child_off[u + 1]; ++k) {
            if (depth[child_ids[k]] >= 0) ORBX_FAIL(ORBX_ERR_ARG, "node has two parents");
            depth[child_ids[k]] = depth[u] + 1;
            stack.push_back(child_ids[k]);
        }
    }
    if (visited != nnodes) ORBX_FAIL(ORBX_ERR_ARG, "unreachable nodes in the vocabulary");
    ORBX_NEED_DEVICE();
    orbm_vocabulary *v = new orbm_vocabulary();
    v->nnodes = nnodes; v->L = L;
    v->root_c1 = child_off[1];
    std::vector<uint8_t> sdesc((size_t)32 * nch);
    std::vector<int4> srec((size_t)nch);
    for (int k = 0; k < nch; ++k) {     // slot k holds node child_ids[k]
        const int id = child_ids[k];
        memcpy(&sdesc[(size_t)32 * k], node_desc + (size_t)32 * id, 32);
        srec[k] = make_int4(child_off[id], child_off[id + 1], id, node_word[id]);
    }
    if (hipMalloc((void **)&v->d_slot_desc, (size_t)32 * nch) != hipSuccess || hipMalloc((void **)&v->d_slot_rec, sizeof(int4) * (size_t)nch) != hipSuccess) {
        (void)orbm_vocab_destroy(v);
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    }
    {   // on a leased workspace's stream, waited for there (35 + 18 MB for ORBvoc: pageable source, staged by the runtime) -- never
        // the legacy stream, whose blocking copies wait for every other thread's work and fail while any thread captures a graph
        orbm_detail::WorkspaceLease lease;
        const hipStream_t st = lease.w->own_stream();
        hipError_t e = st ? hipMemcpyAsync(v->d_slot_desc, sdesc.data(), sdesc.size(), hipMemcpyHostToDevice, st) : hipErrorInvalidResourceHandle;
        if (e == hipSuccess) e = hipMemcpyAsync(v->d_slot_rec, srec.data(), sizeof(int4) * srec.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)orbm_vocab_destroy(v); ORBX_FAIL(ORBX_ERR_HIP, hipGetErrorString(e)); }
    }
    v->word_of_node.assign(node_word, node_word + nnodes);
    int nwords = 0;
    for (int i = 0; i < nnodes; ++i) nwords = std::max(nwords, node_word[i] + 1);
    v->weight.assign(nwords, 0.0);
    for (int i = 0; i < nnodes; ++i)
        if (node_word[i] >= 0) v->weight[node_word[i]] = node_weight[i];
    *out = v;
    return ORBX_OK;
}

int orbm_vocab_destroy(orbm_vocabulary *v)
{
    if (!v) return ORBX_OK;
    if (v->d_slot_desc) (void)hipFree(v->d_slot_desc);
    if (v->d_slot_rec) (void)hipFree(v->d_slot_rec);
    delete v;
    return ORBX_OK;
}

int orbm_bow_transform(orbm_vocabulary *v, const uint8_t *features, int n, int levelsup, int32_t *word_id, int32_t *node_id,
                       double *weight)
{
    if (!v || n < 0 || (n && (!features || !word_id || !node_id))) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (n == 0) return ORBX_OK;
    StagedCall sc;
    const size_t o_f = sc.in(features, (size_t)32 * n), o_o = sc.out(sizeof(int) * 2 * (size_t)n);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    int *ob = sc.d<int>(o_o);
    hipLaunchKernelGGL(k_bow_transform, dim3((n + MT / 16 - 1) / (MT / 16), 1), dim3(MT), 0, sc.stream(), (const uint4 *)v->d_slot_desc,
                       (const int4 *)v->d_slot_rec, v->root_c1, v->L - levelsup, sc.d<const uint4>(o_f), (const int *)nullptr, n, n, ob, ob + n);
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(word_id, sc.r<int>(o_o), sizeof(int) * n);
    memcpy(node_id, sc.r<int>(o_o) + n, sizeof(int) * n);
    if (weight)
        for (int i = 0; i < n; ++i) weight[i] = word_id[i] >= 0 ? v->weight[word_id[i]] : 0.0;
    if (v->L - levelsup <= 0)
        for (int i = 0; i < n; ++i) node_id[i] = 0; // root (TemplatedVocabulary.h:1230)
    return ORBX_OK;
}

int orbm_bow_transform_batch_dev(orbm_vocabulary *v, const uint8_t *desc_dev, const int32_t *counts_dev, int cap, int nsets,
                                 int levelsup, int32_t *word_id_dev, int32_t *node_id_dev, void *stream)
{
    if (!v || !desc_dev || !counts_dev || cap <= 0 || nsets <= 0 || !word_id_dev || !node_id_dev) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    hipEvent_t e0, e1;
    g_prof.pair(1, &e0, &e1);                      // (orbm_profile_enable(2): the kernel's own start and end, as a kernel trace sees them)
    hipExtLaunchKernelGGL(k_bow_transform, dim3((cap + MT / 16 - 1) / (MT / 16), nsets), dim3(MT), 0, (hipStream_t)stream, e0, e1, 0,
                          (const uint4 *)v->d_slot_desc, (const int4 *)v->d_slot_rec, v->root_c1, v->L - levelsup, (const uint4 *)desc_dev, counts_dev, cap, 0,
                          word_id_dev, node_id_dev);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbm_match_triangulation(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbx_keypoint *kps2,
                             const uint8_t *desc2, int n2, const int32_t *cand_off, const int32_t *cand_idx,
                             const uint8_t *has_mappoint1, const uint8_t *has_mappoint2, const uint8_t *stereo1,
                             const uint8_t *stereo2, int only_stereo, const float *F12, float ex, float ey,
                             const float *scale_factors2, const float *level_sigma2, int nlevels, int32_t *match12,
                             int32_t *best_dist)
{
    if (n1 < 0 || n2 < 0 || nlevels < 1 || (n1 && (!kps1 || !desc1 || !has_mappoint1 || !stereo1 || !match12 || !best_dist)) ||
        !cand_off || !F12 || !scale_factors2 || !level_sigma2)
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (n1 == 0) return ORBX_OK;
    const int nc = cand_off[n1];
    if (nc < 0 || (nc && (!cand_idx || !kps2 || !desc2 || !has_mappoint2 || !stereo2))) ORBX_FAIL(ORBX_ERR_ARG, "bad candidate lists");
    for (int k = 0; k < nc; ++k)
        if (cand_idx[k] < 0 || cand_idx[k] >= n2) ORBX_FAIL(ORBX_ERR_ARG, "candidate index out of range");
    for (int i = 0; i < n1; ++i)
        if (cand_off[i] > cand_off[i + 1] || cand_off[i] < 0) ORBX_FAIL(ORBX_ERR_ARG, "candidate offsets not monotone");
    for (int i = 0; i < n1; ++i)
        if (cand_off[i + 1] - cand_off[i] > 65535) ORBX_FAIL(ORBX_ERR_CAPACITY, "more than 65,535 candidates for one keypoint (the tie rule's position field)");
    for (int j = 0; j < n2; ++j)
        if (kps2[j].octave < 0 || kps2[j].octave >= nlevels) ORBX_FAIL(ORBX_ERR_ARG, "octave out of range");
    StagedCall sc;
    const size_t o_k1 = sc.in(kps1, sizeof(orbx_keypoint) * n1), o_a = sc.in(desc1, (size_t)32 * n1),
                 o_k2 = sc.in(kps2, sizeof(orbx_keypoint) * (size_t)n2), o_b = sc.in(desc2, (size_t)32 * n2),
                 o_off = sc.in(cand_off, sizeof(int) * (n1 + 1)), o_ci = sc.in(cand_idx, sizeof(int) * (size_t)nc),
                 o_m1 = sc.in(has_mappoint1, n1), o_m2 = sc.in(has_mappoint2, n2), o_s1 = sc.in(stereo1, n1), o_s2 = sc.in(stereo2, n2),
                 o_sc = sc.in(scale_factors2, sizeof(float) * nlevels), o_sg = sc.in(level_sigma2, sizeof(float) * nlevels),
                 o_o = sc.out(sizeof(int) * 2 * (size_t)n1);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    TriParams tp;
    for (int i = 0; i < 9; ++i) tp.F12[i] = F12[i];
    tp.ex = ex; tp.ey = ey; tp.only_stereo = only_stereo ? 1 : 0;
    int *ob = sc.d<int>(o_o);
    hipLaunchKernelGGL(k_match_triang, dim3((unsigned)(((size_t)n1 * 16 + MT - 1) / MT)), dim3(MT), 0, sc.stream(), sc.d<const orbx_keypoint>(o_k1),
                       sc.d<const uint4>(o_a), n1, sc.d<const orbx_keypoint>(o_k2), sc.d<const uint4>(o_b), sc.d<const int>(o_off),
                       sc.d<const int>(o_ci), sc.d<const uint8_t>(o_m1), sc.d<const uint8_t>(o_m2), sc.d<const uint8_t>(o_s1),
                       sc.d<const uint8_t>(o_s2), tp, sc.d<const float>(o_sc), sc.d<const float>(o_sg), ob, ob + n1);
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(match12, sc.r<int>(o_o), sizeof(int) * n1);
    memcpy(best_dist, sc.r<int>(o_o) + n1, sizeof(int) * n1);
    return ORBX_OK;
}

// The whole ORBmatcher::SearchForTriangulation (ORBmatcher.cc:858-1024) in one call: the FeatureVector co-iteration (:881-891,
// :1004-1012: equal node ids -> every keypoint of the node in KF1 scans the node's members of KF2, in member order), the gated loop
// on the device (orbm_match_triangulation), the rotation histogram, ComputeThreeMaxima and the rejection (:992-1012).  A
// FeatureVector = (nodes ascending, off, items), as for orbm_search_by_bow.  match12[n1] = index in KF2 or -1; the pair list
// vMatchedPairs is its non-negative entries in index order (:1014-1021).
int orbm_search_for_triangulation(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const int32_t *nodes1, const int32_t *off1,
                                  const int32_t *items1, int nn1, const uint8_t *has_mappoint1, const uint8_t *stereo1,
                                  const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const int32_t *nodes2, const int32_t *off2,
                                  const int32_t *items2, int nn2, const uint8_t *has_mappoint2, const uint8_t *stereo2, int only_stereo,
                                  const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2, int nlevels,
                                  int check_orientation, int32_t *match12, int *nmatches)
{
    if (n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || !nmatches || (n1 && !match12) || (nn1 && (!nodes1 || !off1 || !items1)) ||
        (nn2 && (!nodes2 || !off2 || !items2)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    *nmatches = 0;
    if (n1 == 0) return ORBX_OK;
    for (int a = 0; a < nn1; ++a)
        for (int k = off1[a]; k < off1[a + 1]; ++k)
            if (items1[k] < 0 || items1[k] >= n1) ORBX_FAIL(ORBX_ERR_ARG, "feature index out of range");
    for (int b = 0; b < nn2; ++b)
        for (int k = off2[b]; k < off2[b + 1]; ++k)
            if (items2[k] < 0 || items2[k] >= n2) ORBX_FAIL(ORBX_ERR_ARG, "feature index out of range");
    std::vector<int32_t> cnt(n1, 0), start2(n1, 0), cand_off((size_t)n1 + 1, 0);
    for (int a = 0, b = 0; a < nn1 && b < nn2;) {
        if (nodes1[a] == nodes2[b]) {
            for (int k = off1[a]; k < off1[a + 1]; ++k) { cnt[items1[k]] = off2[b + 1] - off2[b]; start2[items1[k]] = off2[b]; }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) ++a;     // lower_bound on the other map (:1004-1011)
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) ++b;
        }
    }
    for (int i = 0; i < n1; ++i) {
        if ((long long)cand_off[i] + cnt[i] > 0x7fffffffll) ORBX_FAIL(ORBX_ERR_CAPACITY, "candidate lists exceed 2^31 entries");
        cand_off[i + 1] = cand_off[i] + cnt[i];
    }
    std::vector<int32_t> cand((size_t)std::max(cand_off[n1], 1));
    for (int i = 0; i < n1; ++i)
        if (cnt[i]) memcpy(&cand[cand_off[i]], items2 + start2[i], sizeof(int32_t) * (size_t)cnt[i]);
    std::vector<int32_t> best((size_t)n1);
    const int rc = orbm_match_triangulation(kps1, desc1, n1, kps2, desc2, n2, cand_off.data(), cand.data(), has_mappoint1, has_mappoint2,
                                            stereo1, stereo2, only_stereo, F12, ex, ey, scale_factors2, level_sigma2, nlevels, match12, best.data());
    if (rc != ORBX_OK) return rc;
    if (check_orientation) {
        constexpr int HL = 30;     // HISTO_LENGTH, ORBmatcher.cc:40
        int hist[HL] = {0};
        std::vector<int> bin((size_t)n1, -1);
        const float factor = 1.0f / HL;
        for (int i = 0; i < n1; ++i)
            if (match12[i] >= 0) {
                float rot = kps1[i].angle - kps2[match12[i]].angle;     // :994-1001
                if (rot < 0.0f) rot += 360.0f;
                int b = (int)roundf(rot * factor);
                if (b == HL) b = 0;
                bin[i] = b; hist[b]++;
            }
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;      // ComputeThreeMaxima, :1802-1843
        for (int i = 0; i < HL; i++) {
            const int sz = hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < n1; ++i)
            if (bin[i] >= 0 && bin[i] != ind1 && bin[i] != ind2 && bin[i] != ind3) match12[i] = -1;
    }
    int nm = 0;
    for (int i = 0; i < n1; ++i) nm += match12[i] >= 0;
    *nmatches = nm;
    return ORBX_OK;
}

int orbm_hamming_matrix(const uint8_t *A, int nA, const uint8_t *B, int nB, uint16_t *out)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || ((nA && nB) && !out)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (nA == 0 || nB == 0) return ORBX_OK;
    StagedCall sc;
    const size_t o_a = sc.in(A, (size_t)nA * 32), o_b = sc.in(B, (size_t)nB * 32), o_o = sc.out(sizeof(uint16_t) * (size_t)nA * nB);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nB + MT - 1) / MT, nA), dim3(MT), 0, sc.stream(), sc.d<const uint4>(o_a), nA,
                       sc.d<const uint4>(o_b), nB, sc.d<unsigned short>(o_o));
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(out, sc.r<uint16_t>(o_o), sizeof(uint16_t) * (size_t)nA * nB);
    return ORBX_OK;
}

int orbm_set_allpairs_kernel(int kind)
{
    if (kind != ORBM_ALLPAIRS_AUTO && kind != ORBM_ALLPAIRS_POPCOUNT && kind != ORBM_ALLPAIRS_MFMA) return ORBX_ERR_ARG;
    return g_allpairs_kind.exchange(kind);
}

int orbm_profile_enable(int on)      // bit 0: the all-pairs matcher's launches, bit 1: orbm_bow_transform_batch_dev's
{
    g_prof.names[0] = "k_match_sets_mfma"; g_prof.names[1] = "k_bow_transform";
    g_prof.reset();
    g_prof.mask = (unsigned)on & 3u;
    return ORBX_OK;
}

int orbm_profile_read(double *total_ms, int64_t *launches)
{
    g_prof.flush();
    if (total_ms) *total_ms = g_prof.ms[0];
    if (launches) *launches = g_prof.launches[0];
    return ORBX_OK;
}

int orbm_profile_read_bow(double *total_ms, int64_t *launches)
{
    g_prof.flush();
    if (total_ms) *total_ms = g_prof.ms[1];
    if (launches) *launches = g_prof.launches[1];
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
