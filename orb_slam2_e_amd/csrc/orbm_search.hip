// orbm_search.hip -- whole ORBmatcher search loops for gfx950, with the reference's
// in-loop bookkeeping and its rotation-consistency check:
//   SearchByProjection(Frame, vector<MapPoint*>, th)         src/ORBmatcher.cc:46-132
//   SearchByProjection(CurrentFrame, LastFrame, th, mono)    :1529-1671
//   SearchByProjection(CurrentFrame, KeyFrame, found, ...)   :1673-1800
//   SearchByProjection(KeyFrame, Scw, points, matched, th)   :491-604
//   SearchForInitialization(F1, F2, prevMatched, ...)        :606-721
//
// These loops are sequential in the reference: a query sees what earlier queries left in
// mvpMapPoints / vMatchedDistance.  Here
//   1. the frame's keypoints are sorted by (grid cell, index), so a scan in array order
//      visits candidates exactly in Frame::GetFeaturesInArea order (src/Frame.cc:342-395);
//   2. k_win_count / k_win_fill build, fully in parallel, every query's candidate list with
//      its Hamming distances (the expensive part: window tests + 256-bit distances);
//   3. k_resolve walks the queries in order, 64 at a time: every lane selects best / second
//      for its query against the committed state; a lane whose best or second candidate
//      was claimed by an earlier lane of the same batch is a conflict; lanes below the
//      first conflict commit, the rest select again.  A lane's selection can only change
//      when an earlier query claims its best or second candidate, so the committed
//      results are the sequential loop's results;
//   4. k_rotation builds the 30-bin rotation histogram, ORBmatcher::ComputeThreeMaxima
//      (:1802-1843) and rejects the matches outside the three main bins.
#include <limits.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "orbm_internal.h"

using namespace orbm_detail;

namespace {

constexpr int SEQ_CAP = 8192;      // candidate entries staged in LDS per batch of queries
constexpr int SEQ_MAXN = 8192;     // keypoints per frame the resolver's LDS state holds
constexpr int HISTO_LENGTH = 30;   // ORBmatcher.cc:40
enum { ACCEPT_BEST = 0, ACCEPT_RATIO_SAME_LEVEL = 1, ACCEPT_RATIO = 2 };

// Sorted keypoint record: position sp in this array = rank in GetFeaturesInArea order.
struct SeqKp { float x, y, uright; int octave; };

// One lane per query, the sorted keypoints streamed through LDS.  FILL = 0: count the
// candidates of each query (window + level + stereo tests of Frame.cc:365-390 and
// ORBmatcher.cc:91-96); FILL = 1: also the distances, entry = dist << 20 | octave << 16 | sp.
template <int FILL>
__global__ __launch_bounds__(MT) void k_win_list(const WinQuery *__restrict__ q, const uint4 *__restrict__ A, int nq,
                                                 const SeqKp *__restrict__ kp, const uint4 *__restrict__ B, int ns, int has_uright,
                                                 int init_dist, int *__restrict__ cnt, const int *__restrict__ off,
                                                 unsigned *__restrict__ ent)
{
    __shared__ SeqKp s_kp[MT];
    __shared__ uint4 s_d[FILL ? MT * 2 : 1];
    const int i = blockIdx.x * MT + threadIdx.x, tid = threadIdx.x;
    const bool act = i < nq;
    WinQuery w = {0.f, 0.f, -1.f, 0.f, 0, -1};
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (act) {
        w = q[i];
        if (FILL) { a0 = A[2 * i]; a1 = A[2 * i + 1]; }
    }
    const bool check_levels = (w.min_level > 0) || (w.max_level >= 0);
    int c = 0;
    unsigned *out = FILL && act ? ent + off[i] : nullptr;
    for (int j0 = 0; j0 < ns; j0 += MT) {
        __syncthreads();
        if (j0 + tid < ns) {
            s_kp[tid] = kp[j0 + tid];
            if (FILL) { s_d[2 * tid] = B[2 * (j0 + tid)]; s_d[2 * tid + 1] = B[2 * (j0 + tid) + 1]; }
        }
        __syncthreads();
        const int nt = min(MT, ns - j0);
        for (int j = 0; j < nt; ++j) {
            const SeqKp k = s_kp[j];
            bool ok = act;
            if (check_levels) ok = ok && !(k.octave < w.min_level) && !(w.max_level >= 0 && k.octave > w.max_level);
            const float distx = k.x - w.u, disty = k.y - w.v;
            ok = ok && fabsf(distx) < w.r && fabsf(disty) < w.r;
            if (has_uright && k.uright > 0) ok = ok && !(fabsf(w.xr - k.uright) > w.r);
            if (FILL) {
                if (ok) {
                    const int dist = popc256(a0, a1, s_d[2 * j], s_d[2 * j + 1]);
                    // dist<bestDist / dist<bestDist2 can only fire below the initial value; dropped
                    // entries keep their slot (count pass is distance-free) as "never selectable"
                    out[c] = dist < init_dist ? ((unsigned)dist << 20) | ((unsigned)k.octave << 16) | (unsigned)(j0 + j) : 0xffffffffu;
                }
            }
            c += ok;
        }
    }
    if (!FILL && act) cnt[i] = c;
}

// Entries for explicit candidate lists (BoW-node members in member order): one lane per
// query, entry = dist << 20 | candidate index; a distance of 256 can never be selected.
__global__ __launch_bounds__(MT) void k_list_fill(const uint4 *__restrict__ A, int nq, const uint4 *__restrict__ B,
                                                  const int *__restrict__ off, const int *__restrict__ cand,
                                                  unsigned *__restrict__ ent)
{
    const int i = blockIdx.x * MT + threadIdx.x;
    if (i >= nq) return;
    const uint4 a0 = A[2 * i], a1 = A[2 * i + 1];
    for (int k = off[i]; k < off[i + 1]; ++k) {
        const int j = cand[k];
        const int dist = popc256(a0, a1, B[2 * j], B[2 * j + 1]);
        ent[k] = dist < 256 ? ((unsigned)dist << 20) | (unsigned)j : 0xffffffffu;
    }
}

// Exclusive scan of cnt[0..n) into off[0..n], one block (n is a few thousand).
__global__ __launch_bounds__(MT) void k_scan_counts(const int *__restrict__ cnt, int n, int *__restrict__ off)
{
    __shared__ int part[MT];
    const int tid = threadIdx.x, per = (n + MT - 1) / MT, lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int r = 0;
        for (int i = 0; i < MT; ++i) { const int v = part[i]; part[i] = r; r += v; }
        off[n] = r;
    }
    __syncthreads();
    int r = part[tid];
    for (int i = lo; i < hi; ++i) { off[i] = r; r += cnt[i]; }
}

// The sequential loop, 64 queries per batch (see the file header).  MODE 0 = projection
// family: state = blocked[sp] ("mvpMapPoints[sp] holds a point later queries must skip"),
// match_kp[sp] = last query assigned to the slot.  MODE 1 = SearchForInitialization:
// state = vMatchedDistance[sp], vnMatches21[sp], vnMatches12[i].
// acc_sp[i] = candidate chosen by query i when it was accepted (else -1).
template <int MODE>
__global__ __launch_bounds__(64) void k_resolve(const unsigned *__restrict__ ent, const int *__restrict__ off, int nq, int ns,
                                                const uint8_t *__restrict__ takes, int th, float nnratio, int accept_mode,
                                                int *__restrict__ acc_sp, int *__restrict__ out_a, int *__restrict__ nmatches)
{
    extern __shared__ __align__(16) unsigned sm[];
    unsigned *s_ent = sm;
    int *s_a = reinterpret_cast<int *>(sm + SEQ_CAP);                // MODE 0: match_kp[ns]   MODE 1: vMatchedDistance[ns]
    int *s_b = s_a + ns;                                             // MODE 0: blocked bytes  MODE 1: vnMatches21[ns]
    int *s_c = s_b + ns;                                             // MODE 1: vnMatches12[nq]
    uint8_t *s_blk = reinterpret_cast<uint8_t *>(s_b);
    const int lane = threadIdx.x;
    for (int j = lane; j < ns; j += 64) {
        s_a[j] = MODE == 0 ? -1 : INT_MAX;
        if (MODE == 1) s_b[j] = -1;
    }
    if (MODE == 0) for (int j = lane; j < (ns + 3) / 4; j += 64) s_b[j] = 0;
    if (MODE == 1) for (int j = lane; j < nq; j += 64) s_c[j] = -1;
    __syncthreads();
    int nm = 0;
    for (int i0 = 0; i0 < nq;) {
        // batch = the leading queries whose entries fit the staging buffer (at least one)
        const int lo = off[i0];
        const bool in = i0 + lane < nq;
        const int b_g = in ? off[i0 + lane] : 0, e_g = in ? off[i0 + lane + 1] : 0;
        const unsigned long long fits = __ballot(in && e_g - lo <= SEQ_CAP);
        int nb = fits == ~0ull ? 64 : __ffsll((long long)~fits) - 1;
        const unsigned *src = s_ent;
        int base = lo;
        if (nb == 0) { nb = 1; src = ent; base = 0; }               // one oversized list: read it from HBM
        else {
            const int hi = __shfl(e_g, nb - 1);
            for (int k = lo + lane; k < hi; k += 64) s_ent[k - lo] = ent[k];
        }
        __syncthreads();
        const int i = i0 + lane, b = b_g - base, e = e_g - base;
        unsigned long long pending = nb == 64 ? ~0ull : (1ull << nb) - 1ull;
        const int tk = (MODE == 0 && lane < nb) ? takes[i] : 1;
        while (pending) {
            const bool mine = (pending >> lane) & 1ull;
            unsigned k1 = 0xffffffffu, k2 = 0xffffffffu;
            if (mine) {
                for (int k = b; k < e; ++k) {
                    const unsigned en = src[k];
                    const int sp = en & 0xffffu, dist = (int)(en >> 20);
                    const bool ok = en != 0xffffffffu && (MODE == 0 ? s_blk[sp] == 0 : s_a[sp] > dist);
                    if (ok) {
                        const unsigned key = ((unsigned)dist << 16) | (unsigned)(k - b);
                        const unsigned hi = max(k1, key);
                        k2 = min(k2, hi);
                        k1 = min(k1, key);
                    }
                }
            }
            int sp1 = -1, sp2 = -1, best = INT_MAX, second = INT_MAX, l1 = -1, l2 = -1;
            if (k1 != 0xffffffffu) { const unsigned en = src[b + (k1 & 0xffffu)]; sp1 = en & 0xffffu; l1 = (en >> 16) & 15; best = (int)(k1 >> 16); }
            if (k2 != 0xffffffffu) { const unsigned en = src[b + (k2 & 0xffffu)]; sp2 = en & 0xffffu; l2 = (en >> 16) & 15; second = (int)(k2 >> 16); }
            bool acc = mine && sp1 >= 0 && best <= th;
            if (MODE == 0) {
                // initial bestDist2 = 256 when there is no second candidate (:79-81)
                const int sec = sp2 >= 0 ? second : 256;
                if (accept_mode == ACCEPT_RATIO_SAME_LEVEL && l1 == l2 && (float)best > nnratio * (float)sec) acc = false; // :121
                if (accept_mode == ACCEPT_RATIO && !((float)best < nnratio * (float)sec)) acc = false;                    // :431, :801
            } else {
                acc = acc && (float)best < (float)second * nnratio; // INT_MAX when alone (:637-638,674-676)
            }
            // conflicts: an earlier accepted lane of this batch claims my best or second candidate
            bool conflict = false;
            for (unsigned long long m = __ballot(acc && tk); m; m &= m - 1ull) {
                const int l = __ffsll((long long)m) - 1;
                const int x = __shfl(sp1, l);
                conflict = conflict || (mine && lane > l && (x == sp1 || x == sp2));
            }
            const unsigned long long cm = __ballot(conflict);
            const int lc = cm ? __ffsll((long long)cm) - 1 : 64;
            const bool commit = mine && lane < lc;
            int old = -1;
            if (commit) {
                acc_sp[i] = acc ? sp1 : -1;
                if (acc) {
                    if (MODE == 0) {
                        atomicMax(&s_a[sp1], i);               // the last query assigned to the slot stays
                        if (tk) s_blk[sp1] = 1;
                    } else {                                   // accepted candidates of one pass are distinct
                        old = s_b[sp1];
                        if (old >= 0) s_c[old] = -1;           // steal (:678-682)
                        s_c[i] = sp1; s_b[sp1] = i; s_a[sp1] = best;
                    }
                }
            }
            nm += __popcll(__ballot(commit && acc)) - __popcll(__ballot(old >= 0));
            pending &= ~((lc == 64 ? ~0ull : (1ull << lc) - 1ull));
            __syncthreads();
        }
        i0 += nb;
    }
    __syncthreads();
    if (MODE == 0) for (int j = lane; j < ns; j += 64) out_a[j] = s_a[j];   // match_kp by sorted position
    else for (int j = lane; j < nq; j += 64) out_a[j] = s_c[j];            // vnMatches12 by sorted position
    if (lane == 0) *nmatches = nm;
}

// Rotation histogram + ComputeThreeMaxima (:1802-1843) + rejection, and the translation of
// sorted positions back to keypoint indices.  One block.
// MODE 0: match_kp[perm[sp]] = query (or -1 untouched / -2 cleared), match_q[i] = keypoint.
// MODE 1: match12[i] = keypoint or -1.
template <int MODE>
__global__ __launch_bounds__(MT) void k_rotation(const int *__restrict__ acc_sp, const int *__restrict__ state, int nq, int ns,
                                                 const float *__restrict__ qangle, const float *__restrict__ kangle,
                                                 const int *__restrict__ perm, int check, int *__restrict__ match_q,
                                                 int *__restrict__ match_kp, int *__restrict__ nmatches)
{
    __shared__ int hist[HISTO_LENGTH], keep[3], removed;
    const int tid = threadIdx.x;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) removed = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    if (check)
        for (int i = tid; i < nq; i += MT) {
            const int sp = acc_sp[i];
            if (sp >= 0) {
                float rot = qangle[i] - kangle[sp];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                atomicAdd(&hist[bin], 1);
            }
        }
    __syncthreads();
    if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
    }
    if (MODE == 0) for (int j = tid; j < ns; j += MT) match_kp[perm[j]] = state[j];
    __syncthreads();
    for (int i = tid; i < nq; i += MT) {
        const int sp = acc_sp[i];
        int out = MODE == 0 ? (sp >= 0 ? perm[sp] : -1) : (state[i] >= 0 ? perm[state[i]] : -1);
        if (check && sp >= 0) {
            float rot = qangle[i] - kangle[sp];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin != keep[0] && bin != keep[1] && bin != keep[2]) {
                if (MODE == 0) { match_kp[perm[sp]] = -2; atomicAdd(&removed, 1); }   // every entry of the bin: slot cleared, nmatches--
                else if (state[i] >= 0) { out = -1; atomicAdd(&removed, 1); }         // only matches still standing (:704-708)
            }
        }
        match_q[i] = out;
    }
    __syncthreads();
    if (tid == 0) *nmatches -= removed;
}

struct SortedFrame {
    std::vector<SeqKp> kp;
    std::vector<int> perm;
    std::vector<float> angle;
    std::vector<uint8_t> desc;
};

// Keypoints in the grid and not excluded, ordered by (cell, index).
void sort_frame(const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *skip, const float *uright, float min_x,
                float min_y, float max_x, float max_y, SortedFrame &sf)
{
    std::vector<WinKp> wk;
    build_winkp(kps, n, skip, uright, min_x, min_y, max_x, max_y, wk);
    std::vector<unsigned> order;
    order.reserve(n);
    for (int j = 0; j < n; ++j)
        if (wk[j].order != 0xffffffffu) order.push_back(wk[j].order);
    std::sort(order.begin(), order.end());
    const size_t ns = order.size();
    sf.kp.resize(ns ? ns : 1); sf.perm.resize(ns ? ns : 1); sf.angle.resize(ns ? ns : 1); sf.desc.resize(32 * (ns ? ns : 1));
    for (size_t s = 0; s < ns; ++s) {
        const int j = (int)(order[s] & 0xffffu);
        sf.kp[s] = {wk[j].x, wk[j].y, wk[j].uright, wk[j].octave};
        sf.perm[s] = j;
        sf.angle[s] = kps[j].angle;
        memcpy(&sf.desc[32 * s], desc + 32 * (size_t)j, 32);
    }
    sf.kp.resize(ns); sf.perm.resize(ns); sf.angle.resize(ns); sf.desc.resize(32 * ns);
}

// Shared driver.  mode 0: projection family; mode 1: SearchForInitialization.
int run_sequential(int mode, const WinQuery *queries, const uint8_t *qdesc, const float *qangle, const uint8_t *qtakes, int nq,
                   const SortedFrame &sf, int n, int has_uright, int th, float nnratio, int accept_mode, int check,
                   int32_t *match_kp, int32_t *match_q, int *nmatches, const int32_t *cand_off = nullptr,
                   const int32_t *cand_idx = nullptr)
{
    const int ns = (int)sf.kp.size();
    if (ns > SEQ_MAXN || nq > 65536) ORBX_FAIL(ORBX_ERR_CAPACITY, "frame too large for the sequential resolver");
    for (int j = 0; j < n && mode == 0; ++j) match_kp[j] = -1;
    for (int i = 0; i < nq; ++i) match_q[i] = -1;
    *nmatches = 0;
    if (nq == 0 || ns == 0) return ORBX_OK;
    DevBuf dq, da, dk, db, dang, dqang, dperm, dtk, dcnt, doff, dacc, dstate, dmq, dmk, dnm, dent;
    if (dq.alloc(sizeof(WinQuery) * nq) || da.alloc((size_t)32 * nq) || dk.alloc(sizeof(SeqKp) * ns) || db.alloc((size_t)32 * ns) ||
        dang.alloc(sizeof(float) * ns) || dqang.alloc(sizeof(float) * nq) || dperm.alloc(sizeof(int) * ns) || dtk.alloc(nq) ||
        dcnt.alloc(sizeof(int) * nq) || doff.alloc(sizeof(int) * (nq + 1)) || dacc.alloc(sizeof(int) * nq) ||
        dstate.alloc(sizeof(int) * std::max(ns, nq)) || dmq.alloc(sizeof(int) * nq) || dmk.alloc(sizeof(int) * (n ? n : 1)) ||
        dnm.alloc(sizeof(int)))
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    if (queries) ORBX_HIP(hipMemcpy(dq.p, queries, sizeof(WinQuery) * nq, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(da.p, qdesc, (size_t)32 * nq, hipMemcpyHostToDevice));
    if (queries) ORBX_HIP(hipMemcpy(dk.p, sf.kp.data(), sizeof(SeqKp) * ns, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(db.p, sf.desc.data(), (size_t)32 * ns, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(dang.p, sf.angle.data(), sizeof(float) * ns, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(dperm.p, sf.perm.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
    if (qangle) ORBX_HIP(hipMemcpy(dqang.p, qangle, sizeof(float) * nq, hipMemcpyHostToDevice));
    else ORBX_HIP(hipMemset(dqang.p, 0, sizeof(float) * nq));
    if (qtakes) ORBX_HIP(hipMemcpy(dtk.p, qtakes, nq, hipMemcpyHostToDevice));
    else ORBX_HIP(hipMemset(dtk.p, 1, nq));
    ORBX_HIP(hipMemset(dmk.p, 0xff, sizeof(int) * (n ? n : 1))); // -1: slot untouched
    const int init_dist = mode == 0 ? 256 : INT_MAX;
    const dim3 g((nq + MT - 1) / MT);
    if (cand_off) { // explicit candidate lists
        DevBuf dcand;
        const int total = cand_off[nq];
        if (dcand.alloc(sizeof(int) * (size_t)(total ? total : 1)) || dent.alloc(sizeof(unsigned) * (size_t)(total ? total : 1)))
            ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
        ORBX_HIP(hipMemcpy(doff.p, cand_off, sizeof(int) * (nq + 1), hipMemcpyHostToDevice));
        if (total) ORBX_HIP(hipMemcpy(dcand.p, cand_idx, sizeof(int) * total, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_list_fill, g, dim3(MT), 0, 0, (const uint4 *)da.p, nq, (const uint4 *)db.p, (const int *)doff.p,
                           (const int *)dcand.p, (unsigned *)dent.p);
        ORBX_HIP(hipGetLastError());
        ORBX_HIP(hipDeviceSynchronize()); // dcand is released at the end of this scope
    } else {
        hipLaunchKernelGGL(k_win_list<0>, g, dim3(MT), 0, 0, (const WinQuery *)dq.p, (const uint4 *)da.p, nq, (const SeqKp *)dk.p,
                           (const uint4 *)db.p, ns, has_uright, init_dist, (int *)dcnt.p, (const int *)nullptr, (unsigned *)nullptr);
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(MT), 0, 0, (const int *)dcnt.p, nq, (int *)doff.p);
        ORBX_HIP(hipGetLastError());
        int total = 0;
        ORBX_HIP(hipMemcpy(&total, (int *)doff.p + nq, sizeof(int), hipMemcpyDeviceToHost));
        if (dent.alloc(sizeof(unsigned) * (size_t)(total ? total : 1))) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
        hipLaunchKernelGGL(k_win_list<1>, g, dim3(MT), 0, 0, (const WinQuery *)dq.p, (const uint4 *)da.p, nq, (const SeqKp *)dk.p,
                           (const uint4 *)db.p, ns, has_uright, init_dist, (int *)nullptr, (const int *)doff.p, (unsigned *)dent.p);
    }
    const size_t lds = sizeof(unsigned) * SEQ_CAP + sizeof(int) * (2 * (size_t)ns + (mode == 1 ? nq : 0)) + 16;
    if (lds > 160 * 1024) ORBX_FAIL(ORBX_ERR_CAPACITY, "resolver state exceeds LDS");
    if (mode == 0) {
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_resolve<0>, dim3(1), dim3(64), lds, 0, (const unsigned *)dent.p, (const int *)doff.p, nq, ns,
                           (const uint8_t *)dtk.p, th, nnratio, accept_mode, (int *)dacc.p, (int *)dstate.p, (int *)dnm.p);
        hipLaunchKernelGGL(k_rotation<0>, dim3(1), dim3(MT), 0, 0, (const int *)dacc.p, (const int *)dstate.p, nq, ns,
                           (const float *)dqang.p, (const float *)dang.p, (const int *)dperm.p, check, (int *)dmq.p, (int *)dmk.p,
                           (int *)dnm.p);
    } else {
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_resolve<1>, dim3(1), dim3(64), lds, 0, (const unsigned *)dent.p, (const int *)doff.p, nq, ns,
                           (const uint8_t *)dtk.p, th, nnratio, 0, (int *)dacc.p, (int *)dstate.p, (int *)dnm.p);
        hipLaunchKernelGGL(k_rotation<1>, dim3(1), dim3(MT), 0, 0, (const int *)dacc.p, (const int *)dstate.p, nq, ns,
                           (const float *)dqang.p, (const float *)dang.p, (const int *)dperm.p, check, (int *)dmq.p, (int *)dmk.p,
                           (int *)dnm.p);
    }
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(match_q, dmq.p, sizeof(int) * nq, hipMemcpyDeviceToHost));
    if (mode == 0 && n) ORBX_HIP(hipMemcpy(match_kp, dmk.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    ORBX_HIP(hipMemcpy(nmatches, dnm.p, sizeof(int), hipMemcpyDeviceToHost));
    return ORBX_OK;
}

} // namespace

extern "C" {

int orbm_search_projection(const orbm_window_query *queries, const uint8_t *qdesc, const float *qangle, const uint8_t *qtakes,
                           int nq, const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *occupied,
                           const float *uright, float min_x, float min_y, float max_x, float max_y, int th_accept, float nnratio,
                           int ratio_same_level, int check_orientation, int32_t *match_kp, int32_t *match_q, int *nmatches)
{
    if (nq < 0 || n < 0 || n > 65535 || (nq && (!queries || !qdesc || !match_q)) || (n && (!kps || !desc || !match_kp)) || !nmatches ||
        (check_orientation && nq && !qangle) || !(max_x > min_x) || !(max_y > min_y))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    SortedFrame sf;
    sort_frame(kps, desc, n, occupied, uright, min_x, min_y, max_x, max_y, sf);
    return run_sequential(0, reinterpret_cast<const WinQuery *>(queries), qdesc, qangle, qtakes, nq, sf, n, uright ? 1 : 0,
                          th_accept, nnratio, ratio_same_level ? ACCEPT_RATIO_SAME_LEVEL : ACCEPT_BEST, check_orientation, match_kp,
                          match_q, nmatches);
}

int orbm_search_for_initialization(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbx_keypoint *kps2,
                                   const uint8_t *desc2, int n2, float *prev_matched, float min_x, float min_y, float max_x,
                                   float max_y, int window_size, float nnratio, int check_orientation, int32_t *matches12,
                                   int *nmatches)
{
    if (n1 < 0 || n2 < 0 || n2 > 65535 || (n1 && (!kps1 || !desc1 || !prev_matched || !matches12)) || (n2 && (!kps2 || !desc2)) ||
        !nmatches || !(max_x > min_x) || !(max_y > min_y))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    SortedFrame sf;
    sort_frame(kps2, desc2, n2, nullptr, nullptr, min_x, min_y, max_x, max_y, sf);
    // queries: level-0 keypoints of F1 around their previous match (:619-626); others get an empty window
    std::vector<WinQuery> q(n1 ? n1 : 1);
    std::vector<float> ang(n1 ? n1 : 1);
    for (int i = 0; i < n1; ++i) {
        const bool use = !(kps1[i].octave > 0);
        q[i] = {prev_matched[2 * i], prev_matched[2 * i + 1], use ? (float)window_size : -1.0f, 0.f, kps1[i].octave, kps1[i].octave};
        ang[i] = kps1[i].angle;
    }
    std::vector<int32_t> dummy(n2 ? n2 : 1);
    const int rc = run_sequential(1, q.data(), desc1, ang.data(), nullptr, n1, sf, n2, 0, 45 /* TH_LOW, :38 */, nnratio, 0,
                                  check_orientation, dummy.data(), matches12, nmatches);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < n1; ++i) // :715-718
        if (matches12[i] >= 0) { prev_matched[2 * i] = kps2[matches12[i]].x; prev_matched[2 * i + 1] = kps2[matches12[i]].y; }
    return ORBX_OK;
}

int orbm_search_by_bow(const uint8_t *desc1, const float *angle1, int n1, const int32_t *qidx, int nq, const uint8_t *desc2,
                       const float *angle2, int n2, const int32_t *cand_off, const int32_t *cand_idx, int th, int strict_th,
                       float nnratio, int check_orientation, int32_t *match12, int32_t *match21, int *nmatches)
{
    if (n1 < 0 || n2 < 0 || nq < 0 || (n1 && (!desc1 || !match12)) || (n2 && !desc2) || (nq && (!qidx || !cand_off)) || !nmatches ||
        (check_orientation && nq && (!angle1 || !angle2)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    for (int i = 0; i < nq; ++i)
        if (qidx[i] < 0 || qidx[i] >= n1 || cand_off[i + 1] < cand_off[i]) ORBX_FAIL(ORBX_ERR_ARG, "bad query list");
    if (nq && cand_off[0] != 0) ORBX_FAIL(ORBX_ERR_ARG, "bad query list");
    for (int k = 0; nq && k < cand_off[nq]; ++k)
        if (!cand_idx || cand_idx[k] < 0 || cand_idx[k] >= n2) ORBX_FAIL(ORBX_ERR_ARG, "candidate index out of range");
    ORBX_NEED_DEVICE();
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    if (match21) for (int j = 0; j < n2; ++j) match21[j] = -1;
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    SortedFrame sf; // the second set as it is: position = feature index
    sf.kp.resize(n2); sf.perm.resize(n2); sf.angle.resize(n2); sf.desc.assign(desc2, desc2 + (size_t)32 * n2);
    for (int j = 0; j < n2; ++j) { sf.perm[j] = j; sf.angle[j] = angle2 ? angle2[j] : 0.f; }
    std::vector<uint8_t> qd((size_t)32 * nq);
    std::vector<float> qa(nq);
    for (int i = 0; i < nq; ++i) { memcpy(&qd[32 * (size_t)i], desc1 + 32 * (size_t)qidx[i], 32); qa[i] = angle1 ? angle1[qidx[i]] : 0.f; }
    std::vector<int32_t> mk(n2), mq(nq);
    // bestDist1 < TH_LOW (:799) == bestDist1 <= TH_LOW - 1
    const int rc = run_sequential(0, nullptr, qd.data(), qa.data(), nullptr, nq, sf, n2, 0, strict_th ? th - 1 : th, nnratio,
                                  ACCEPT_RATIO, check_orientation, mk.data(), mq.data(), nmatches, cand_off, cand_idx);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < nq; ++i) // every accepted query blocks its candidate, so slot mq[i] still names i unless rejected
        if (mq[i] >= 0 && mk[mq[i]] == i) { match12[qidx[i]] = mq[i]; if (match21) match21[mq[i]] = qidx[i]; }
    return ORBX_OK;
}

} // extern "C"
