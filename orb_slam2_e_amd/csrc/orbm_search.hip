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
//   1. the frame's keypoints are sorted by (grid cell, index): a window's candidates are a
//      few contiguous runs of that array, in Frame::GetFeaturesInArea order (src/Frame.cc:342-395);
//   2. k_win_wave (count, then fill; one wave per query) builds every query's candidate list
//      with its Hamming distances, k_topk keeps each list's 8 best sorted -- the expensive
//      part, fully parallel;
//   3. k_resolve walks the queries in order, 64 at a time: every lane selects best / second
//      for its query against the committed state; a lane whose best or second candidate
//      is claimed by an earlier lane of the same batch is a conflict; lanes below the
//      first conflict commit, the rest select again.  A lane's selection can only change
//      when an earlier query claims its best or second candidate, so the committed
//      results are the sequential loop's results;
//      Queries whose candidate sets cannot meet are independent of each other: the BoW searches hand
//      k_resolve one SEGMENT per vocabulary node (a feature belongs to one node, ORBmatcher.cc:384-456),
//      one workgroup per segment, so the nodes resolve side by side instead of one after the other;
//   4. k_rotation builds the 30-bin rotation histogram, ORBmatcher::ComputeThreeMaxima
//      (:1802-1843) and rejects the matches outside the three main bins.
// The host never waits in the middle of a call: the entry buffer is sized from a running estimate and the
// real total comes back with the results (a call that outgrew it is repeated once with the right size).
#include <limits.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "orbm_internal.h"
#include "orbx_internal.h"
#include "orbx_math.h"

using namespace orbm_detail;

namespace {

constexpr int SEQ_MAXN = 8192;     // keypoints per frame the resolver's LDS state holds
constexpr int HISTO_LENGTH = 30;   // ORBmatcher.cc:40
enum { ACCEPT_BEST = 0, ACCEPT_RATIO_SAME_LEVEL = 1, ACCEPT_RATIO = 2 };


constexpr int TOPK = 8; // best candidates per query kept sorted for the resolver


// Minimum over the wave, in every lane: a butterfly inside each row of 16 lanes on the DPP path (no LDS crossbar round trips),
// then the four row results through scalar registers.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1, 0, 3, 2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2, 3, 0, 1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

// The TOPK entries of a list with the smallest (distance, position) keys, in that
// order -- all the sequential resolver normally needs.  Run by the wave that has just written the list
// (its own stores are visible to it after the workgroup-scope fence).  Lists of up to 256 entries -- a window's usual few tens --
// are read ONCE, four entries per lane, and the rounds run on registers (every round used to re-read the list: eight dependent
// trips to L1 / L2 per query).
// the rounds on registers: every lane holds up to four (key, entry) pairs, keys unique (0xffffffff: none)
__device__ __forceinline__ void wave_topk_regs(unsigned (&key)[4], const unsigned (&en)[4], int lane, unsigned *__restrict__ top_i)
{
#pragma unroll
    for (int t = 0; t < TOPK; ++t) {
        const unsigned g = wave_min_u32(min(min(key[0], key[1]), min(key[2], key[3])));
        if (g == 0xffffffffu) { if (lane == 0) top_i[t] = 0xffffffffu; continue; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (key[r] == g) { top_i[t] = en[r]; key[r] = 0xffffffffu; }     // keys are unique: one lane, one slot
    }
}
__device__ __forceinline__ void wave_topk(const unsigned *__restrict__ ent, int b, int len, int lane, unsigned *__restrict__ top_i)
{
    if (len <= 256) {
        unsigned key[4], en[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = lane + 64 * r;
            en[r] = p < len ? ent[b + p] : 0xffffffffu;
            key[r] = en[r] != 0xffffffffu ? ((en[r] >> 20) << 16) | (unsigned)p : 0xffffffffu;
        }
        wave_topk_regs(key, en, lane, top_i);
        return;
    }
    unsigned prev = 0;
    bool done = false;
    for (int r = 0; r < TOPK; ++r) {
        unsigned g = 0xffffffffu;
        if (!done) {
            unsigned m = 0xffffffffu;
            for (int p = lane; p < len; p += 64) {
                const unsigned en = ent[b + p];
                if (en != 0xffffffffu) {
                    const unsigned key = ((en >> 20) << 16) | (unsigned)p;
                    if (r == 0 || key > prev) m = min(m, key);
                }
            }
            g = wave_min_u32(m);
            done = g == 0xffffffffu;
            prev = g;
        }
        if (lane == 0) top_i[r] = done ? 0xffffffffu : ent[b + (g & 0xffffu)];
    }
}

// One WAVE per query (4 per block).  Frame::GetFeaturesInArea (src/Frame.cc:342-395): the
// cell range of the window in the reference's float arithmetic, then, column by column,
// the keypoints of rows [nMinCellY, nMaxCellY] -- one contiguous run of the (cell, index)
// sorted array -- tested 64 at a time: level range, |dx| < r && |dy| < r, and the stereo
// check of ORBmatcher.cc:91-96.  FILL = 2 (the usual path) writes the survivors of query i into its own region of
// `stride` entries and the region's bounds into lbeg / lend, and raises *overflow if a list does not fit (the host
// then repeats the call on the exact path); FILL = 0 counts the survivors; FILL = 1 writes them in
// that order with their distances: entry = dist << 20 | octave << 16 | sp (0xffffffff for
// a distance that can never be selected).
template <int FILL>
__global__ __launch_bounds__(MT) void k_win_wave(const WinQuery *__restrict__ q, const uint4 *__restrict__ A, int nq,
                                                 const SeqKp *__restrict__ kp, const uint4 *__restrict__ B,
                                                 const int *__restrict__ cell_off, const uint8_t *__restrict__ occ, GridParams gp, int has_uright, int init_dist,
                                                 int *__restrict__ cnt, const int *__restrict__ off, unsigned *__restrict__ ent,
                                                 int stride, int *__restrict__ lbeg, int *__restrict__ lend,
                                                 int *__restrict__ overflow, unsigned *__restrict__ top, int gen)
{
    // (the query index is the same in all lanes: said so, the query record, its cell range and the runs' bounds become scalar loads)
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (MT / 64) + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const int obase = FILL == 2 ? i * stride : (FILL == 1 ? off[i] : 0), ocap = FILL == 2 ? stride : INT_MAX;
    const WinQuery w = q[i];
    int c = 0;
    const int nMinCellX = (int)fmaxf(0.f, floorf((w.u - gp.min_x - w.r) * gp.inv_w));
    const int nMaxCellX = (int)fminf((float)FRAME_GRID_COLS - 1, ceilf((w.u - gp.min_x + w.r) * gp.inv_w));
    const int nMinCellY = (int)fmaxf(0.f, floorf((w.v - gp.min_y - w.r) * gp.inv_h));
    const int nMaxCellY = (int)fminf((float)FRAME_GRID_ROWS - 1, ceilf((w.v - gp.min_y + w.r) * gp.inv_h));
    const bool none = !(w.r >= 0.f) || nMinCellX >= FRAME_GRID_COLS || nMaxCellX < 0 || nMinCellY >= FRAME_GRID_ROWS || nMaxCellY < 0;
    // FILL = 2: the entries a lane produces also stay in its registers (one per 64-candidate step, up to four steps): the short list is
    // then built without reading the list back (a fence and a memory round trip at the end of every wave's life)
    unsigned rkey[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, ren[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    int step = 0;
    if (!none) {
        const bool check_levels = (w.min_level > 0) || (w.max_level >= 0);
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (FILL) { a0 = A[2 * i]; a1 = A[2 * i + 1]; }
        unsigned *out = FILL ? ent + obase : nullptr;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            const int k0 = cell_off[ix * FRAME_GRID_ROWS + nMinCellY], k1 = cell_off[ix * FRAME_GRID_ROWS + nMaxCellY + 1];
            for (int kb = k0; kb < k1; kb += 64) {
                const int k = kb + lane;
                bool ok = k < k1;
                SeqKp p = {0.f, 0.f, 0.f, 0};
                uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
                if (ok) {
                    p = kp[k];
                    if (FILL) { b0 = B[2 * k]; b1 = B[2 * k + 1]; }    // with the keypoint, not behind its tests: one memory round trip less per step
                }
                if (occ && ok) ok = occ[k] == 0;   // the slot holds a point from the start (e.g. ORBmatcher.cc:87-89): never a candidate
                if (check_levels) ok = ok && !(p.octave < w.min_level) && !(w.max_level >= 0 && p.octave > w.max_level);
                const float distx = p.x - w.u, disty = p.y - w.v;
                ok = ok && fabsf(distx) < w.r && fabsf(disty) < w.r;
                if (has_uright && p.uright > 0) ok = ok && !(fabsf(w.xr - p.uright) > w.r);
                const unsigned long long bal = __ballot(ok);
                if (FILL && ok && c + __popcll(bal & lt) < ocap) {
                    const int dist = popc256(a0, a1, b0, b1), pos = c + __popcll(bal & lt);
                    const unsigned en = dist < init_dist ? ((unsigned)dist << 20) | ((unsigned)p.octave << 16) | (unsigned)k : 0xffffffffu;
                    out[pos] = en;
                    if (FILL == 2 && en != 0xffffffffu) {
                        const unsigned key = ((unsigned)dist << 16) | (unsigned)pos;
                        if (step == 0) { rkey[0] = key; ren[0] = en; } else if (step == 1) { rkey[1] = key; ren[1] = en; }
                        else if (step == 2) { rkey[2] = key; ren[2] = en; } else if (step == 3) { rkey[3] = key; ren[3] = en; }
                    }
                }
                c += __popcll(bal);
                ++step;
            }
        }
    }
    if (!FILL && lane == 0) cnt[i] = c;
    if (FILL) {
        if (FILL == 2 && lane == 0) {
            lbeg[i] = obase; lend[i] = obase + min(c, ocap);
            if (c > ocap) *overflow = gen;          // this call's generation number: some list did not fit (no preset needed)
        }
        if (FILL == 2 && step <= 4) {
            wave_topk_regs(rkey, ren, lane, top + (size_t)i * TOPK);
        } else {
            __threadfence_block();
            wave_topk(ent, obase, min(c, ocap), lane, top + (size_t)i * TOPK);
        }
    }
}

// Entries for explicit candidate lists (BoW-node members in member order), one wave per
// query: entry = dist << 20 | candidate index; a distance of 256 can never be selected.  Query i's candidates are
// cand[cbeg[i] .. cbeg[i] + off[i+1] - off[i]): the queries of one node share one copy of the node's member list.
// qmap (queries taken from a resident frame): query i is the feature at sorted position qmap[i] of that frame -- A and qang_src are the
// frame's arrays, and the query's angle is set down in qang_dst[i] for the rotation check.
__global__ __launch_bounds__(MT) void k_list_fill(const uint4 *__restrict__ A, int nq, const uint4 *__restrict__ B,
                                                  const int *__restrict__ off, const int *__restrict__ cbeg,
                                                  const int *__restrict__ cand, unsigned *__restrict__ ent,
                                                  unsigned *__restrict__ top, const int *__restrict__ qmap,
                                                  const float *__restrict__ qang_src, float *__restrict__ qang_dst)
{
    const int i = blockIdx.x * (MT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const int qi = qmap ? qmap[i] : i;
    if (qmap && lane == 0) qang_dst[i] = qang_src[qi];
    const uint4 a0 = A[2 * qi], a1 = A[2 * qi + 1];
    const int shift = cbeg[i] - off[i];
    for (int k = off[i] + lane; k < off[i + 1]; k += 64) {
        const int j = cand[k + shift];
        const int dist = popc256(a0, a1, B[2 * j], B[2 * j + 1]);
        ent[k] = dist < 256 ? ((unsigned)dist << 20) | (unsigned)j : 0xffffffffu;
    }
    __threadfence_block();
    wave_topk(ent, off[i], off[i + 1] - off[i], lane, top + (size_t)i * TOPK);
}

// Exclusive scan of cnt[0..n) into off[0..n], one block (n is a few thousand).
__global__ __launch_bounds__(MT) void k_scan_counts(const int *__restrict__ cnt, int n, int *__restrict__ off, int *__restrict__ total_out)
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
        *total_out = r;
    }
    __syncthreads();
    int r = part[tid];
    for (int i = lo; i < hi; ++i) { off[i] = r; r += cnt[i]; }
}

// The sequential loop, 64 queries per batch (see the file header).  MODE 0 = projection
// family / BoW: state = blocked[sp] ("the slot holds a point later queries must skip"),
// match_kp[sp] = last query assigned to the slot.  MODE 1 = SearchForInitialization:
// state = vMatchedDistance[sp], vnMatches21[sp], vnMatches12[i].
// A lane selects the first two eligible entries of its query's TOPK list (sorted by
// distance, then list position = the reference's strict-'<' first-wins order); only when
// fewer than it needs remain there and the list is longer does it scan the whole list.
// Conflicts go through a claim table: every accepted lane whose match blocks its slot
// claims it with its lane number (minimum wins); a lane whose best or second slot is
// claimed by a lower lane has to select again after that lane has committed.
// acc_sp[i] = candidate chosen by query i when it was accepted (else -1).
// seg != nullptr: workgroup s resolves the queries [seg[s], seg[s+1]) -- segments whose candidate sets are disjoint
// (BoW nodes) -- on its own copy of the state and writes back only the slots it touched (out_a is preset to -1, the
// match count is added up); seg == nullptr: one workgroup, all queries.
template <int MODE>
__global__ __launch_bounds__(64) void k_resolve(const unsigned *__restrict__ ent, const unsigned *__restrict__ top,
                                                const int *__restrict__ lbeg, const int *__restrict__ lend, int nq_all, int ns, const uint8_t *__restrict__ takes, int th,
                                                float nnratio, int accept_mode, int *__restrict__ acc_sp, int *__restrict__ out_a,
                                                int *__restrict__ nmatches, const int *__restrict__ seg)
{
    const int q_begin = seg ? seg[blockIdx.x] : 0, nq = seg ? seg[blockIdx.x + 1] : nq_all;
    extern __shared__ __align__(16) int sm[];
    int *s_a = sm;            // MODE 0: match_kp[ns]   MODE 1: vMatchedDistance[ns]
    int *s_b = s_a + ns;      // MODE 0: blocked[ns]    MODE 1: vnMatches21[ns]
    int *s_claim = s_b + ns;  // lane that claims the slot in this pass, 64 = none
    int *s_c = s_claim + ns;  // MODE 1: vnMatches12[nq]
    const int lane = threadIdx.x;
    for (int j = lane; j < ns; j += 64) { s_a[j] = MODE == 0 ? -1 : INT_MAX; s_b[j] = MODE == 0 ? 0 : -1; s_claim[j] = 64; }
    if (MODE == 1) for (int j = lane; j < nq; j += 64) s_c[j] = -1;   // MODE 1 never runs in segments
    __syncthreads();
    const bool need2 = MODE == 1 || accept_mode != ACCEPT_BEST;
    int nm = 0;
    // a batch's inputs are fetched while the previous batch is resolved (they do not depend on the state): without
    // this every batch starts with a ~1.5 us round trip to L2 / HBM
    int nb_ = 0, ne_ = 0, ntk = 1;
    uint4 nt0 = make_uint4(~0u, ~0u, ~0u, ~0u), nt1 = nt0;
    auto fetch = [&](int i) {
        nb_ = ne_ = 0; ntk = 1; nt0 = nt1 = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (i < nq) {
            nb_ = lbeg[i]; ne_ = lend[i];
            nt0 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i]; nt1 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i + 1];
            if (MODE == 0) ntk = takes[i];
        }
    };
    fetch(q_begin + lane);
    for (int i0 = q_begin; i0 < nq; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < nq;
        const int b = nb_, e = ne_;
        const uint4 t0 = nt0, t1 = nt1;
        const unsigned tp[TOPK] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const int tk = ntk;
        fetch(i + 64);
        unsigned long long pending = __ballot(in);
        bool dirty = true;
        int sp1 = -1, sp2 = -1, best = INT_MAX, second = INT_MAX, l1 = -1, l2 = -1;
        while (pending) {
            const bool mine = (pending >> lane) & 1ull;
            if (mine && dirty) {
                sp1 = sp2 = -1; best = second = INT_MAX; l1 = l2 = -1;
                int found = 0;
                // the state of all TOPK short-list slots is read at once (one after the other, each behind the test of the one
                // before, they were up to eight dependent LDS round trips per pass on a workgroup of ONE wave)
                int st[TOPK];
#pragma unroll
                for (int r = 0; r < TOPK; ++r) st[r] = MODE == 0 ? s_b[tp[r] == 0xffffffffu ? 0 : (tp[r] & 0xffffu)] : s_a[tp[r] == 0xffffffffu ? 0 : (tp[r] & 0xffffu)];
#pragma unroll
                for (int r = 0; r < TOPK; ++r) {
                    const unsigned en = tp[r];
                    if (en != 0xffffffffu && found < 2) {
                        const int sp = en & 0xffffu, dist = (int)(en >> 20);
                        if (MODE == 0 ? st[r] == 0 : st[r] > dist) {
                            if (found == 0) { sp1 = sp; best = dist; l1 = (en >> 16) & 15; }
                            else { sp2 = sp; second = dist; l2 = (en >> 16) & 15; }
                            ++found;
                        }
                    }
                }
                bool scan = found < (need2 ? 2 : 1) && e - b > TOPK;   // the short list ran dry: the whole list ...
                if (scan) {    // ... unless it cannot change the decision (see k_resolve_init_par / k_resolve_par)
                    const int d7 = (int)(tp[TOPK - 1] >> 20);
                    const bool passes = MODE == 1 ? (float)best < (float)d7 * nnratio
                                                  : (accept_mode == ACCEPT_RATIO ? (float)best < nnratio * (float)d7 : !((float)best > nnratio * (float)d7));
                    if (found == 0) scan = d7 <= th;
                    else if (best > th || passes) { scan = false; second = d7; sp2 = -1; l2 = -2; }
                }
                if (scan) {
                    unsigned k1 = 0xffffffffu, k2 = 0xffffffffu;
                    for (int k = b; k < e; ++k) {
                        const unsigned en = ent[k];
                        const int sp = en & 0xffffu, dist = (int)(en >> 20);
                        if (en != 0xffffffffu && (MODE == 0 ? s_b[sp] == 0 : s_a[sp] > dist)) {
                            const unsigned key = ((unsigned)dist << 16) | (unsigned)(k - b);
                            const unsigned hi = max(k1, key);
                            k2 = min(k2, hi);
                            k1 = min(k1, key);
                        }
                    }
                    sp1 = sp2 = -1; best = second = INT_MAX; l1 = l2 = -1;
                    if (k1 != 0xffffffffu) { const unsigned en = ent[b + (k1 & 0xffffu)]; sp1 = en & 0xffffu; l1 = (en >> 16) & 15; best = (int)(k1 >> 16); }
                    if (k2 != 0xffffffffu) { const unsigned en = ent[b + (k2 & 0xffffu)]; sp2 = en & 0xffffu; l2 = (en >> 16) & 15; second = (int)(k2 >> 16); }
                }
                if (!need2) sp2 = -1;
            }
            bool acc = mine && sp1 >= 0 && best <= th;
            if (MODE == 0) {
                // initial bestDist2 = 256 when there is no second candidate (:79-81)
                const int sec = sp2 >= 0 ? second : 256;
                if (accept_mode == ACCEPT_RATIO_SAME_LEVEL && l1 == l2 && (float)best > nnratio * (float)sec) acc = false; // :121
                if (accept_mode == ACCEPT_RATIO && !((float)best < nnratio * (float)sec)) acc = false;                    // :431, :801
            } else {
                acc = acc && (float)best < (float)second * nnratio; // INT_MAX when alone (:637-638,674-676)
            }
            // claims of this pass
            const bool claims = acc && tk;
            if (claims) atomicMin(&s_claim[sp1], lane);
            __syncthreads();
            const bool conflict = mine && ((sp1 >= 0 && s_claim[sp1] < lane) || (sp2 >= 0 && s_claim[sp2] < lane));
            __syncthreads();
            if (claims) s_claim[sp1] = 64;
            const unsigned long long cm = __ballot(conflict);
            const int lc = cm ? __ffsll((long long)cm) - 1 : 64;
            const bool commit = mine && lane < lc;
            int old = -1;
            if (commit) {
                acc_sp[i] = acc ? sp1 : -1;
                if (acc) {
                    if (MODE == 0) {
                        atomicMax(&s_a[sp1], i);               // the last query assigned to the slot stays
                        if (tk) s_b[sp1] = 1;
                    } else {                                   // accepted candidates of one pass are distinct
                        old = s_b[sp1];
                        if (old >= 0) s_c[old] = -1;           // steal (:678-682)
                        s_c[i] = sp1; s_b[sp1] = i; s_a[sp1] = best;
                    }
                }
            }
            nm += __popcll(__ballot(commit && acc)) - __popcll(__ballot(old >= 0));
            pending &= ~((lc == 64 ? ~0ull : (1ull << lc) - 1ull));
            dirty = conflict;
            __syncthreads();
        }
    }
    __syncthreads();
    if (MODE == 0) {
        for (int j = lane; j < ns; j += 64)                                  // match_kp by sorted position
            if (!seg || s_a[j] >= 0) out_a[j] = s_a[j];
    } else {
        for (int j = lane; j < nq; j += 64) out_a[j] = s_c[j];              // vnMatches12 by sorted position
    }
    if (lane == 0) { if (seg) atomicAdd(nmatches, nm); else *nmatches = nm; }
}

// The projection family's sequential loop as a parallel fixed point, with the rotation check of k_rotation<0> as its tail: ONE
// workgroup of 1024 threads, every query evaluated in every iteration.
//
// In the reference's loop (ORBmatcher.cc:69-130, :1588-1660, ...) query i skips the keypoints an EARLIER query j < i was matched
// to (if j's point blocks the slot: `takes`), and nothing else couples the queries.  Let first[s] be the smallest index of a
// query whose accepted match takes slot s.  Given `first`, query i's selection is a function of it alone: candidates with
// first[s] < i are skipped.  Iterate: all queries select against first_k, their claims give first_{k+1}.  Query 0 never skips
// anything, so it is final after one iteration; once every j < i is final, i is final one iteration later: the iteration reaches
// the sequential loop's result after at most as many iterations as the longest chain "j's match changes what i selects", and a
// state whose claims reproduce themselves is that result (by the same induction).  Real frames have chains of a few queries
// (a window holds a handful of keypoints); the one-wave batch walk of k_resolve took 32 dependent batches for 2,000 queries.
// If the claims still change after RES_MAXIT iterations the kernel says so (*converged = 0) and the host repeats the call on
// k_resolve: the result never depends on how the chains fall.
constexpr int RES_T = 1024, RES_QREG = 2, RES_MAXIT = 48;
__global__ __launch_bounds__(RES_T) void k_resolve_par(const unsigned *__restrict__ ent, const unsigned *__restrict__ top,
                                                       const int *__restrict__ lbeg, const int *__restrict__ lend, int nq, int ns,
                                                       const uint8_t *__restrict__ takes, int th, float nnratio, int accept_mode,
                                                       const float *__restrict__ qangle, const float *__restrict__ kangle,
                                                       const int *__restrict__ perm, int check, int *__restrict__ match_q,
                                                       int *__restrict__ match_kp, int n, const int *__restrict__ flags, int gen,
                                                       int *__restrict__ host_q, int *__restrict__ host_kp, int *__restrict__ host_nm)
{
    // host_*: the call's result block in PINNED HOST memory.  The kernel's last phase writes the results there itself (posted
    // writes over PCIe, each element once): a device-to-host copy behind the kernel cost the call its 5 us plus ~9 us of
    // hand-over between the compute queue and the copy engine.  match_q / match_kp stay the device-side working copies.
    extern __shared__ __align__(16) int sm[];
    int *f_cur = sm, *f_nxt = sm + ns;     // first[s] of the last / of this iteration (INT_MAX: nobody takes s)
    int *s_perm = sm + 2 * ns;             // what the tail gathers by sorted position sits in LDS before the iterations end:
    float *s_kang = reinterpret_cast<float *>(sm + 3 * ns);   // the kernel is one workgroup of dependent round trips
    __shared__ int s_changed, s_nm, hist[HISTO_LENGTH], keep[3], removed;
    const int tid = threadIdx.x;
    const bool need2 = accept_mode != ACCEPT_BEST;
    // everything that does not depend on the iterations is requested up front: the first RES_QREG queries of a thread keep
    // their short lists (and their angles) in registers, the permutation and the keypoint angles go to LDS, every slot of the
    // result is preset (keypoints outside the grid included: no fill by the host)
    unsigned tpr[RES_QREG][TOPK];
    int br[RES_QREG], er[RES_QREG], tkr[RES_QREG];
    float qar[RES_QREG];
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) {
        const int i = tid + r * RES_T;
        br[r] = er[r] = 0; tkr[r] = 1; qar[r] = 0.f;
#pragma unroll
        for (int k = 0; k < TOPK; ++k) tpr[r][k] = 0xffffffffu;
        if (i < nq) {
            const uint4 t0 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i], t1 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i + 1];
            tpr[r][0] = t0.x; tpr[r][1] = t0.y; tpr[r][2] = t0.z; tpr[r][3] = t0.w;
            tpr[r][4] = t1.x; tpr[r][5] = t1.y; tpr[r][6] = t1.z; tpr[r][7] = t1.w;
            br[r] = lbeg[i]; er[r] = lend[i]; tkr[r] = takes[i]; qar[r] = qangle[i];
        }
    }
    for (int j = tid; j < ns; j += RES_T) { f_cur[j] = INT_MAX; f_nxt[j] = INT_MAX; s_perm[j] = perm[j]; s_kang[j] = kangle[j]; }
    for (int j = tid; j < n; j += RES_T) match_kp[j] = -1;
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) match_q[i] = -2;   // queries beyond the registers keep their last selection here
    if (tid == 0) { s_changed = 0; s_nm = 0; removed = 0; }
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    // query i's accepted candidate against `first` (or -1): the selection of k_resolve<0>, eligibility = first[sp] >= i
    auto select = [&](int i, const unsigned (&tp)[TOPK], int b, int e, const int *first) -> int {
        int sp1 = -1, sp2 = -1, best = INT_MAX, second = INT_MAX, l1 = -1, l2 = -1, found = 0;
        int st[TOPK];
#pragma unroll
        for (int r = 0; r < TOPK; ++r) st[r] = first[tp[r] == 0xffffffffu ? 0 : (tp[r] & 0xffffu)];
        if (!need2) {       // "best only" (wave-uniform): the first eligible entry of the sorted short list, a select chain
            unsigned en1 = 0xffffffffu;
#pragma unroll
            for (int r = TOPK - 1; r >= 0; --r) en1 = (tp[r] != 0xffffffffu && st[r] >= i) ? tp[r] : en1;
            if (en1 != 0xffffffffu || e - b <= TOPK) return en1 != 0xffffffffu && (int)(en1 >> 20) <= th ? (int)(en1 & 0xffffu) : -1;
        }
#pragma unroll
        for (int r = 0; r < TOPK; ++r) {
            const unsigned en = tp[r];
            if (en != 0xffffffffu && found < 2 && st[r] >= i) {
                const int sp = en & 0xffffu, dist = (int)(en >> 20);
                if (found == 0) { sp1 = sp; best = dist; l1 = (en >> 16) & 15; }
                else { sp2 = sp; second = dist; l2 = (en >> 16) & 15; }
                ++found;
            }
        }
        // the short list ran dry: the whole list -- unless it cannot change the decision (every entry beyond the short list is at
        // least as far as its last one, d7: nothing eligible so far and d7 > th rejects; one eligible entry that passes its ratio
        // test against d7, a lower bound of the second best (256 at most, the reference's initial bestDist2), passes)
        bool scan = found < (need2 ? 2 : 1) && e - b > TOPK;
        if (scan) {
            const int d7 = (int)(tp[TOPK - 1] >> 20);
            if (found == 0) scan = d7 <= th;
            else if (best > th || (accept_mode == ACCEPT_RATIO ? (float)best < nnratio * (float)d7 : !((float)best > nnratio * (float)d7))) {
                scan = false; sp2 = -1; second = d7; l2 = -2;     // (l2 = -2: never equal to l1, and the bound passes the same-level test anyway)
            }
        }
        if (scan) {
            unsigned k1 = 0xffffffffu, k2 = 0xffffffffu;
            for (int k = b; k < e; ++k) {
                const unsigned en = ent[k];
                if (en != 0xffffffffu && first[en & 0xffffu] >= i) {
                    const unsigned key = ((en >> 20) << 16) | (unsigned)(k - b);
                    const unsigned hi = max(k1, key);
                    k2 = min(k2, hi);
                    k1 = min(k1, key);
                }
            }
            sp1 = sp2 = -1; best = second = INT_MAX; l1 = l2 = -1;
            if (k1 != 0xffffffffu) { const unsigned en = ent[b + (k1 & 0xffffu)]; sp1 = en & 0xffffu; l1 = (en >> 16) & 15; best = (int)(k1 >> 16); }
            if (k2 != 0xffffffffu) { const unsigned en = ent[b + (k2 & 0xffffu)]; sp2 = en & 0xffffu; l2 = (en >> 16) & 15; second = (int)(k2 >> 16); }
        }
        if (!need2) sp2 = -1;
        bool acc = sp1 >= 0 && best <= th;
        const int sec = sp2 >= 0 ? second : 256;     // initial bestDist2 = 256 when there is no second candidate (:79-81)
        if (accept_mode == ACCEPT_RATIO_SAME_LEVEL && l1 == l2 && (float)best > nnratio * (float)sec) acc = false; // :121
        if (accept_mode == ACCEPT_RATIO && !((float)best < nnratio * (float)sec)) acc = false;                    // :431, :801
        return acc ? sp1 : -1;
    };
    auto select_global = [&](int i, const int *first) -> int {
        unsigned tp[TOPK];
        const uint4 t0 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i], t1 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i + 1];
        tp[0] = t0.x; tp[1] = t0.y; tp[2] = t0.z; tp[3] = t0.w; tp[4] = t1.x; tp[5] = t1.y; tp[6] = t1.z; tp[7] = t1.w;
        return select(i, tp, lbeg[i], lend[i], first);
    };
    __syncthreads();
    int selr[RES_QREG];
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) selr[r] = -2;
    bool done = false;
    int it = 0;
    for (; it < RES_MAXIT; ++it) {
        bool changed = false;
#pragma unroll
        for (int r = 0; r < RES_QREG; ++r) {
            const int i = tid + r * RES_T;
            if (i < nq) {
                const int sel = select(i, tpr[r], br[r], er[r], f_cur);
                const int claim = sel >= 0 && tkr[r] ? sel : -1, was = selr[r] >= 0 && tkr[r] ? selr[r] : -1;
                changed |= claim != was || (selr[r] == -2);
                selr[r] = sel;
                if (claim >= 0) atomicMin(&f_nxt[claim], i);
            }
        }
        for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
            const int sel = select_global(i, f_cur), prev = match_q[i], tk = takes[i];
            const int claim = sel >= 0 && tk ? sel : -1, was = prev >= 0 && tk ? prev : -1;
            changed |= claim != was || prev == -2;
            match_q[i] = sel;
            if (claim >= 0) atomicMin(&f_nxt[claim], i);
        }
        if (changed) s_changed = 1;
        __syncthreads();
        done = s_changed == 0;
        __syncthreads();
        if (done) break;
        if (tid == 0) s_changed = 0;
        for (int j = tid; j < ns; j += RES_T) f_cur[j] = INT_MAX;     // becomes the next iteration's claim table
        int *t = f_cur; f_cur = f_nxt; f_nxt = t;
        __syncthreads();
    }
    if (tid == 0) host_nm[1] = flags[1];      // "a list outgrew its region" as k_win_wave left it
    if (!done) {            // chains longer than RES_MAXIT: the host repeats the call on the sequential kernel
        if (tid == 0) host_nm[2] = 0;
        return;
    }
    if (tid == 0) { host_nm[2] = gen; host_nm[3] = it + 1; }   // reached the fixed point (+ the number of iterations it took)
    // ---- tail: match_kp = the LAST accepted query of a slot (:125), the rotation histogram, ComputeThreeMaxima, rejection
    int *s_a = f_nxt;       // (f_nxt holds this iteration's claims = f_cur's content: no longer needed)
    for (int j = tid; j < ns; j += RES_T) s_a[j] = -1;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    auto bin_of = [&](float qa, int sp) {
        float rot = qa - s_kang[sp];
        if (rot < 0.0f) rot += 360.0f;
        const int bin = (int)roundf(rot * factor);
        return bin == HISTO_LENGTH ? 0 : bin;
    };
    int binr[RES_QREG], nacc = 0;
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) {
        const int i = tid + r * RES_T;
        binr[r] = -1;
        if (i < nq && selr[r] >= 0) {
            atomicMax(&s_a[selr[r]], i);
            ++nacc;
            if (check) { binr[r] = bin_of(qar[r], selr[r]); atomicAdd(&hist[binr[r]], 1); }
        }
    }
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
        const int sel = match_q[i];
        if (sel >= 0) {
            atomicMax(&s_a[sel], i);
            ++nacc;
            if (check) atomicAdd(&hist[bin_of(qangle[i], sel)], 1);
        }
    }
    if (nacc) atomicAdd(&s_nm, nacc);
    __syncthreads();
    if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
    }
    for (int j = tid; j < ns; j += RES_T)
        if (s_a[j] >= 0) match_kp[s_perm[j]] = s_a[j];        // (the other slots keep their preset -1)
    __syncthreads();
    auto finish = [&](int i, int sp, int bin) {
        if (check && sp >= 0 && bin != keep[0] && bin != keep[1] && bin != keep[2]) { match_kp[s_perm[sp]] = -2; atomicAdd(&removed, 1); }
        host_q[i] = sp >= 0 ? s_perm[sp] : -1;
    };
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r)
        if (tid + r * RES_T < nq) finish(tid + r * RES_T, selr[r], binr[r]);
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
        const int sp = match_q[i];
        finish(i, sp, check && sp >= 0 ? bin_of(qangle[i], sp) : -1);
    }
    __syncthreads();
    for (int j = tid; j < n; j += RES_T) host_kp[j] = match_kp[j];
    if (tid == 0) host_nm[0] = s_nm - removed;
}

// SearchForInitialization's loop (ORBmatcher.cc:626-696) as a parallel fixed point.  What couples its queries is
// vMatchedDistance: query i leaves a candidate s out of BOTH its best and its second best when an EARLIER query that accepted s
// did so at a distance <= dist(i, s) (:645-646); a later acceptor takes the keypoint from the earlier one (:678-682), and since
// it can only accept s at a distance below the recorded one, vMatchedDistance[s] as query i sees it is M_i(s) = min { d_j : j < i,
// query j accepted s }.  Given every query's accepted (slot, distance), every query's selection is a function of those alone;
// iterate from "nobody has accepted anything": query 0 is final after one iteration, and once all j < i are final so is i one
// iteration later -- a state that reproduces itself is the sequential loop's (the induction of k_resolve_par).  The acceptors
// of a slot sit in a per-slot list of C entries (query << 9 | distance), rebuilt every iteration; a slot with more acceptors
// than that, or chains longer than RES_MAXIT, end the kernel unconverged and the host repeats the call on k_resolve<1>.
// Tail = k_rotation<1>: the match of a slot is its LAST acceptor (the steals), the histogram counts every acceptor, stolen
// or not (:684-693: rotHist keeps them), only standing matches are removed (:704-708).
__global__ __launch_bounds__(RES_T) void k_resolve_init_par(const unsigned *__restrict__ ent, const unsigned *__restrict__ top,
                                                            const int *__restrict__ lbeg, const int *__restrict__ lend, int nq, int ns,
                                                            int C, int th, float nnratio, const float *__restrict__ qangle,
                                                            const float *__restrict__ kangle, const int *__restrict__ perm, int check,
                                                            unsigned *__restrict__ sel_g, const int *__restrict__ flags, int gen,
                                                            int *__restrict__ host_q, int *__restrict__ host_nm)
{
    extern __shared__ __align__(16) int sm[];
    int *s_cnt = sm, *s_lst = sm + ns;                 // acceptors of slot s: s_lst[s * C .. + min(s_cnt[s], C))
    int *s_perm = sm + (size_t)ns * (C + 1);
    float *s_kang = reinterpret_cast<float *>(sm + (size_t)ns * (C + 2));
    __shared__ int s_changed, s_over, s_nm, hist[HISTO_LENGTH], keep[3], removed;
    const int tid = threadIdx.x;
    unsigned tpr[RES_QREG][TOPK];
    int br[RES_QREG], er[RES_QREG];
    float qar[RES_QREG];
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) {
        const int i = tid + r * RES_T;
        br[r] = er[r] = 0; qar[r] = 0.f;
#pragma unroll
        for (int k = 0; k < TOPK; ++k) tpr[r][k] = 0xffffffffu;
        if (i < nq) {
            const uint4 t0 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i], t1 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i + 1];
            tpr[r][0] = t0.x; tpr[r][1] = t0.y; tpr[r][2] = t0.z; tpr[r][3] = t0.w;
            tpr[r][4] = t1.x; tpr[r][5] = t1.y; tpr[r][6] = t1.z; tpr[r][7] = t1.w;
            br[r] = lbeg[i]; er[r] = lend[i]; qar[r] = qangle[i];
        }
    }
    for (int j = tid; j < ns; j += RES_T) { s_cnt[j] = 0; s_perm[j] = perm[j]; s_kang[j] = kangle[j]; }
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) sel_g[i] = 0xfffffffeu;   // queries beyond the registers keep their selection here
    if (tid == 0) { s_changed = 0; s_over = 0; s_nm = 0; removed = 0; }
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    // vMatchedDistance[sp] as query i sees it
    auto matched_before = [&](int sp, int i) -> int {
        const int c = min(s_cnt[sp], C);
        int m = INT_MAX;
        for (int k = 0; k < c; ++k) {
            const int v = s_lst[sp * C + k];
            if ((v >> 9) < i) m = min(m, v & 511);
        }
        return m;
    };
    // query i's accepted entry (dist << 20 | octave << 16 | sp) or 0xffffffff: the selection of k_resolve<1>
    auto select = [&](int i, const unsigned (&tp)[TOPK], int b, int e) -> unsigned {
        unsigned en1 = 0xffffffffu;
        int best = INT_MAX, second = INT_MAX, found = 0;
#pragma unroll
        for (int r = 0; r < TOPK; ++r) {
            const unsigned en = tp[r];
            if (en != 0xffffffffu && found < 2) {
                const int dist = (int)(en >> 20);
                if (matched_before((int)(en & 0xffffu), i) > dist) {
                    if (found == 0) { en1 = en; best = dist; } else second = dist;
                    ++found;
                }
            }
        }
        // The short list ran dry -- but the rest of the list (every entry's key is above the short list's last) is needed only if it
        // can change the DECISION: with no eligible entry so far, a best beyond the last short-list distance d7 must still be <= th;
        // with one, the second best is >= d7, so best < d7 * ratio already passes the ratio test (:674-676) whatever it is.  Most of a
        // late query's short list is taken by earlier matches at small distances (:645-646), so this case is the common one.
        bool scan = found < 2 && e - b > TOPK;
        if (scan) {
            const int d7 = (int)(tp[TOPK - 1] >> 20);
            if (found == 0) scan = d7 <= th;
            else if (best > th || (float)best < (float)d7 * nnratio) { scan = false; second = d7; }
        }
        if (scan) {
            unsigned k1 = 0xffffffffu, k2 = 0xffffffffu;
            for (int k = b; k < e; ++k) {
                const unsigned en = ent[k];
                if (en != 0xffffffffu && matched_before((int)(en & 0xffffu), i) > (int)(en >> 20)) {
                    const unsigned key = ((en >> 20) << 16) | (unsigned)(k - b);
                    const unsigned hi = max(k1, key);
                    k2 = min(k2, hi);
                    k1 = min(k1, key);
                }
            }
            en1 = 0xffffffffu; best = second = INT_MAX;
            if (k1 != 0xffffffffu) { en1 = ent[b + (k1 & 0xffffu)]; best = (int)(k1 >> 16); }
            if (k2 != 0xffffffffu) second = (int)(k2 >> 16);
        }
        const bool acc = en1 != 0xffffffffu && best <= th && (float)best < (float)second * nnratio;   // INT_MAX when alone (:637-638, :674-676)
        return acc ? en1 : 0xffffffffu;
    };
    auto select_global = [&](int i) -> unsigned {
        unsigned tp[TOPK];
        const uint4 t0 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i], t1 = reinterpret_cast<const uint4 *>(top)[2 * (size_t)i + 1];
        tp[0] = t0.x; tp[1] = t0.y; tp[2] = t0.z; tp[3] = t0.w; tp[4] = t1.x; tp[5] = t1.y; tp[6] = t1.z; tp[7] = t1.w;
        return select(i, tp, lbeg[i], lend[i]);
    };
    auto claim = [&](int i, unsigned en) {
        if (en == 0xffffffffu) return;
        const int sp = (int)(en & 0xffffu), k = atomicAdd(&s_cnt[sp], 1);
        if (k < C) s_lst[sp * C + k] = (i << 9) | (int)(en >> 20); else s_over = 1;
    };
    __syncthreads();
    unsigned selr[RES_QREG];
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) selr[r] = 0xfffffffeu;
    bool done = false;
    int it = 0;
    for (; it < RES_MAXIT; ++it) {
        bool changed = false;
#pragma unroll
        for (int r = 0; r < RES_QREG; ++r) {
            const int i = tid + r * RES_T;
            if (i < nq) {
                const unsigned en = select(i, tpr[r], br[r], er[r]);
                changed |= en != selr[r];
                selr[r] = en;
            }
        }
        for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
            const unsigned en = select_global(i);
            changed |= en != sel_g[i];
            sel_g[i] = en;
        }
        if (changed) s_changed = 1;
        __syncthreads();
        done = s_changed == 0;
        __syncthreads();
        if (done) break;
        if (tid == 0) s_changed = 0;
        for (int j = tid; j < ns; j += RES_T) s_cnt[j] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RES_QREG; ++r)
            if (tid + r * RES_T < nq) claim(tid + r * RES_T, selr[r]);
        for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) claim(i, sel_g[i]);
        __syncthreads();
        if (s_over) break;
    }
    if (tid == 0) host_nm[1] = flags[1];      // "a list outgrew its region" as k_win_wave left it
    if (!done) {            // a crowded slot or chains longer than RES_MAXIT: the host repeats the call on the sequential kernel
        if (tid == 0) host_nm[2] = 0;
        return;
    }
    if (tid == 0) { host_nm[2] = gen; host_nm[3] = it + 1; }
    // ---- tail: the standing match of a slot is its last acceptor; rotation histogram over all acceptors; rejection
    int *s_last = s_cnt;
    for (int j = tid; j < ns; j += RES_T) s_last[j] = -1;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    auto bin_of = [&](float qa, int sp) {
        float rot = qa - s_kang[sp];
        if (rot < 0.0f) rot += 360.0f;
        const int bin = (int)roundf(rot * factor);
        return bin == HISTO_LENGTH ? 0 : bin;
    };
    int binr[RES_QREG];
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r) {
        const int i = tid + r * RES_T;
        binr[r] = -1;
        if (i < nq && selr[r] != 0xffffffffu) {
            const int sp = (int)(selr[r] & 0xffffu);
            atomicMax(&s_last[sp], i);
            if (check) { binr[r] = bin_of(qar[r], sp); atomicAdd(&hist[binr[r]], 1); }
        }
    }
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
        const unsigned en = sel_g[i];
        if (en != 0xffffffffu) {
            atomicMax(&s_last[(int)(en & 0xffffu)], i);
            if (check) atomicAdd(&hist[bin_of(qangle[i], (int)(en & 0xffffu))], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
    }
    int nst = 0;
    for (int j = tid; j < ns; j += RES_T) nst += s_last[j] >= 0;     // nmatches before the rotation check = slots that hold a match
    if (nst) atomicAdd(&s_nm, nst);
    __syncthreads();
    auto finish = [&](int i, unsigned en, int bin) {
        int out = -1;
        if (en != 0xffffffffu) {
            const int sp = (int)(en & 0xffffu);
            if (s_last[sp] == i) {      // still standing
                out = s_perm[sp];
                if (check && bin != keep[0] && bin != keep[1] && bin != keep[2]) { out = -1; atomicAdd(&removed, 1); }
            }
        }
        host_q[i] = out;
    };
#pragma unroll
    for (int r = 0; r < RES_QREG; ++r)
        if (tid + r * RES_T < nq) finish(tid + r * RES_T, selr[r], binr[r]);
    for (int i = tid + RES_QREG * RES_T; i < nq; i += RES_T) {
        const unsigned en = sel_g[i];
        finish(i, en, check && en != 0xffffffffu ? bin_of(qangle[i], (int)(en & 0xffffu)) : -1);
    }
    __syncthreads();
    if (tid == 0) host_nm[0] = s_nm - removed;
}

// Window search without coupling between queries = Frame::GetFeaturesInArea (src/Frame.cc:342-395)
// fused with the best / second-best-with-levels loop of SearchByProjection (ORBmatcher.cc:69-118).
// One wave per query walks the window's runs of the sorted keypoint array as k_win_wave does; every
// lane keeps the two smallest keys dist << 16 | position-in-visiting-order of its candidates (strict
// '<', first wins = smallest key; bestDist2 = second smallest), two wave reductions pick the winners.
// Outputs: best / second distance (init_dist when absent), their octaves (-1), arg-best as a
// keypoint index (-1).
__global__ __launch_bounds__(MT) void k_win_best(const WinQuery *__restrict__ q, const uint4 *__restrict__ A, int nq,
                                                 const SeqKp *__restrict__ kp, const uint4 *__restrict__ B,
                                                 const int *__restrict__ cell_off, const int *__restrict__ perm, const uint8_t *__restrict__ occ, GridParams gp,
                                                 int has_uright, int init_dist, const float *__restrict__ inv_sigma2, int fuse_gate,
                                                 int *__restrict__ best_o, int *__restrict__ bl_o, int *__restrict__ second_o,
                                                 int *__restrict__ sl_o, int *__restrict__ idx_o)
{
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (MT / 64) + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const WinQuery w = q[i];
    unsigned k1 = 0xffffffffu, k2 = 0xffffffffu, e1 = 0, e2 = 0; // e = octave << 16 | sorted position
    int c = 0;
    const int nMinCellX = (int)fmaxf(0.f, floorf((w.u - gp.min_x - w.r) * gp.inv_w));
    const int nMaxCellX = (int)fminf((float)FRAME_GRID_COLS - 1, ceilf((w.u - gp.min_x + w.r) * gp.inv_w));
    const int nMinCellY = (int)fmaxf(0.f, floorf((w.v - gp.min_y - w.r) * gp.inv_h));
    const int nMaxCellY = (int)fminf((float)FRAME_GRID_ROWS - 1, ceilf((w.v - gp.min_y + w.r) * gp.inv_h));
    const bool none = !(w.r >= 0.f) || nMinCellX >= FRAME_GRID_COLS || nMaxCellX < 0 || nMinCellY >= FRAME_GRID_ROWS || nMaxCellY < 0;
    if (!none) {
        const bool check_levels = (w.min_level > 0) || (w.max_level >= 0);
        const uint4 a0 = A[2 * i], a1 = A[2 * i + 1];
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            const int r0 = cell_off[ix * FRAME_GRID_ROWS + nMinCellY], r1 = cell_off[ix * FRAME_GRID_ROWS + nMaxCellY + 1];
            for (int kb = r0; kb < r1; kb += 64) {
                const int k = kb + lane;
                bool ok = k < r1;
                SeqKp p = {0.f, 0.f, 0.f, 0};
                if (ok) p = kp[k];
                if (occ && ok) ok = occ[k] == 0;
                if (check_levels) ok = ok && !(p.octave < w.min_level) && !(w.max_level >= 0 && p.octave > w.max_level);
                const float distx = p.x - w.u, disty = p.y - w.v;
                ok = ok && fabsf(distx) < w.r && fabsf(disty) < w.r;
                bool use = ok; // in the visiting order (position) every keypoint of the window counts; `use`: it is scored
                if (fuse_gate == 0) {
                    if (has_uright && p.uright > 0) ok = ok && !(fabsf(w.xr - p.uright) > w.r);
                    use = ok;
                } else if (ok && inv_sigma2) { // reprojection-error gate of ORBmatcher::Fuse (:1109-1132)
                    const float ex = w.u - p.x, ey = w.v - p.y;
                    if (has_uright && p.uright >= 0) {
                        const float er = w.xr - p.uright;
                        const float e2 = ex * ex + ey * ey + er * er;
                        use = !((double)(e2 * inv_sigma2[p.octave]) > 7.8);
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        use = !((double)(e2 * inv_sigma2[p.octave]) > 5.99);
                    }
                }
                const unsigned long long bal = __ballot(ok);
                if (use) {
                    const int dist = popc256(a0, a1, B[2 * k], B[2 * k + 1]);
                    if (dist < init_dist) { // dist<bestDist / dist<bestDist2 can only fire below the initial value
                        const unsigned key = ((unsigned)dist << 16) | (unsigned)(c + __popcll(bal & lt));
                        const unsigned en = ((unsigned)p.octave << 16) | (unsigned)k;
                        if (key < k1) { k2 = k1; e2 = e1; k1 = key; e1 = en; }
                        else if (key < k2) { k2 = key; e2 = en; }
                    }
                }
                c += __popcll(bal);
            }
        }
    }
    // smallest key of the wave, then the smallest of what is left (keys are unique)
    const unsigned g1 = wave_min_u32(k1);
    const bool own1 = g1 != 0xffffffffu && k1 == g1;
    const unsigned long long b1 = __ballot(own1);
    const unsigned c2 = own1 ? k2 : k1;
    const unsigned g2 = wave_min_u32(c2);
    const unsigned long long b2 = __ballot(g2 != 0xffffffffu && c2 == g2);
    const unsigned w1 = b1 ? (unsigned)__shfl((int)e1, __ffsll((long long)b1) - 1) : 0u;
    const unsigned w2 = b2 ? (unsigned)__shfl((int)(own1 ? e2 : e1), __ffsll((long long)b2) - 1) : 0u;
    if (lane == 0) {
        best_o[i] = b1 ? (int)(g1 >> 16) : init_dist;
        idx_o[i] = b1 ? perm[w1 & 0xffffu] : -1;
        bl_o[i] = b1 ? (int)(w1 >> 16) : -1;
        second_o[i] = b2 ? (int)(g2 >> 16) : init_dist;
        sl_o[i] = b2 ? (int)(w2 >> 16) : -1;
    }
}

// The fork's whole-map relocalisation search (ORBmatcher.cc:134-222): isInFrustum
// (:262-330) + ComputeDistance (:224-260) per map point in the reference's mixed
// float / double arithmetic (fixed op order, no contraction), producing the
// GetFeaturesInArea query of :162-163; k_win_best does the search; then the
// TH_RELOC / same-level ratio acceptance (:205-216) with "last map point wins".
// The upload of a call's staged inputs (orbx::stage_in: pinned host memory read by the compute queue) folded into the call's first
// kernel: blocks [0, nblocks) of the launch copy 16 bytes per thread, the others do the kernel's own work and read THEIR inputs --
// which nothing else needs on the device -- straight from the pinned block.  One launch less per call.
struct StageJob { const uint4 *src; uint4 *dst; size_t n16; int nblocks; };
__device__ __forceinline__ bool stage_block(const StageJob &job)
{
    if ((int)blockIdx.x >= job.nblocks) return false;
    const size_t i = (size_t)blockIdx.x * MT + threadIdx.x;
    if (i < job.n16) job.dst[i] = job.src[i];
    return true;
}
inline StageJob stage_job(void *dev, const void *pin, size_t bytes)
{
    return {static_cast<const uint4 *>(pin), static_cast<uint4 *>(dev), bytes / 16, (int)((bytes / 16 + MT - 1) / MT)};
}

struct MapCam { float fx, fy, cx, cy; int bminx, bmaxx, bminy, bmaxy; double R[9], t[3]; float th; int nlevels; };

__global__ __launch_bounds__(MT) void k_map_frustum(const float *__restrict__ pos, const float *__restrict__ nrm,
                                                    const float *__restrict__ mind, const float *__restrict__ maxd, int m,
                                                    MapCam cam, const float *__restrict__ scale, WinQuery *__restrict__ q,
                                                    float *__restrict__ proj)
{
    const int i = blockIdx.x * MT + threadIdx.x;
    if (i >= m) return;
    WinQuery w = {0.f, 0.f, -1.f, 0.f, 0, -1}; // r < 0: no candidates
    float out[4] = {0.f, 0.f, 0.f, -1.f};
    const float ptX = pos[3 * i], ptY = pos[3 * i + 1], ptZ = pos[3 * i + 2];
    const double *R = cam.R, *t = cam.t;
    const float PcX = (float)(R[0] * ptX + R[1] * ptY + R[2] * ptZ + t[0]);
    const float PcY = (float)(R[3] * ptX + R[4] * ptY + R[5] * ptZ + t[1]);
    const float PcZ = (float)(R[6] * ptX + R[7] * ptY + R[8] * ptZ + t[2]);
    bool ok = !(PcZ < 0.0f);
    const float invz = (float)(1.0 / (double)PcZ);
    const float u = cam.fx * PcX * invz + cam.cx;
    const float v = cam.fy * PcY * invz + cam.cy;
    ok = ok && !(u < (float)cam.bminx || u > (float)cam.bmaxx) && !(v < (float)cam.bminy || v > (float)cam.bmaxy);
    // ComputeDistance: PO = Pt - (-R^T) t, norm in double
    double PO[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double rtt = ((-1) * R[a]) * t[0] + ((-1) * R[3 + a]) * t[1] + ((-1) * R[6 + a]) * t[2];
        PO[a] = (double)(a == 0 ? ptX : a == 1 ? ptY : ptZ) - rtt;
    }
    const double normSum = PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2];
    const float dist = (float)sqrt(normSum);
    const float minDistance = mind[i], maxDistance = maxd[i];
    ok = ok && !((double)dist < (0.9 * (double)minDistance) || (double)dist > ((double)maxDistance / 0.9));
    float viewCos = (float)(PO[0] * nrm[3 * i] + PO[1] * nrm[3 * i + 1] + PO[2] * nrm[3 * i + 2]);
    viewCos = viewCos / dist;
    ok = ok && !(viewCos < 0.5f);
    const float ratio = dist / minDistance;
    int level = 0;
    while (level < cam.nlevels && scale[level] < ratio) ++level; // lower_bound(mvScaleFactors, ratio)
    if (level >= cam.nlevels) level = cam.nlevels - 1;
    if (ok) {
        float r = (double)viewCos > 0.998 ? 3.0f : 4.5f; // RadiusByViewingCos
        if ((double)cam.th != 1.0) r *= cam.th;
        w.u = u; w.v = v; w.r = r * scale[level]; w.min_level = level - 1; w.max_level = level;
        out[0] = u; out[1] = v; out[2] = viewCos; out[3] = (float)level;
    }
    q[i] = w;
    if (proj) { proj[4 * i] = out[0]; proj[4 * i + 1] = out[1]; proj[4 * i + 2] = out[2]; proj[4 * i + 3] = out[3]; }
}

// Batch projection of map points into a Frame / KeyFrame, in the reference's float / double order, op by op:
//   mode 0  Frame::isInFrustum (src/Frame.cc:284-340) + MapPoint::PredictScale (src/MapPoint.cc:464-480) and the window
//           of SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:62-70, RadiusByViewingCos :332-338)
//   mode 1  the projection block of ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>&, th) (:1053-1094)
//   mode 2  the same block of the Sim3 form (:1212-1250): invz = 1.0 / z in double, no right coordinate
// cv::Mat arithmetic restated (OpenCV 3.4, CV_32F; parity unpinned like the other OpenCV primitives):
//   Rcw * P + tcw  : gemm's 3 x 3 special case -- the row sum a0 b0 + a1 b1 + a2 b2 in float, left to right, then
//                    float(double(sum) + double(t));
//   cv::norm(PO)   : sqrt of the double sum of squares, left to right, cast to float;
//   PO.dot(Pn)     : double sum of double products, left to right.
// std::log / std::ceil on a float argument are the float overloads (using namespace std): logf as glibc >= 2.27 evaluates it,
// restated in orbx_math.h (orbx_logf_glibc_f32: equal to the library at every positive float, tools/trig/logf_count.c; it differs
// from the rounded double logarithm at 416,909 floats -- at none of them by enough to move a level at scale factor 1.2).
struct ProjectCam { float fx, fy, cx, cy, min_x, max_x, min_y, max_y, mbf, cos_limit, log_scale, th; float R[9], t[3], Ow[3]; int nlevels, mode; };

__global__ __launch_bounds__(MT) void k_project_points(const uint8_t *__restrict__ valid, const float *__restrict__ pos, const float *__restrict__ nrm,
                                                       const float *__restrict__ mind, const float *__restrict__ maxd, int m,
                                                       ProjectCam cam, const float *__restrict__ scale,
                                                       orbm_projected_point *__restrict__ out, WinQuery *__restrict__ q, StageJob job)
{
    if (stage_block(job)) return;
    const int i = (blockIdx.x - job.nblocks) * MT + threadIdx.x;
    if (i >= m) return;
    orbm_projected_point o = {0.f, 0.f, 0.f, 0.f, 0.f, -1, 0};
    WinQuery w = {0.f, 0.f, -1.f, 0.f, 0, -1}; // r < 0: no candidates
    const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
    const float *R = cam.R;
    const float PcX = (float)((double)(R[0] * P0 + R[1] * P1 + R[2] * P2) + (double)cam.t[0]);
    const float PcY = (float)((double)(R[3] * P0 + R[4] * P1 + R[5] * P2) + (double)cam.t[1]);
    const float PcZ = (float)((double)(R[6] * P0 + R[7] * P1 + R[8] * P2) + (double)cam.t[2]);
    bool ok = !(PcZ < 0.0f) && (!valid || valid[i]);
    float invz, u, v;
    if (cam.mode == 0) {
        invz = 1.0f / PcZ;                                  // Frame.cc:302-304
        u = cam.fx * PcX * invz + cam.cx;
        v = cam.fy * PcY * invz + cam.cy;
        ok = ok && !(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y);
    } else {
        invz = cam.mode == 1 ? 1 / PcZ : (float)(1.0 / (double)PcZ);   // ORBmatcher.cc:1060 / :1222
        const float x = PcX * invz, y = PcY * invz;
        u = cam.fx * x + cam.cx;
        v = cam.fy * y + cam.cy;
        ok = ok && (u >= cam.min_x && u < cam.max_x && v >= cam.min_y && v < cam.max_y); // KeyFrame::IsInImage
    }
    const float ur = u - cam.mbf * invz;
    const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i]; // MapPoint.cc:430-440
    const float PO0 = P0 - cam.Ow[0], PO1 = P1 - cam.Ow[1], PO2 = P2 - cam.Ow[2];
    const float dist = (float)sqrt((double)PO0 * (double)PO0 + (double)PO1 * (double)PO1 + (double)PO2 * (double)PO2);
    const double dot = (double)PO0 * (double)nrm[3 * i] + (double)PO1 * (double)nrm[3 * i + 1] + (double)PO2 * (double)nrm[3 * i + 2];
    float viewCos = 0.f;
    if (cam.mode == 0) {
        ok = ok && !((double)dist < 0.9 * (double)minDistance || (double)dist > (double)maxDistance / 0.9);
        viewCos = (float)(dot / (double)dist);
        ok = ok && !(viewCos < cam.cos_limit);
    } else {
        ok = ok && !(dist < minDistance || dist > maxDistance);
        ok = ok && !(dot < 0.5 * (double)dist);
    }
    // PredictScale: ratio = mfMaxDistance / dist; ceil(log(ratio) / mfLogScaleFactor), clamped
    const float ratio = maxd[i] / dist;
    const float lg = orbx_logf_glibc_f32(ratio);        // log(float) = logf under the reference's headers; glibc's logf restated (orbx_math.h)
    int level = (int)ceilf(lg / cam.log_scale);
    if (level < 0) level = 0; else if (level >= cam.nlevels) level = cam.nlevels - 1;
    if (ok) {
        o.u = u; o.v = v; o.ur = ur; o.view_cos = viewCos; o.dist = dist; o.level = level; o.visible = 1;
        float r;
        if (cam.mode == 0) {
            r = (double)viewCos > 0.998 ? 3.0f : 4.5f;      // RadiusByViewingCos
            if ((double)cam.th != 1.0) r *= cam.th;
        } else {
            r = cam.th;
        }
        w.u = u; w.v = v; w.r = r * scale[level]; w.xr = ur; w.min_level = level - 1; w.max_level = level;
    }
    out[i] = o;
    if (q) q[i] = w;
}

__global__ __launch_bounds__(MT) void k_reloc_accept(const int *__restrict__ best, const int *__restrict__ bidx,
                                                     const int *__restrict__ second, const int *__restrict__ blevel,
                                                     const int *__restrict__ slevel, int m, int th_reloc, float nnratio,
                                                     int *__restrict__ matched, int *__restrict__ nmatches)
{
    const int i = blockIdx.x * MT + threadIdx.x;
    bool acc = false;
    if (i < m && bidx[i] >= 0 && best[i] <= th_reloc) {
        acc = !(blevel[i] == slevel[i] && (float)best[i] > nnratio * (float)second[i]);
        if (acc) atomicMax(&matched[bidx[i]], i); // vMatchedMPs[bestIdx] = pMP: the last map point wins
    }
    const unsigned long long b = __ballot(acc);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(nmatches, __popcll(b));
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
    auto bin_of = [&](int i, int sp) {
        float rot = qangle[i] - kangle[sp];
        if (rot < 0.0f) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        return bin == HISTO_LENGTH ? 0 : bin;
    };
    // the first RPT * MT queries keep (candidate, bin) in registers between the two passes: one dependent gather chain
    // instead of two (the kernel is one block of pure latency)
    constexpr int RPT = 12;
    int spc[RPT], binc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = tid + r * MT;
        spc[r] = i < nq ? acc_sp[i] : -1;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        binc[r] = -1;
        if (check && spc[r] >= 0) { binc[r] = bin_of(tid + r * MT, spc[r]); atomicAdd(&hist[binc[r]], 1); }
    }
    if (check)
        for (int i = tid + RPT * MT; i < nq; i += MT) {
            const int sp = acc_sp[i];
            if (sp >= 0) atomicAdd(&hist[bin_of(i, sp)], 1);
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
    auto finish = [&](int i, int sp, int bin) {
        int out = MODE == 0 ? (sp >= 0 ? perm[sp] : -1) : (state[i] >= 0 ? perm[state[i]] : -1);
        if (check && sp >= 0 && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
            if (MODE == 0) { match_kp[perm[sp]] = -2; atomicAdd(&removed, 1); }   // every entry of the bin: slot cleared, nmatches--
            else if (state[i] >= 0) { out = -1; atomicAdd(&removed, 1); }         // only matches still standing (:704-708)
        }
        match_q[i] = out;
    };
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (tid + r * MT < nq) finish(tid + r * MT, spc[r], binc[r]);
    for (int i = tid + RPT * MT; i < nq; i += MT) {
        const int sp = acc_sp[i];
        finish(i, sp, check && sp >= 0 ? bin_of(i, sp) : -1);
    }
    __syncthreads();
    if (tid == 0) *nmatches -= removed;
}

// ---------------------------------------------------------------------------------------------------------------
// A frame resident in HBM (orbm_frame): Frame::AssignFeaturesToGrid (src/Frame.cc:245-260, PosInGrid :397-407) on the device.
// ONE workgroup lays the keypoints out in (cell, index) order -- the order Frame::GetFeaturesInArea visits a window in --
//   1. cell of every keypoint (the reference's float arithmetic), cell histogram in LDS;
//   2. exclusive scan of the histogram = cell_off;
//   3. rank of a keypoint inside its cell = keypoints of the same cell with a smaller index: per 64-keypoint chunk the
//      lanes compare cells pairwise (all waves), then one wave walks the chunks in order with a running count per cell;
//   4. gather: position, level, right coordinate, angle, descriptor, permutation.
// Keypoints outside the grid (Frame::PosInGrid false: no window search ever returns them) follow the sorted part at positions
// ns .. n-1 in index order: the searches over explicit candidate lists (SearchByBoW) address features by index, grid or not.
// Keypoints come as cv::KeyPoint records (mvKeysUn; or the extractor's device results, optionally with the undistorted
// coordinates beside them), n from the host or from the extractor's count array.
constexpr int FB_T = 1024, FB_MAXN = SEQ_MAXN, FB_NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
struct FrameHdr { int n, ns, min_octave, max_octave; };
constexpr size_t FB_LDS = sizeof(int) * (2 * (size_t)FB_NC + 2) + (size_t)FB_MAXN * (2 + 2 + 1 + 1);

__global__ __launch_bounds__(FB_T) void k_frame_build(const orbx_keypoint *__restrict__ kps, const uint4 *__restrict__ desc_raw,
                                                      const float2 *__restrict__ xy_un, const float *__restrict__ uright_raw,
                                                      const int *__restrict__ n_dev, int n_host, GridParams gp, SeqKp *__restrict__ kp,
                                                      uint4 *__restrict__ desc, float *__restrict__ angle, int *__restrict__ perm,
                                                      int *__restrict__ cell_off, FrameHdr *__restrict__ hdr)
{
    extern __shared__ __align__(16) unsigned char fb_sm[];
    int *s_off = reinterpret_cast<int *>(fb_sm);                 // [FB_NC + 1] histogram, then exclusive scan
    int *s_run = s_off + FB_NC + 1;                              // [FB_NC + 1] keypoints of the cell placed so far (last entry: scan carry)
    unsigned short *s_cell = reinterpret_cast<unsigned short *>(s_run + FB_NC + 1);   // [n] cell or 0xffff
    unsigned short *s_pos = s_cell + FB_MAXN;                    // [n] sorted position
    unsigned char *s_low = reinterpret_cast<unsigned char *>(s_pos + FB_MAXN), *s_tot = s_low + FB_MAXN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(n_dev ? *n_dev : n_host, FB_MAXN);
    __shared__ int s_omin, s_omax;     // octave range of the keypoints (the searches index per-level tables with it)
    if (tid == 0) { s_omin = INT_MAX; s_omax = INT_MIN; }
    for (int c = tid; c <= FB_NC; c += FB_T) { s_off[c] = 0; s_run[c] = 0; }
    __syncthreads();
    for (int j = tid; j < n; j += FB_T) {
        const float x = xy_un ? xy_un[j].x : kps[j].x, y = xy_un ? xy_un[j].y : kps[j].y;
        const int px = (int)roundf((x - gp.min_x) * gp.inv_w), py = (int)roundf((y - gp.min_y) * gp.inv_h);
        const bool in = !(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS);
        const int c = in ? px * FRAME_GRID_ROWS + py : FB_NC;   // FB_NC: outside the grid -- kept behind the sorted part, in index order
        s_cell[j] = (unsigned short)c;
        if (in) atomicAdd(&s_off[c], 1);
    }
    __syncthreads();
    {   // exclusive scan of FB_NC counts: 3 per thread, wave scan, wave totals through s_run's spare slots
        constexpr int PER = FB_NC / FB_T;
        static_assert(FB_NC % FB_T == 0, "cells per thread");
        int v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = s_off[tid * PER + k]; sum += v[k]; }
        const int inc = orbx::wave_incl_scan(sum);
        __shared__ int wsum[FB_T / 64];
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + inc - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) { s_off[tid * PER + k] = run; run += v[k]; }
        if (tid == FB_T - 1) s_off[FB_NC] = run;
    }
    __syncthreads();
    for (int c = tid; c <= FB_NC; c += FB_T) cell_off[c] = s_off[c];
    // rank inside the chunk: lanes of one cell, in lane order
    const int nchunk = (n + 63) / 64;
    for (int ch = wave; ch < nchunk; ch += FB_T / 64) {
        const int j = ch * 64 + lane;
        const int c = j < n ? (int)s_cell[j] : 0xffff;
        int low = 0, tot = 0;
#pragma unroll 8
        for (int l = 0; l < 64; ++l) {
            const int cl = __shfl(c, l);
            const int same = cl == c;
            tot += same;
            low += same && l < lane;
        }
        if (j < n) { s_low[j] = (unsigned char)low; s_tot[j] = (unsigned char)(tot - 1); }   // (tot - 1 <= 63 fits a byte)
    }
    __syncthreads();
    if (wave == 0) {        // chunks in index order: the running count of a cell is what the earlier chunks put there
        for (int ch = 0; ch < nchunk; ++ch) {
            const int j = ch * 64 + lane;
            const int c = j < n ? (int)s_cell[j] : 0xffff;
            int base = 0;
            if (c != 0xffff) base = s_run[c];
            if (c != 0xffff) {
                s_pos[j] = (unsigned short)(s_off[c] + base + s_low[j]);
                if (s_low[j] == s_tot[j]) s_run[c] = base + s_tot[j] + 1;    // the cell's last lane of the chunk
            }
        }
    }
    __syncthreads();
    int omin = INT_MAX, omax = INT_MIN;
    for (int j = tid; j < n; j += FB_T) {
        const int sp = s_pos[j];
        const orbx_keypoint k = kps[j];
        omin = min(omin, k.octave); omax = max(omax, k.octave);
        SeqKp o;
        o.x = xy_un ? xy_un[j].x : k.x; o.y = xy_un ? xy_un[j].y : k.y;
        o.uright = uright_raw ? uright_raw[j] : -1.0f;
        o.octave = k.octave;
        kp[sp] = o; angle[sp] = k.angle; perm[sp] = j;
        desc[2 * sp] = desc_raw[2 * j]; desc[2 * sp + 1] = desc_raw[2 * j + 1];
    }
    if (omin <= omax) { atomicMin(&s_omin, omin); atomicMax(&s_omax, omax); }
    __syncthreads();
    if (tid == 0) { hdr->n = n; hdr->ns = s_off[FB_NC]; hdr->min_octave = s_omin; hdr->max_octave = s_omax; }
}

// ORBmatcher::SearchForTriangulation's gated loop (src/ORBmatcher.cc:892-990 with CheckDistEpipolarLine :341-358) on two resident
// frames: 16 lanes per keypoint of KF1, taken in KF1's sorted order (s -> feature i = perm1[s]); its candidates = the members of its
// vocabulary node in KF2, one copy per node, given as positions in KF2's arrays.  "Stereo" = the frame's right coordinate >= 0
// (mvuRight[idx] >= 0, :911 / :931).  The reference keeps the LAST candidate among those of smallest distance that pass the gates
// (`dist > bestDist` is non-strict): the minimum of dist << 16 | (0xffff - position in the list).  Out, by feature index of KF1:
// match12 = feature index in KF2 or -1; rot = angle1 - angle2 of the pair (:994), for the host's histogram.
struct TriForm { float F12[9]; float ex, ey; int only_stereo; };
__global__ __launch_bounds__(MT) void k_triang_frames(const SeqKp *__restrict__ kp1, const uint4 *__restrict__ A, const float *__restrict__ ang1,
                                                      const int *__restrict__ perm1, int n1, const SeqKp *__restrict__ kp2,
                                                      const uint4 *__restrict__ B, const float *__restrict__ ang2, const int *__restrict__ perm2,
                                                      const int *__restrict__ cbeg, const int *__restrict__ clen, const int *__restrict__ cand,
                                                      const uint8_t *__restrict__ hasmp1, const uint8_t *__restrict__ hasmp2, TriForm tp,
                                                      const float *__restrict__ scale2, const float *__restrict__ sigma2,
                                                      int *__restrict__ match12, float *__restrict__ rot)
{
    const int s = (blockIdx.x * MT + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (s >= n1) return;                       // (whole 16-lane groups leave together)
    const int i = perm1[s];
    const SeqKp k1 = kp1[s];
    const bool st1 = k1.uright >= 0;
    unsigned key = 0xffffffffu;
    int k0 = 0;
    if (!hasmp1[i] && !(tp.only_stereo && !st1)) {
        const uint4 a0 = A[2 * s], a1 = A[2 * s + 1];
        // epipolar line in the second image l = x1' F12 = [a b c]
        const float a = k1.x * tp.F12[0] + k1.y * tp.F12[3] + tp.F12[6];
        const float b = k1.x * tp.F12[1] + k1.y * tp.F12[4] + tp.F12[7];
        const float c = k1.x * tp.F12[2] + k1.y * tp.F12[5] + tp.F12[8];
        const float den = a * a + b * b;
        k0 = cbeg[i];
        const int kend = k0 + clen[i];
        for (int k = k0 + sub; k < kend; k += 16) {
            const int p = cand[k];
            if (hasmp2[perm2[p]]) continue;
            const SeqKp k2 = kp2[p];
            const bool st2 = k2.uright >= 0;
            if (tp.only_stereo && !st2) continue;
            const int dist = popc256(a0, a1, B[2 * p], B[2 * p + 1]);
            if (dist > 45) continue;           // TH_LOW
            if (!st1 && !st2) {
                const float distex = tp.ex - k2.x, distey = tp.ey - k2.y;
                if (distex * distex + distey * distey < 100 * scale2[k2.octave]) continue;
            }
            const float num = a * k2.x + b * k2.y + c;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if ((double)dsqr < 3.84 * (double)sigma2[k2.octave]) {
                const unsigned kk = ((unsigned)dist << 16) | (0xffffu - (unsigned)min(k - k0, 0xffff));
                key = kk < key ? kk : key;
            }
        }
    }
    key = orbx::row_min_u32(key);
    if (sub == 0) {
        const bool hit = key != 0xffffffffu;
        const int p = hit ? cand[k0 + (int)(0xffffu - (key & 0xffffu))] : 0;
        match12[i] = hit ? perm2[p] : -1;
        rot[i] = hit ? ang1[s] - ang2[p] : 0.0f;
    }
}

// The projection prefixes of the remaining SearchByProjection forms and of SearchBySim3, one thread per list entry, in the
// reference's float / double order op by op (cv::Mat arithmetic as k_project_points restates it):
//   FORM_LAST   SearchByProjection(CurrentFrame, LastFrame, th, bMono)            src/ORBmatcher.cc:1557-1591
//   FORM_KF     SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) :1702-1735  (no depth test, as the reference)
//   FORM_SIM3   SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)             :518-563    (pose = the decomposed Scw)
//   FORM_PAIR   one direction of SearchBySim3                                     :1360-1396 / :1442-1478 (two transforms)
// Output: the GetFeaturesInArea query of every entry (r < 0: the entry is skipped before the search).
enum { FORM_LAST = 0, FORM_KF = 1, FORM_SIM3 = 2, FORM_PAIR = 3 };
struct FormCam {
    float fx, fy, cx, cy, min_x, max_x, min_y, max_y, mbf, log_scale, th;
    float R[9], t[3], Ow[3];      // Rcw, tcw, Ow  (FORM_PAIR: Ra, ta = first transform)
    float R2[9], t2[3];           // FORM_PAIR: the second transform (sR21, t21 / sR12, t12)
    int nlevels, form, forward, backward;
};

__device__ __forceinline__ float gemm_row(const float *R, int r, float b0, float b1, float b2, float t)
{
    return (float)((double)(R[3 * r] * b0 + R[3 * r + 1] * b1 + R[3 * r + 2] * b2) + (double)t);
}

__global__ __launch_bounds__(MT) void k_project_form(const uint8_t *__restrict__ valid, const float *__restrict__ pos,
                                                     const float *__restrict__ nrm, const float *__restrict__ mind,
                                                     const float *__restrict__ maxd, const int *__restrict__ octave, int m, FormCam cam,
                                                     const float *__restrict__ scale, WinQuery *__restrict__ q, StageJob job)
{
    if (stage_block(job)) return;
    const int i = (blockIdx.x - job.nblocks) * MT + threadIdx.x;
    if (i >= m) return;
    WinQuery w = {0.f, 0.f, -1.f, 0.f, 0, -1}; // r < 0: no candidates
    if (valid[i]) {
        const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
        float X = gemm_row(cam.R, 0, P0, P1, P2, cam.t[0]), Y = gemm_row(cam.R, 1, P0, P1, P2, cam.t[1]),
              Z = gemm_row(cam.R, 2, P0, P1, P2, cam.t[2]);
        if (cam.form == FORM_LAST) {
            const float invzc = (float)(1.0 / (double)Z);                          // :1565
            bool ok = !(invzc < 0);
            const float u = cam.fx * X * invzc + cam.cx, v = cam.fy * Y * invzc + cam.cy;
            ok = ok && !(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y);
            if (ok) {
                const int oct = octave[i];
                w.u = u; w.v = v; w.r = cam.th * scale[oct]; w.xr = u - cam.mbf * invzc;
                w.min_level = cam.forward ? oct : (cam.backward ? 0 : oct - 1);     // :1586-1591
                w.max_level = cam.forward ? -1 : (cam.backward ? oct : oct + 1);
            }
        } else if (cam.form == FORM_KF) {
            const float invzc = (float)(1.0 / (double)Z);                          // :1711
            const float u = cam.fx * X * invzc + cam.cx, v = cam.fy * Y * invzc + cam.cy;
            bool ok = !(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y);
            const float PO0 = P0 - cam.Ow[0], PO1 = P1 - cam.Ow[1], PO2 = P2 - cam.Ow[2];
            const float dist3D = (float)sqrt((double)PO0 * (double)PO0 + (double)PO1 * (double)PO1 + (double)PO2 * (double)PO2);
            const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
            ok = ok && !(dist3D < minDistance || dist3D > maxDistance);
            if (ok) {
                const float ratio = maxd[i] / dist3D;
                int level = (int)ceilf(orbx_logf_glibc_f32(ratio) / cam.log_scale);
                if (level < 0) level = 0; else if (level >= cam.nlevels) level = cam.nlevels - 1;
                w.u = u; w.v = v; w.r = cam.th * scale[level]; w.min_level = level - 1; w.max_level = level + 1;
            }
        } else {
            if (cam.form == FORM_PAIR) {     // p3Dc2 = sR21 * (R1w * p3Dw + t1w) + t21
                const float a0 = X, a1 = Y, a2 = Z;
                X = gemm_row(cam.R2, 0, a0, a1, a2, cam.t2[0]); Y = gemm_row(cam.R2, 1, a0, a1, a2, cam.t2[1]);
                Z = gemm_row(cam.R2, 2, a0, a1, a2, cam.t2[2]);
            }
            bool ok = !(Z < 0.0f);
            const float invz = cam.form == FORM_SIM3 ? 1 / Z : (float)(1.0 / (double)Z);      // :531 / :1367
            const float x = X * invz, y = Y * invz;
            const float u = cam.fx * x + cam.cx, v = cam.fy * y + cam.cy;
            ok = ok && (u >= cam.min_x && u < cam.max_x && v >= cam.min_y && v < cam.max_y);  // KeyFrame::IsInImage
            const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
            float dist;
            if (cam.form == FORM_SIM3) {
                const float PO0 = P0 - cam.Ow[0], PO1 = P1 - cam.Ow[1], PO2 = P2 - cam.Ow[2];
                dist = (float)sqrt((double)PO0 * (double)PO0 + (double)PO1 * (double)PO1 + (double)PO2 * (double)PO2);
                ok = ok && !(dist < minDistance || dist > maxDistance);
                const double dot = (double)PO0 * (double)nrm[3 * i] + (double)PO1 * (double)nrm[3 * i + 1] + (double)PO2 * (double)nrm[3 * i + 2];
                ok = ok && !(dot < 0.5 * (double)dist);
            } else {
                dist = (float)sqrt((double)X * (double)X + (double)Y * (double)Y + (double)Z * (double)Z);   // cv::norm(p3Dc2)
                ok = ok && !(dist < minDistance || dist > maxDistance);
            }
            if (ok) {
                const float ratio = maxd[i] / dist;
                int level = (int)ceilf(orbx_logf_glibc_f32(ratio) / cam.log_scale);
                if (level < 0) level = 0; else if (level >= cam.nlevels) level = cam.nlevels - 1;
                w.u = u; w.v = v; w.r = cam.th * scale[level]; w.min_level = level - 1; w.max_level = level;
            }
        }
    }
    q[i] = w;
}

// SearchBySim3's acceptance and agreement check (src/ORBmatcher.cc:1426-1429, :1505-1507, :1509-1524) over the two window searches.
__global__ __launch_bounds__(MT) void k_sim3_agree(const int *__restrict__ best1, const int *__restrict__ idx1, int n1,
                                                   const int *__restrict__ best2, const int *__restrict__ idx2, int n2, int th_high,
                                                   int *__restrict__ vn1, int *__restrict__ vn2, int *__restrict__ m12, int *__restrict__ nfound)
{
    const int i = blockIdx.x * MT + threadIdx.x;
    if (i < n2) vn2[i] = idx2[i] >= 0 && best2[i] <= th_high ? idx2[i] : -1;
    bool found = false;
    if (i < n1) {
        const int a = idx1[i] >= 0 && best1[i] <= th_high ? idx1[i] : -1;
        vn1[i] = a;
        if (a >= 0 && a < n2) found = (idx2[a] >= 0 && best2[a] <= th_high ? idx2[a] : -1) == i;
        m12[i] = found ? a : -1;
    }
    const unsigned long long b = __ballot(found);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(nfound, __popcll(b));
}

std::atomic<int> g_last_iterations{0};    // iterations of the last k_resolve_par (-1: it gave up and k_resolve ran)
std::atomic<int> g_force_sequential{0};   // orbm_debug_force_sequential_resolver: tests run both resolvers

struct SortedFrame {
    std::vector<SeqKp> kp;
    std::vector<int> cell_off; // [COLS*ROWS + 1] runs of the sorted array per grid cell (col * ROWS + row)
    GridParams gp = {0.f, 0.f, 0.f, 0.f};
    std::vector<int> perm;
    std::vector<float> angle;
    std::vector<uint8_t> desc;
};

// Keypoints in the grid (and not excluded by `skip`), ordered by (cell, index).
void sort_frame(const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *skip, const float *uright, float min_x,
                float min_y, float max_x, float max_y, SortedFrame &sf)
{
    std::vector<WinKp> wk;
    build_winkp(kps, n, skip, uright, min_x, min_y, max_x, max_y, wk);
    // counting sort by grid cell; inside a cell ascending keypoint index (= insertion order, Frame.cc:245-260)
    sf.gp = {min_x, min_y, (float)FRAME_GRID_COLS / (max_x - min_x), (float)FRAME_GRID_ROWS / (max_y - min_y)}; // Frame.cc:95-96
    sf.cell_off.assign(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, 0);
    for (int j = 0; j < n; ++j)
        if (wk[j].order != 0xffffffffu) sf.cell_off[(wk[j].order >> 16) + 1]++;
    for (int c = 0; c < FRAME_GRID_COLS * FRAME_GRID_ROWS; ++c) sf.cell_off[c + 1] += sf.cell_off[c];
    const size_t ns = (size_t)sf.cell_off[FRAME_GRID_COLS * FRAME_GRID_ROWS];
    std::vector<unsigned> order(ns ? ns : 1);
    {
        std::vector<int> fill(sf.cell_off.begin(), sf.cell_off.end() - 1);
        for (int j = 0; j < n; ++j)
            if (wk[j].order != 0xffffffffu) order[fill[wk[j].order >> 16]++] = wk[j].order;
    }
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

// What the search kernels read of a frame, wherever it lives.
struct FrameView {
    const SeqKp *kp; const uint4 *desc; const float *angle; const int *perm; const int *cell_off;
    GridParams gp;
};

} // namespace

// A frame resident in HBM: the sorted keypoint records, descriptors, angles, the permutation and the cell table of
// k_frame_build in ONE device block (recycled through a pool: a frame per image must not cost a hipMalloc), plus a host
// copy of the permutation (the per-call occupancy masks are given by keypoint index and staged in sorted order).
// Read-only after creation: any number of searches, from any thread, may use it at once.
struct orbm_frame {
    int n = 0, ns = 0, cap = 0, has_uright = 0;
    int min_octave = 0, max_octave = -1;                    // over the keypoints (empty frame: 0, -1)
    float min_x = 0, min_y = 0, max_x = 0, max_y = 0;       // the bounds the SEARCHES use (cell range of a window, image tests)
    GridParams gp = {0.f, 0.f, 0.f, 0.f};                   // ... with the cell pitch the grid was built with
    char *block = nullptr;
    std::atomic<int> *refs = nullptr;                       // handles sharing the block (orbm_frame_alias)
    SeqKp *kp = nullptr; uint4 *desc = nullptr; float *angle = nullptr; int *perm = nullptr, *cell_off = nullptr;
    FrameHdr *hdr = nullptr;
    std::vector<int> perm_host;                             // [n]: keypoint index at sorted position (the first ns: inside the grid)
    std::vector<int> inv_host;                              // [n]: sorted position of keypoint index
};

namespace {

std::mutex g_frame_mu;
std::vector<std::pair<int, char *>> g_frame_pool;     // (capacity, block) of destroyed frames

size_t frame_block_bytes(int cap)
{
    return 256 + ((sizeof(SeqKp) + 32 + 4 + 4) * (size_t)cap + 255) / 256 * 256 + sizeof(int) * (FB_NC + 1);
}

void frame_pointers(orbm_frame *f)
{
    const int cap = f->cap;
    char *p = f->block;
    f->hdr = reinterpret_cast<FrameHdr *>(p); p += 256;
    f->desc = reinterpret_cast<uint4 *>(p); p += (size_t)32 * cap;
    f->kp = reinterpret_cast<SeqKp *>(p); p += sizeof(SeqKp) * (size_t)cap;
    f->angle = reinterpret_cast<float *>(p); p += sizeof(float) * (size_t)cap;
    f->perm = reinterpret_cast<int *>(p);
    f->cell_off = reinterpret_cast<int *>(f->block + frame_block_bytes(cap) - sizeof(int) * (FB_NC + 1));
}

int frame_alloc(orbm_frame *f, int n)
{
    const int cap = std::max(2048, (n + 1023) / 1024 * 1024);
    {
        std::lock_guard<std::mutex> lk(g_frame_mu);
        for (size_t i = 0; i < g_frame_pool.size(); ++i)
            if (g_frame_pool[i].first == cap) {
                f->block = g_frame_pool[i].second;
                g_frame_pool.erase(g_frame_pool.begin() + (long)i);
                break;
            }
    }
    if (!f->block && hipMalloc((void **)&f->block, frame_block_bytes(cap)) != hipSuccess) { f->block = nullptr; return -1; }
    f->cap = cap;
    f->refs = new std::atomic<int>(1);
    frame_pointers(f);
    return 0;
}

void frame_release(orbm_frame *f)
{
    if (f->block && f->refs && f->refs->fetch_sub(1) == 1) {      // the last handle on the block
        delete f->refs;
        {
            std::lock_guard<std::mutex> lk(g_frame_mu);
            if (g_frame_pool.size() < 64) { g_frame_pool.emplace_back(f->cap, f->block); f->block = nullptr; }
        }
        if (f->block) (void)hipFree(f->block);
    }
    delete f;
}

// k_frame_build on `st` from device-resident inputs, then the header and the permutation back to the host (one small
// block through the pinned arena of the leased workspace `w`, whose first pin_need bytes must be free for it).
int frame_build(orbm_frame *f, Workspace &w, const orbx_keypoint *d_kps, const uint4 *d_desc, const float2 *d_xy, const float *d_ur,
                const int *d_n, int n_host, int n_bound)
{
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_frame_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
        attr_set.store(true, std::memory_order_release);
    }
    hipStream_t st = w.st;
    hipLaunchKernelGGL(k_frame_build, dim3(1), dim3(FB_T), FB_LDS, st, d_kps, d_desc, d_xy, d_ur, d_n, n_host, f->gp, f->kp, f->desc,
                       f->angle, f->perm, f->cell_off, f->hdr);
    ORBX_HIP(hipGetLastError());
    // results: hdr (256-byte slot) + perm[n_bound], staged out through the arena's tail
    const size_t bytes = (256 + sizeof(int) * (size_t)n_bound + 15) & ~(size_t)15;
    char *pin = w.pin + w.pin_cap - ((bytes + 255) & ~(size_t)255);
    ORBX_HIP(hipMemcpyAsync(pin, f->hdr, sizeof(FrameHdr), hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(pin + 256, f->perm, sizeof(int) * (size_t)n_bound, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    FrameHdr h;
    memcpy(&h, pin, sizeof(h));
    f->n = h.n; f->ns = h.ns;
    f->min_octave = h.n ? h.min_octave : 0; f->max_octave = h.n ? h.max_octave : -1;
    if (h.n < 0 || h.n > n_bound || h.ns < 0 || h.ns > h.n) ORBX_FAIL(ORBX_ERR_HIP, "frame header out of range");
    f->perm_host.assign(reinterpret_cast<const int *>(pin + 256), reinterpret_cast<const int *>(pin + 256) + h.n);
    f->inv_host.assign((size_t)h.n, 0);
    for (int sp = 0; sp < h.n; ++sp) {
        const int j = f->perm_host[sp];
        if (j < 0 || j >= h.n) ORBX_FAIL(ORBX_ERR_HIP, "frame permutation out of range");
        f->inv_host[j] = sp;
    }
    return ORBX_OK;
}

// The frame of a call: sorted on the host and staged with the call's inputs (the host-array entry points), or resident.
struct FrameSrc {
    const SortedFrame *host = nullptr;
    const orbm_frame *res = nullptr;
    size_t o_k = 0, o_b = 0, o_kang = 0, o_perm = 0, o_cell = 0;
    bool all_positions = false;     // resident frame addressed by explicit candidate lists: the positions behind the grid's count too
    explicit FrameSrc(const SortedFrame &sf) : host(&sf) {}
    explicit FrameSrc(const orbm_frame *f, bool all = false) : res(f), all_positions(all) {}
    int ns() const { return host ? (int)host->perm.size() : (all_positions ? res->n : res->ns); }
    const int *perm() const { return host ? host->perm.data() : res->perm_host.data(); }
    GridParams gp() const { return host ? host->gp : res->gp; }
    void carve(Workspace &w)
    {
        if (!host) return;
        const size_t m = ns() ? (size_t)ns() : 1;
        o_k = w.carve(sizeof(SeqKp) * m); o_b = w.carve(32 * m); o_kang = w.carve(sizeof(float) * m); o_perm = w.carve(sizeof(int) * m);
        o_cell = w.carve(sizeof(int) * (FRAME_GRID_COLS * FRAME_GRID_ROWS + 1));
    }
    void fill(Workspace &w) const
    {
        if (!host) return;
        const size_t m = (size_t)ns();
        if (!host->kp.empty()) memcpy(w.h<char>(o_k), host->kp.data(), sizeof(SeqKp) * m);
        if (!host->cell_off.empty()) memcpy(w.h<char>(o_cell), host->cell_off.data(), sizeof(int) * host->cell_off.size());
        memcpy(w.h<char>(o_b), host->desc.data(), 32 * m);
        memcpy(w.h<char>(o_kang), host->angle.data(), sizeof(float) * m);
        memcpy(w.h<char>(o_perm), host->perm.data(), sizeof(int) * m);
    }
    FrameView view(const Workspace &w) const
    {
        if (host) return {w.d<SeqKp>(o_k), w.d<uint4>(o_b), w.d<float>(o_kang), w.d<int>(o_perm), w.d<int>(o_cell), host->gp};
        return {res->kp, res->desc, res->angle, res->perm, res->cell_off, res->gp};
    }
    // occupancy by keypoint index -> by sorted position
    void fill_occ(uint8_t *dst, const uint8_t *occupied) const
    {
        const int m = ns();
        const int *pm = perm();
        for (int s = 0; s < m; ++s) dst[s] = occupied[pm[s]];
    }
};

// A list of map points as the whole-function searches take it (orbm_points), staged with a call, and the prefix kernel that
// turns it into window queries on the device.
struct PointsPrefix {
    const orbm_points *pts = nullptr;
    FormCam cam;
    const float *scale = nullptr;
    bool frustum = false;        // Frame::isInFrustum + the window of SearchByProjection(Frame&, vector<MapPoint*>&, th): k_project_points, mode 0
    ProjectCam pcam;
    orbm_projected_point *proj_host = nullptr;   // optional: the projections back to the caller
    size_t o_proj = 0;
    void carve_scratch(Workspace &w) { if (frustum) o_proj = w.carve(sizeof(orbm_projected_point) * (size_t)(pts->n ? pts->n : 1)); }
    size_t o_valid = 0, o_pos = 0, o_nrm = 0, o_min = 0, o_max = 0, o_oct = 0, o_sc = 0;
    void carve(Workspace &w)
    {
        const size_t m = pts->n ? (size_t)pts->n : 1;
        o_valid = w.carve(m); o_pos = w.carve(sizeof(float) * 3 * m);
        o_nrm = w.carve(pts->normal ? sizeof(float) * 3 * m : 1);
        o_min = w.carve(pts->min_distance ? sizeof(float) * m : 1); o_max = w.carve(pts->max_distance ? sizeof(float) * m : 1);
        o_oct = w.carve(pts->octave ? sizeof(int) * m : 1);
        o_sc = w.carve(sizeof(float) * cam.nlevels);
    }
    void fill(Workspace &w) const
    {
        const size_t m = (size_t)pts->n;
        memcpy(w.h<char>(o_valid), pts->valid, m);
        memcpy(w.h<char>(o_pos), pts->pos, sizeof(float) * 3 * m);
        if (pts->normal) memcpy(w.h<char>(o_nrm), pts->normal, sizeof(float) * 3 * m);
        if (pts->min_distance) memcpy(w.h<char>(o_min), pts->min_distance, sizeof(float) * m);
        if (pts->max_distance) memcpy(w.h<char>(o_max), pts->max_distance, sizeof(float) * m);
        if (pts->octave) memcpy(w.h<char>(o_oct), pts->octave, sizeof(int) * m);
        memcpy(w.h<char>(o_sc), scale, sizeof(float) * cam.nlevels);
    }
    // the prefix's inputs are read where the host staged them (pinned memory); `job` = the call's upload, done by the same launch
    void launch(const Workspace &w, WinQuery *dq, hipStream_t st, StageJob job = StageJob{nullptr, nullptr, 0, 0}) const
    {
        const int nb = (pts->n + MT - 1) / MT + job.nblocks;
        if (!nb) return;
        if (frustum) {
            hipLaunchKernelGGL(k_project_points, dim3(nb), dim3(MT), 0, st, (const uint8_t *)w.h<uint8_t>(o_valid),
                               (const float *)w.h<float>(o_pos), (const float *)w.h<float>(o_nrm), (const float *)w.h<float>(o_min),
                               (const float *)w.h<float>(o_max), pts->n, pcam, (const float *)w.h<float>(o_sc), w.d<orbm_projected_point>(o_proj), dq, job);
            return;
        }
        hipLaunchKernelGGL(k_project_form, dim3(nb), dim3(MT), 0, st, (const uint8_t *)w.h<uint8_t>(o_valid),
                           (const float *)w.h<float>(o_pos), (const float *)w.h<float>(o_nrm), (const float *)w.h<float>(o_min),
                           (const float *)w.h<float>(o_max), (const int *)w.h<int>(o_oct), pts->n, cam, (const float *)w.h<float>(o_sc), dq, job);
    }
};

// Shared driver.  mode 0: projection family / BoW lists; mode 1: SearchForInitialization.
// Inputs are staged in the workspace's pinned arena and uploaded with one copy; results
// come back with one copy.  Everything runs on the workspace's stream.  The window queries come from the host (`queries`)
// or from `prefix` on the device (then queries_out, if given, receives them).
int run_sequential(int mode, const WinQuery *queries, PointsPrefix *prefix, const uint8_t *qdesc, const float *qangle, const uint8_t *qtakes,
                   int nq, FrameSrc &fs, int n, const uint8_t *occupied, int has_uright, int th, float nnratio, int accept_mode, int check,
                   int32_t *match_kp, int32_t *match_q, int *nmatches, WinQuery *queries_out = nullptr, const int32_t *cand_off = nullptr,
                   const int32_t *cand_beg = nullptr, const int32_t *cand_idx = nullptr, int ncand = 0,
                   const int32_t *seg = nullptr, int nseg = 0, const orbm_frame *qframe = nullptr, const int32_t *qsrc = nullptr)
{
    // qframe / qsrc (explicit candidate lists only): query i = the feature at sorted position qsrc[i] of the resident frame qframe;
    // qdesc and qangle are not read
    const int ns = fs.ns();
    if (ns > SEQ_MAXN || nq > 65536) ORBX_FAIL(ORBX_ERR_CAPACITY, "frame too large for the sequential resolver");
    for (int j = 0; j < n && mode == 0; ++j) match_kp[j] = -1;
    for (int i = 0; i < nq; ++i) match_q[i] = -1;
    *nmatches = 0;
    const bool windows = queries || prefix;
    if (queries_out) for (int i = 0; i < nq; ++i) queries_out[i] = {0.f, 0.f, -1.f, 0.f, 0, -1};
    if (nq == 0 || (ns == 0 && !queries_out && !(prefix && prefix->proj_host))) return ORBX_OK;
    const size_t lds = sizeof(int) * (3 * (size_t)ns + (mode == 1 ? nq : 0)) + 16;
    if (lds > 160 * 1024) ORBX_FAIL(ORBX_ERR_CAPACITY, "resolver state exceeds LDS");
    if (mode != 0) { seg = nullptr; nseg = 0; }

    WorkspaceLease lease;
    Workspace &w = *lease.w;
    static bool lds_attr_set = false;
    if (!lds_attr_set) { // the largest request either instantiation can make
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve_par), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));   // (it also has static LDS)
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve_init_par), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        lds_attr_set = true;
    }
    // Window lists: their lengths are known only on the device.  First attempt: every query fills its own region of
    // WIN_STRIDE entries in one pass (no count, no scan, no host wait); if a list does not fit, the overflow flag comes
    // back with the results and the call is repeated on the exact path (count, scan, fill), whose total also travels
    // with the results -- the host never waits in the middle of a call.
    // (SearchForInitialization's windows are 200 px wide: a few hundred candidates each, where the projection searches see tens)
    const int WIN_STRIDE = mode == 1 ? 1024 : 256;
    bool exact = false, sized = false;
    size_t ent_need = cand_off ? (size_t)cand_off[nq] : (size_t)nq * WIN_STRIDE;
    // the projection family (one segment, MODE 0) resolves as a parallel fixed point (k_resolve_par, rotation check fused); if its
    // dependency chains are longer than the kernel iterates, the call is repeated on the one-wave sequential resolver
    // SearchForInitialization (MODE 1) likewise (k_resolve_init_par): its per-slot acceptor lists take C + 3 ints of LDS per keypoint
    // (the BoW searches' segments -- one per vocabulary node, disjoint candidate sets -- only matter to the sequential resolver:
    // the fixed point needs no partition, queries of different nodes simply never meet)
    bool sequential = g_force_sequential.load(std::memory_order_relaxed) != 0;
    const int init_c = ns ? std::min(7, (144 * 1024 / 4) / ns - 3) : 7;
    if (mode == 1 && init_c < 2) sequential = true;
    for (int attempt = 0; attempt < 4; ++attempt) {
        w.used = 0;
        // staged inputs (same offsets on both sides), then device-only arrays, then the result block
        const size_t o_qh = w.carve(queries ? sizeof(WinQuery) * nq : 1), o_a = w.carve(qframe ? sizeof(int) * (size_t)nq : (size_t)32 * nq);
        fs.carve(w);
        const size_t o_qang = w.carve(sizeof(float) * nq), o_tk = w.carve(nq), o_occ = w.carve(occupied && ns ? (size_t)ns : 1),
                     o_off = w.carve(sizeof(int) * (nq + 1)),
                     o_cbeg = w.carve(sizeof(int) * (size_t)nq), o_cand = w.carve(sizeof(int) * (size_t)(ncand ? ncand : 1)),
                     o_seg = w.carve(sizeof(int) * (size_t)(nseg + 1));
        const size_t staged = w.used;
        if (prefix) prefix->carve(w);              // read by the prefix kernel where they are staged: not part of the upload
        const size_t pin_in = w.used;
        if (prefix) prefix->carve_scratch(w);
        const size_t o_qd = w.carve(queries ? 1 : sizeof(WinQuery) * nq);
        const size_t o_q = queries ? o_qh : o_qd;
        const size_t o_top = w.carve(sizeof(unsigned) * TOPK * (size_t)nq), o_cnt = w.carve(sizeof(int) * nq), o_acc = w.carve(sizeof(int) * nq),
                     o_lend = w.carve(sizeof(int) * nq),
                     o_state = w.carve(sizeof(int) * (size_t)std::max(std::max(ns, nq), 1));
        const size_t o_res = w.used;
        const size_t o_mq = w.carve(sizeof(int) * nq), o_mk = w.carve(sizeof(int) * (size_t)(n ? n : 1)), o_nm = w.carve(4 * sizeof(int));
        const size_t total_bytes = w.used, res_bytes = total_bytes - o_res;
        // optional outputs that are not part of the result block travel through the tail of the pinned arena
        const size_t tq_bytes = queries_out ? (sizeof(WinQuery) * nq + 255) & ~(size_t)255 : 0;
        const size_t tp_bytes = prefix && prefix->frustum && prefix->proj_host ? (sizeof(orbm_projected_point) * (size_t)nq + 255) & ~(size_t)255 : 0;
        if (w.reserve(total_bytes, std::max(pin_in, res_bytes) + tq_bytes + tp_bytes)) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
        char *tq_pin = w.pin + w.pin_cap - tq_bytes, *tp_pin = tq_pin - tp_bytes;
        if (w.reserve_entries(ent_need)) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
        hipStream_t st = w.st;

        if (queries) memcpy(w.h<char>(o_qh), queries, sizeof(WinQuery) * nq);
        if (qframe) memcpy(w.h<char>(o_a), qsrc, sizeof(int) * (size_t)nq);
        else memcpy(w.h<char>(o_a), qdesc, (size_t)32 * nq);
        fs.fill(w);
        if (!qframe) {   // (from a resident frame: k_list_fill sets the angles down)
            if (qangle) memcpy(w.h<char>(o_qang), qangle, sizeof(float) * nq); else memset(w.h<char>(o_qang), 0, sizeof(float) * nq);
        }
        if (qtakes) memcpy(w.h<char>(o_tk), qtakes, nq); else memset(w.h<char>(o_tk), 1, nq);
        if (occupied && ns) fs.fill_occ(w.h<uint8_t>(o_occ), occupied);
        if (cand_off) {
            memcpy(w.h<char>(o_off), cand_off, sizeof(int) * (nq + 1));
            memcpy(w.h<char>(o_cbeg), cand_beg, sizeof(int) * (size_t)nq);
            if (ncand) memcpy(w.h<char>(o_cand), cand_idx, sizeof(int) * ncand);
        }
        if (seg) memcpy(w.h<char>(o_seg), seg, sizeof(int) * (size_t)(nseg + 1));
        if (prefix) prefix->fill(w);
        const bool fused_upload = prefix && orbx::stage_ok(w.dev, w.pin, staged);
        if (!fused_upload) ORBX_HIP(orbx::stage_in(w.dev, w.pin, staged, st));
        // ONE fill for everything that needs a preset (a launch each was 3-4 us of a 0.1-ms call): match_kp = -1 (slot untouched);
        // segments write back touched slots of the state only, so it is preset to -1 as well (the span in between, match_q, is
        // rewritten in full anyway); and the two counters START AT -1: the match count is overwritten (one workgroup) or added
        // to (segments: the host adds the 1 back), the overflow flag is CLEARED to 0 by a list that does not fit
        // (k_resolve_par writes every slot and its count itself; the two flags behind the count -- "a list outgrew its region",
        // "the fixed point was reached" -- are raised by writing this call's generation number, so nothing needs a preset there)
        const int gen = (int)(w.gen = (w.gen % 0x7ffffffe) + 1);
        if (sequential) {
            const size_t f0 = seg ? o_state : o_mk;
            ORBX_HIP(hipMemsetAsync(w.d<char>(f0), 0xff, o_nm + sizeof(int) - f0, st));
        }

        const FrameView fv = fs.view(w);
        WinQuery *dq = w.d<WinQuery>(o_q);
        if (prefix) prefix->launch(w, dq, st, fused_upload ? stage_job(w.dev, w.pin, staged) : StageJob{nullptr, nullptr, 0, 0});
        if (tq_bytes) ORBX_HIP(hipMemcpyAsync(tq_pin, dq, sizeof(WinQuery) * nq, hipMemcpyDeviceToHost, st));
        if (tp_bytes) ORBX_HIP(hipMemcpyAsync(tp_pin, w.d<char>(prefix->o_proj), sizeof(orbm_projected_point) * (size_t)nq, hipMemcpyDeviceToHost, st));
        if (ns == 0) {      // nothing to search: the prefix's outputs are the whole result
            ORBX_HIP(hipStreamSynchronize(st));
            if (tq_bytes) memcpy(queries_out, tq_pin, sizeof(WinQuery) * nq);
            if (tp_bytes) memcpy(prefix->proj_host, tp_pin, sizeof(orbm_projected_point) * (size_t)nq);
            return ORBX_OK;
        }
        const uint4 *da = w.d<uint4>(o_a), *db = fv.desc;
        const SeqKp *dk = fv.kp;
        const uint8_t *docc = occupied ? w.d<uint8_t>(o_occ) : nullptr;
        int *doff = w.d<int>(o_off), *dnm = w.d<int>(o_nm);
        const int init_dist = mode == 0 ? 256 : INT_MAX;
        const dim3 g((nq + MT / 64 - 1) / (MT / 64)); // one wave per query
        unsigned *dtop = w.d<unsigned>(o_top);
        const int *lbeg = doff, *lend = doff + 1; // CSR lists; the strided path has its own bounds
        if (!windows) { // explicit candidate lists
            hipLaunchKernelGGL(k_list_fill, g, dim3(MT), 0, st, qframe ? (const uint4 *)qframe->desc : da, nq, db, (const int *)doff,
                               (const int *)w.d<int>(o_cbeg), (const int *)w.d<int>(o_cand), w.ent, dtop,
                               qframe ? (const int *)w.d<int>(o_a) : (const int *)nullptr, qframe ? (const float *)qframe->angle : (const float *)nullptr,
                               w.d<float>(o_qang));
        } else if (!exact) {
            lbeg = w.d<int>(o_cnt); lend = w.d<int>(o_lend);
            hipLaunchKernelGGL(k_win_wave<2>, g, dim3(MT), 0, st, (const WinQuery *)dq, da, nq, dk, db, fv.cell_off, docc, fv.gp, has_uright,
                               init_dist, (int *)nullptr, (const int *)nullptr, w.ent, WIN_STRIDE, w.d<int>(o_cnt), w.d<int>(o_lend), dnm + 1, dtop, gen);
        } else {
            hipLaunchKernelGGL(k_win_wave<0>, g, dim3(MT), 0, st, (const WinQuery *)dq, da, nq, dk, db, fv.cell_off, docc, fv.gp, has_uright,
                               init_dist, w.d<int>(o_cnt), (const int *)nullptr, (unsigned *)nullptr, 0, (int *)nullptr, (int *)nullptr,
                               (int *)nullptr, (unsigned *)nullptr, gen);
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(MT), 0, st, (const int *)w.d<int>(o_cnt), nq, doff, dnm + 1);
            if (!sized) { // the total is needed to size the buffer: the one host wait of this (rare) path
                sized = true;
                int total = 0;
                ORBX_HIP(hipMemcpyAsync(&total, dnm + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                ORBX_HIP(hipStreamSynchronize(st));
                ent_need = (size_t)total;
                if (w.reserve_entries(ent_need)) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
            }
            hipLaunchKernelGGL(k_win_wave<1>, g, dim3(MT), 0, st, (const WinQuery *)dq, da, nq, dk, db, fv.cell_off, docc, fv.gp, has_uright,
                               init_dist, (int *)nullptr, (const int *)doff, w.ent, 0, (int *)nullptr, (int *)nullptr, (int *)nullptr, dtop, gen);
        }
        const int *dseg = seg ? w.d<int>(o_seg) : nullptr;
        const dim3 gr(seg ? nseg : 1);
        if (!sequential && mode == 1) {
            hipLaunchKernelGGL(k_resolve_init_par, dim3(1), dim3(RES_T), sizeof(int) * (size_t)(init_c + 3) * ns, st, (const unsigned *)w.ent,
                               (const unsigned *)dtop, lbeg, lend, nq, ns, init_c, th, nnratio, (const float *)w.d<float>(o_qang),
                               fv.angle, fv.perm, check, w.d<unsigned>(o_acc), (const int *)dnm, gen,
                               reinterpret_cast<int *>(w.pin + (o_mq - o_res)), reinterpret_cast<int *>(w.pin + (o_nm - o_res)));
        } else if (!sequential) {
            hipLaunchKernelGGL(k_resolve_par, dim3(1), dim3(RES_T), sizeof(int) * 4 * (size_t)ns, st, (const unsigned *)w.ent, (const unsigned *)dtop, lbeg, lend,
                               nq, ns, (const uint8_t *)w.d<uint8_t>(o_tk), th, nnratio, accept_mode, (const float *)w.d<float>(o_qang),
                               fv.angle, fv.perm, check, w.d<int>(o_mq), w.d<int>(o_mk), n, (const int *)dnm, gen,
                               reinterpret_cast<int *>(w.pin + (o_mq - o_res)), reinterpret_cast<int *>(w.pin + (o_mk - o_res)), reinterpret_cast<int *>(w.pin + (o_nm - o_res)));
        } else if (mode == 0) {
            hipLaunchKernelGGL(k_resolve<0>, gr, dim3(64), lds, st, (const unsigned *)w.ent, (const unsigned *)dtop, lbeg, lend, nq, ns,
                               (const uint8_t *)w.d<uint8_t>(o_tk), th, nnratio, accept_mode, w.d<int>(o_acc), w.d<int>(o_state), dnm, dseg);
            hipLaunchKernelGGL(k_rotation<0>, dim3(1), dim3(MT), 0, st, (const int *)w.d<int>(o_acc), (const int *)w.d<int>(o_state), nq, ns,
                               (const float *)w.d<float>(o_qang), fv.angle, fv.perm, check,
                               w.d<int>(o_mq), w.d<int>(o_mk), dnm);
        } else {
            hipLaunchKernelGGL(k_resolve<1>, gr, dim3(64), lds, st, (const unsigned *)w.ent, (const unsigned *)dtop, lbeg, lend, nq, ns,
                               (const uint8_t *)w.d<uint8_t>(o_tk), th, nnratio, 0, w.d<int>(o_acc), w.d<int>(o_state), dnm, dseg);
            hipLaunchKernelGGL(k_rotation<1>, dim3(1), dim3(MT), 0, st, (const int *)w.d<int>(o_acc), (const int *)w.d<int>(o_state), nq, ns,
                               (const float *)w.d<float>(o_qang), fv.angle, fv.perm, check,
                               w.d<int>(o_mq), w.d<int>(o_mk), dnm);
        }
        ORBX_HIP(hipGetLastError());
        if (sequential) ORBX_HIP(orbx::stage_out(w.pin, w.dev + o_res, res_bytes, st));   // (k_resolve_par wrote the block itself)
        ORBX_HIP(hipStreamSynchronize(st));
        int flag = 0;
        memcpy(&flag, w.pin + (o_nm - o_res) + sizeof(int), sizeof(int));
        if (windows && !exact && flag == gen) { // a window list outgrew its region: once more on the exact path
            exact = true;
            continue;
        }
        if (!sequential) {
            int conv = 0;
            memcpy(&conv, w.pin + (o_nm - o_res) + 2 * sizeof(int), sizeof(int));
            if (conv != gen) { sequential = true; g_last_iterations.store(-1, std::memory_order_relaxed); continue; }
            memcpy(&conv, w.pin + (o_nm - o_res) + 3 * sizeof(int), sizeof(int));
            g_last_iterations.store(conv, std::memory_order_relaxed);
        }
        memcpy(match_q, w.pin + (o_mq - o_res), sizeof(int) * nq);
        if (mode == 0 && n) memcpy(match_kp, w.pin + (o_mk - o_res), sizeof(int) * n);
        memcpy(nmatches, w.pin + (o_nm - o_res), sizeof(int));
        if (tq_bytes) memcpy(queries_out, tq_pin, sizeof(WinQuery) * nq);
        if (tp_bytes) memcpy(prefix->proj_host, tp_pin, sizeof(orbm_projected_point) * (size_t)nq);
        if (seg && sequential) *nmatches += 1;   // the segments added their counts to the preset -1
        break;
    }
    return ORBX_OK;
}

// ---- cv::Mat float arithmetic of the pose handling in front of the projection loops, on the host (a handful of operations
// per call; -ffp-contract=off keeps them unfused).  OpenCV 3.4 semantics restated (gemm's 3 x 3 special case: float row
// sum left to right, then float(double(sum) * alpha + double(c) * beta); scaling by a FLOAT factor with a + 0.0f).
void pose_parts(const float *T16, float *R, float *t)
{
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = T16[4 * r + c]; t[r] = T16[4 * r + 3]; }
}
void gemm3(const float *A, const float *b, double alpha, const float *c, double beta, float *d)
{
    float out[3];
    for (int k = 0; k < 3; ++k) {
        const float t = A[3 * k] * b[0] + A[3 * k + 1] * b[1] + A[3 * k + 2] * b[2];
        out[k] = (float)((double)t * alpha + (double)(c ? c[k] : 0.0f) * beta);
    }
    d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
}
void transpose3(const float *A, float *At)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) At[3 * r + c] = A[3 * c + r];
}
void scale_mat(const float *A, int n, double s, float *out)
{
    const float a = (float)s, b = (float)0.0;
    for (int i = 0; i < n; ++i) out[i] = A[i] * a + b;
}
void neg_Rt_t(const float *R, const float *t, float *out)   // -R.t() * t  (ORBmatcher.cc:1542, :1679, :504)
{
    float Rt[9];
    transpose3(R, Rt);
    gemm3(Rt, t, -1.0, nullptr, 0.0, out);
}

void form_cam_common(FormCam &c, const orbm_view *v, const orbm_frame *f, int form, float th)
{
    memset(&c, 0, sizeof(c));
    c.fx = v->fx; c.fy = v->fy; c.cx = v->cx; c.cy = v->cy;
    c.min_x = f->min_x; c.max_x = f->max_x; c.min_y = f->min_y; c.max_y = f->max_y;
    c.mbf = v->mbf; c.log_scale = v->log_scale_factor; c.th = th; c.nlevels = v->nlevels; c.form = form;
}

bool bad_view(const orbm_view *v) { return !v || !v->scale_factors || v->nlevels < 1 || v->nlevels > 16; }
bool bad_points(const orbm_points *p, bool need_range, bool need_normal, bool need_octave, int nlevels)
{
    if (!p || p->n < 0 || p->n > 65536) return true;
    if (p->n == 0) return false;
    if (!p->valid || !p->pos || !p->desc) return true;
    if (need_range && (!p->min_distance || !p->max_distance)) return true;
    if (need_normal && !p->normal) return true;
    if (need_octave) {
        if (!p->octave) return true;
        for (int i = 0; i < p->n; ++i)
            if (p->valid[i] && (p->octave[i] < 0 || p->octave[i] >= nlevels)) return true;
    }
    return false;
}

// One window search without coupling (k_win_best) for device-resident queries: returns into ob[5][nq] (best, best level, second,
// second level, idx).
void launch_win_best(hipStream_t st, const WinQuery *dq, const uint4 *da, int nq, const FrameView &fv, const uint8_t *docc, int has_uright,
                     int init_dist, const float *inv_sigma2, int fuse_gate, int *ob)
{
    hipLaunchKernelGGL(k_win_best, dim3((nq + MT / 64 - 1) / (MT / 64)), dim3(MT), 0, st, dq, da, nq, fv.kp, fv.desc, fv.cell_off, fv.perm, docc,
                       fv.gp, has_uright, init_dist, inv_sigma2, fuse_gate, ob, ob + nq, ob + 2 * nq, ob + 3 * nq, ob + 4 * nq);
}

int search_window_core(FrameSrc &fs, const orbm_window_query *queries, const uint8_t *qdesc, int nq, const uint8_t *skip, int has_uright,
                       int init_dist, const float *inv_level_sigma2, int nlevels, int fuse_gate, int32_t *best, int32_t *best_level,
                       int32_t *second, int32_t *second_level, int32_t *idx)
{
    const int ns = fs.ns();
    WorkspaceLease lease;
    Workspace &w = *lease.w;
    w.used = 0;
    const size_t o_q = w.carve(sizeof(WinQuery) * nq), o_a = w.carve((size_t)32 * nq);
    fs.carve(w);
    const size_t o_occ = w.carve(skip && ns ? (size_t)ns : 1), o_sg = w.carve(inv_level_sigma2 ? sizeof(float) * nlevels : 1);
    const size_t staged = w.used;
    const size_t o_res = w.carve(sizeof(int) * 5 * (size_t)nq);
    if (w.reserve(w.used, std::max(staged, sizeof(int) * 5 * (size_t)nq + 16))) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
    memcpy(w.h<char>(o_q), queries, sizeof(WinQuery) * nq);
    memcpy(w.h<char>(o_a), qdesc, (size_t)32 * nq);
    fs.fill(w);
    if (skip && ns) fs.fill_occ(w.h<uint8_t>(o_occ), skip);
    if (inv_level_sigma2) memcpy(w.h<char>(o_sg), inv_level_sigma2, sizeof(float) * nlevels);
    ORBX_HIP(orbx::stage_in(w.dev, w.pin, staged, w.st));
    int *ob = w.d<int>(o_res);
    launch_win_best(w.st, w.d<WinQuery>(o_q), w.d<uint4>(o_a), nq, fs.view(w), skip && ns ? w.d<uint8_t>(o_occ) : nullptr, has_uright, init_dist,
                    inv_level_sigma2 ? w.d<float>(o_sg) : nullptr, fuse_gate, ob);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(orbx::stage_out(w.pin, ob, (sizeof(int) * 5 * (size_t)nq + 15) & ~(size_t)15, w.st));
    ORBX_HIP(hipStreamSynchronize(w.st));
    const int *r = w.h<int>(0);
    if (best) memcpy(best, r, sizeof(int) * nq);
    if (best_level) memcpy(best_level, r + nq, sizeof(int) * nq);
    if (second) memcpy(second, r + 2 * nq, sizeof(int) * nq);
    if (second_level) memcpy(second_level, r + 3 * nq, sizeof(int) * nq);
    if (idx) memcpy(idx, r + 4 * nq, sizeof(int) * nq);
    return ORBX_OK;
}

int search_map_core(FrameSrc &fs, int n, const uint8_t *has_mappoint, const float *mp_pos, const float *mp_normal, const float *mp_min_dist,
                    const float *mp_max_dist, const uint8_t *mp_desc, int m, const double *Rcw, const double *tcw, const orbm_camera *cam,
                    const float *scale_factors, int nlevels, float th, float nnratio, int th_reloc, int32_t *matched_mp, int *nmatches,
                    float *proj)
{
    const int ns = fs.ns();
    WorkspaceLease lease;
    Workspace &w = *lease.w;
    w.used = 0;
    const size_t o_pos = w.carve(sizeof(float) * 3 * m), o_nrm = w.carve(sizeof(float) * 3 * m), o_min = w.carve(sizeof(float) * m),
                 o_max = w.carve(sizeof(float) * m), o_md = w.carve((size_t)32 * m);
    fs.carve(w);
    const size_t o_occ = w.carve(has_mappoint && ns ? (size_t)ns : 1), o_sc = w.carve(sizeof(float) * nlevels);
    const size_t staged = w.used;
    const size_t o_q = w.carve(sizeof(WinQuery) * m), o_o = w.carve(sizeof(int) * 5 * (size_t)m);
    const size_t o_res = w.used;
    const size_t o_mk = w.carve(sizeof(int) * n), o_nm = w.carve(sizeof(int)), o_proj = w.carve(sizeof(float) * 4 * m);
    const size_t res_bytes = w.used - o_res;
    if (w.reserve(w.used, std::max(staged, res_bytes))) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
    memcpy(w.h<char>(o_pos), mp_pos, sizeof(float) * 3 * m); memcpy(w.h<char>(o_nrm), mp_normal, sizeof(float) * 3 * m);
    memcpy(w.h<char>(o_min), mp_min_dist, sizeof(float) * m); memcpy(w.h<char>(o_max), mp_max_dist, sizeof(float) * m);
    memcpy(w.h<char>(o_md), mp_desc, (size_t)32 * m);
    fs.fill(w);
    if (has_mappoint && ns) fs.fill_occ(w.h<uint8_t>(o_occ), has_mappoint);
    memcpy(w.h<char>(o_sc), scale_factors, sizeof(float) * nlevels);
    hipStream_t st = w.st;
    ORBX_HIP(orbx::stage_in(w.dev, w.pin, staged, st));
    ORBX_HIP(hipMemsetAsync(w.d<char>(o_mk), 0xff, sizeof(int) * n, st));
    ORBX_HIP(hipMemsetAsync(w.d<char>(o_nm), 0, sizeof(int), st));
    MapCam mc;
    mc.fx = cam->fx; mc.fy = cam->fy; mc.cx = cam->cx; mc.cy = cam->cy;
    mc.bminx = cam->min_x; mc.bmaxx = cam->max_x; mc.bminy = cam->min_y; mc.bmaxy = cam->max_y;
    for (int i = 0; i < 9; ++i) mc.R[i] = Rcw[i];
    for (int i = 0; i < 3; ++i) mc.t[i] = tcw[i];
    mc.th = th; mc.nlevels = nlevels;
    int *ob = w.d<int>(o_o);
    const dim3 g((m + MT - 1) / MT);
    hipLaunchKernelGGL(k_map_frustum, g, dim3(MT), 0, st, (const float *)w.d<float>(o_pos), (const float *)w.d<float>(o_nrm),
                       (const float *)w.d<float>(o_min), (const float *)w.d<float>(o_max), m, mc, (const float *)w.d<float>(o_sc),
                       w.d<WinQuery>(o_q), w.d<float>(o_proj));
    launch_win_best(st, w.d<WinQuery>(o_q), w.d<uint4>(o_md), m, fs.view(w), has_mappoint && ns ? w.d<uint8_t>(o_occ) : nullptr, 0, INT_MAX,
                    nullptr, 0, ob);
    hipLaunchKernelGGL(k_reloc_accept, g, dim3(MT), 0, st, (const int *)ob, (const int *)(ob + 4 * m), (const int *)(ob + 2 * m),
                       (const int *)(ob + m), (const int *)(ob + 3 * m), m, th_reloc, nnratio, w.d<int>(o_mk), w.d<int>(o_nm));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(orbx::stage_out(w.pin, w.dev + o_res, res_bytes, st));
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(matched_mp, w.pin + (o_mk - o_res), sizeof(int) * n);
    if (nmatches) memcpy(nmatches, w.pin + (o_nm - o_res), sizeof(int));
    if (proj) memcpy(proj, w.pin + (o_proj - o_res), sizeof(float) * 4 * m);
    return ORBX_OK;
}

int search_init_core(FrameSrc &fs, int n2, const orbx_keypoint *kps1, const uint8_t *desc1, int n1, float *prev_matched, int window_size,
                     float nnratio, int check_orientation, int32_t *matches12, int *nmatches)
{
    // queries: level-0 keypoints of F1 around their previous match (:619-626); others get an empty window
    std::vector<WinQuery> q(n1 ? n1 : 1);
    std::vector<float> ang(n1 ? n1 : 1);
    for (int i = 0; i < n1; ++i) {
        const bool use = !(kps1[i].octave > 0);
        q[i] = {prev_matched[2 * i], prev_matched[2 * i + 1], use ? (float)window_size : -1.0f, 0.f, kps1[i].octave, kps1[i].octave};
        ang[i] = kps1[i].angle;
    }
    std::vector<int32_t> dummy(n2 ? n2 : 1);
    return run_sequential(1, q.data(), nullptr, desc1, ang.data(), nullptr, n1, fs, n2, nullptr, 0, 45 /* TH_LOW, :38 */, nnratio, 0,
                          check_orientation, dummy.data(), matches12, nmatches);
}

} // namespace

extern "C" {

int orbm_debug_force_sequential_resolver(int on)
{
    return g_force_sequential.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

int orbm_debug_last_resolver_iterations(void) { return g_last_iterations.load(std::memory_order_relaxed); }

int orbm_sorted_frame(const orbx_keypoint *kps, int n, const uint8_t *skip, const float *uright, float min_x, float min_y,
                      float max_x, float max_y, int32_t *perm, int32_t *cell_off, int32_t *nsorted)
{
    if (n < 0 || (n && !kps) || !nsorted || !(max_x > min_x) || !(max_y > min_y)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    if (n > 65535) ORBX_FAIL(ORBX_ERR_CAPACITY, "more than 65,535 keypoints per frame");
    SortedFrame sf;
    std::vector<uint8_t> nodesc((size_t)32 * (n ? n : 1), 0);
    sort_frame(kps, nodesc.data(), n, skip, uright, min_x, min_y, max_x, max_y, sf);
    *nsorted = (int)sf.perm.size();
    if (perm) memcpy(perm, sf.perm.data(), sizeof(int) * sf.perm.size());
    if (cell_off) memcpy(cell_off, sf.cell_off.data(), sizeof(int) * sf.cell_off.size());
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------------------ resident frames

int orbm_frame_create(const orbx_keypoint *kps, const uint8_t *desc, int n, const float *uright, float min_x, float min_y, float max_x,
                      float max_y, orbm_frame **out)
{
    if (!out || n < 0 || (n && (!kps || !desc)) || !(max_x > min_x) || !(max_y > min_y)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    if (n > FB_MAXN) ORBX_FAIL(ORBX_ERR_CAPACITY, "more than 8,192 keypoints in a resident frame");
    ORBX_NEED_DEVICE();
    orbm_frame *f = new orbm_frame();
    f->min_x = min_x; f->min_y = min_y; f->max_x = max_x; f->max_y = max_y; f->has_uright = uright ? 1 : 0;
    f->gp = {min_x, min_y, (float)FRAME_GRID_COLS / (max_x - min_x), (float)FRAME_GRID_ROWS / (max_y - min_y)};
    if (frame_alloc(f, n)) { frame_release(f); ORBX_FAIL(ORBX_ERR_HIP, "frame allocation failed"); }
    WorkspaceLease lease;
    Workspace &w = *lease.w;
    w.used = 0;
    const size_t m = n ? (size_t)n : 1;
    const size_t o_k = w.carve(sizeof(orbx_keypoint) * m), o_d = w.carve(32 * m), o_u = w.carve(sizeof(float) * m);
    const size_t staged = w.used;
    if (w.reserve(staged, staged + sizeof(int) * m + 1024)) { frame_release(f); ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed"); }
    if (n) {
        memcpy(w.h<char>(o_k), kps, sizeof(orbx_keypoint) * m);
        memcpy(w.h<char>(o_d), desc, 32 * m);
        if (uright) memcpy(w.h<char>(o_u), uright, sizeof(float) * m);
    }
    int rc = orbx::stage_in(w.dev, w.pin, staged, w.st) == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
    if (rc == ORBX_OK)
        rc = frame_build(f, w, w.d<orbx_keypoint>(o_k), w.d<uint4>(o_d), nullptr, uright ? w.d<float>(o_u) : nullptr, nullptr, n, n ? n : 1);
    if (rc != ORBX_OK) { frame_release(f); return rc; }
    *out = f;
    return ORBX_OK;
}

int orbm_frame_from_extractor(orbx_extractor *ex, int frame, const float *xy_undistorted, const float *uright, int uright_from_stereo,
                              float min_x, float min_y, float max_x, float max_y, orbm_frame **out)
{
    if (!out || !ex || frame < 0 || frame >= ex->last_batch || !(max_x > min_x) || !(max_y > min_y) || (uright && uright_from_stereo))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    if (ex->kcap > FB_MAXN) ORBX_FAIL(ORBX_ERR_CAPACITY, "more than 8,192 keypoints in a resident frame");
    if (uright_from_stereo && (!ex->d_uright || frame >= ex->st_batch)) ORBX_FAIL(ORBX_ERR_ARG, "no stereo match on this handle");
    ORBX_NEED_DEVICE();
    const int cap = ex->kcap;
    orbm_frame *f = new orbm_frame();
    f->min_x = min_x; f->min_y = min_y; f->max_x = max_x; f->max_y = max_y; f->has_uright = (uright || uright_from_stereo) ? 1 : 0;
    f->gp = {min_x, min_y, (float)FRAME_GRID_COLS / (max_x - min_x), (float)FRAME_GRID_ROWS / (max_y - min_y)};
    if (frame_alloc(f, cap)) { frame_release(f); ORBX_FAIL(ORBX_ERR_HIP, "frame allocation failed"); }
    WorkspaceLease lease;
    Workspace &w = *lease.w;
    w.used = 0;
    const size_t o_xy = w.carve(xy_undistorted ? sizeof(float) * 2 * (size_t)cap : 1), o_u = w.carve(uright ? sizeof(float) * (size_t)cap : 1);
    const size_t staged = w.used;
    if (w.reserve(staged, staged + sizeof(int) * (size_t)cap + 1024)) { frame_release(f); ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed"); }
    // the caller's arrays hold n entries, n = the frame's count: known here only after the extraction has finished -- it has, for
    // the caller to hold undistorted coordinates -- so read it back first in that case
    int n_known = -1;
    const int *d_n = ex->d_counts + frame;
    const uint8_t *d_desc = ex->d_desc + (size_t)frame * cap * 32;
    orbx_extractor *producer = orbx_detail::order_after_producer(d_desc, w.st);
    int rc = ORBX_OK;
    if (xy_undistorted || uright) {
        if (hipMemcpyAsync(&n_known, d_n, sizeof(int), hipMemcpyDeviceToHost, w.st) != hipSuccess || hipStreamSynchronize(w.st) != hipSuccess) rc = ORBX_ERR_HIP;
        if (rc == ORBX_OK && (n_known < 0 || n_known > cap)) rc = ORBX_ERR_HIP;
        if (rc == ORBX_OK) {
            if (xy_undistorted) memcpy(w.h<char>(o_xy), xy_undistorted, sizeof(float) * 2 * (size_t)n_known);
            if (uright) memcpy(w.h<char>(o_u), uright, sizeof(float) * (size_t)n_known);
            if (orbx::stage_in(w.dev, w.pin, staged, w.st) != hipSuccess) rc = ORBX_ERR_HIP;
        }
    }
    if (rc == ORBX_OK)
        rc = frame_build(f, w, ex->d_kps + (size_t)frame * cap, reinterpret_cast<const uint4 *>(d_desc),
                         xy_undistorted ? w.d<float2>(o_xy) : nullptr,
                         uright ? w.d<float>(o_u) : (uright_from_stereo ? ex->d_uright + (size_t)frame * cap : nullptr), d_n, 0, cap);
    (void)producer;      // frame_build has waited for the stream: nothing of this call still reads the extractor's buffers
    if (rc != ORBX_OK) { frame_release(f); if (rc == ORBX_ERR_HIP) ORBX_FAIL(ORBX_ERR_HIP, "building the frame failed"); return rc; }
    *out = f;
    return ORBX_OK;
}

int orbm_frame_destroy(orbm_frame *f)
{
    if (!f) return ORBX_OK;
    frame_release(f);
    return ORBX_OK;
}

int orbm_frame_alias(const orbm_frame *src, float min_x, float min_y, float max_x, float max_y, orbm_frame **out)
{
    if (!src || !out || !(max_x > min_x) || !(max_y > min_y)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    orbm_frame *f = new orbm_frame();
    f->n = src->n; f->ns = src->ns; f->cap = src->cap; f->has_uright = src->has_uright;
    f->min_octave = src->min_octave; f->max_octave = src->max_octave;
    f->min_x = min_x; f->min_y = min_y; f->max_x = max_x; f->max_y = max_y;
    f->gp = {min_x, min_y, src->gp.inv_w, src->gp.inv_h};        // KeyFrame: int bounds, the Frame's mfGridElement*Inv (KeyFrame.cc:36,44)
    f->block = src->block; f->refs = src->refs;
    f->refs->fetch_add(1);
    frame_pointers(f);
    f->perm_host = src->perm_host; f->inv_host = src->inv_host;
    *out = f;
    return ORBX_OK;
}

int orbm_frame_size(const orbm_frame *f, int *n, int *nsorted)
{
    if (!f) ORBX_FAIL(ORBX_ERR_ARG, "null frame");
    if (n) *n = f->n;
    if (nsorted) *nsorted = f->ns;
    return ORBX_OK;
}

int orbm_frame_layout(const orbm_frame *f, int32_t *perm, int32_t *cell_off)
{
    if (!f) ORBX_FAIL(ORBX_ERR_ARG, "null frame");
    if (perm) memcpy(perm, f->perm_host.data(), sizeof(int) * (size_t)f->ns);
    if (cell_off) {   // (debug accessor; on a leased stream like every other call)
        WorkspaceLease lease;
        const hipStream_t st = lease.w->own_stream();
        if (!st) ORBX_FAIL(ORBX_ERR_HIP, "hipStreamCreate failed");
        ORBX_HIP(hipMemcpyAsync(cell_off, f->cell_off, sizeof(int) * (FB_NC + 1), hipMemcpyDeviceToHost, st));
        ORBX_HIP(hipStreamSynchronize(st));
    }
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------------------ window searches

int orbm_search_window(const orbm_window_query *queries, const uint8_t *qdesc, int nq, const orbx_keypoint *kps,
                       const uint8_t *desc, int n, const uint8_t *skip, const float *uright, float min_x, float min_y,
                       float max_x, float max_y, int init_dist, int32_t *best, int32_t *best_level, int32_t *second,
                       int32_t *second_level, int32_t *idx)
{
    if (nq < 0 || n < 0 || n > 65535 || (nq && (!queries || !qdesc || !best || !best_level || !second || !second_level || !idx)) ||
        (n && (!kps || !desc)) || !(max_x > min_x) || !(max_y > min_y))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (nq == 0) return ORBX_OK;
    SortedFrame sf;
    sort_frame(kps, desc, n, nullptr, uright, min_x, min_y, max_x, max_y, sf);
    FrameSrc fs(sf);
    return search_window_core(fs, queries, qdesc, nq, skip, uright ? 1 : 0, init_dist, nullptr, 0, 0, best, best_level, second, second_level, idx);
}

int orbm_frame_search_window(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, int nq, const uint8_t *skip,
                             int init_dist, int32_t *best, int32_t *best_level, int32_t *second, int32_t *second_level, int32_t *idx)
{
    if (!frame || nq < 0 || (nq && (!queries || !qdesc || !best || !best_level || !second || !second_level || !idx)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (nq == 0) return ORBX_OK;
    FrameSrc fs(frame);
    return search_window_core(fs, queries, qdesc, nq, skip, frame->has_uright, init_dist, nullptr, 0, 0, best, best_level, second, second_level, idx);
}

int orbm_search_by_projection_map(const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *has_mappoint,
                                  const float *mp_pos, const float *mp_normal, const float *mp_min_dist,
                                  const float *mp_max_dist, const uint8_t *mp_desc, int m, const double *Rcw,
                                  const double *tcw, const orbm_camera *cam, const float *scale_factors, int nlevels,
                                  float th, float nnratio, int th_reloc, int32_t *matched_mp, int *nmatches, float *proj)
{
    if (n < 0 || m < 0 || n > 65535 || nlevels < 1 || nlevels > 64 || !cam || !Rcw || !tcw || !scale_factors || !matched_mp ||
        (n && (!kps || !desc)) || (m && (!mp_pos || !mp_normal || !mp_min_dist || !mp_max_dist || !mp_desc)) ||
        !(cam->grid_max_x > cam->grid_min_x) || !(cam->grid_max_y > cam->grid_min_y))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    for (int j = 0; j < n; ++j) matched_mp[j] = -1;
    if (nmatches) *nmatches = 0;
    if (n == 0 || m == 0) return ORBX_OK;
    SortedFrame sf;
    sort_frame(kps, desc, n, nullptr, nullptr, cam->grid_min_x, cam->grid_min_y, cam->grid_max_x, cam->grid_max_y, sf);
    FrameSrc fs(sf);
    return search_map_core(fs, n, has_mappoint, mp_pos, mp_normal, mp_min_dist, mp_max_dist, mp_desc, m, Rcw, tcw, cam, scale_factors, nlevels,
                           th, nnratio, th_reloc, matched_mp, nmatches, proj);
}

int orbm_frame_search_by_projection_map(const orbm_frame *frame, const uint8_t *has_mappoint, const float *mp_pos, const float *mp_normal,
                                        const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc, int m, const double *Rcw,
                                        const double *tcw, const orbm_camera *cam, const float *scale_factors, int nlevels, float th,
                                        float nnratio, int th_reloc, int32_t *matched_mp, int *nmatches, float *proj)
{
    if (!frame || m < 0 || nlevels < 1 || nlevels > 64 || !cam || !Rcw || !tcw || !scale_factors || !matched_mp ||
        (m && (!mp_pos || !mp_normal || !mp_min_dist || !mp_max_dist || !mp_desc)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    const int n = frame->n;
    for (int j = 0; j < n; ++j) matched_mp[j] = -1;
    if (nmatches) *nmatches = 0;
    if (n == 0 || m == 0) return ORBX_OK;
    FrameSrc fs(frame);
    return search_map_core(fs, n, has_mappoint, mp_pos, mp_normal, mp_min_dist, mp_max_dist, mp_desc, m, Rcw, tcw, cam, scale_factors, nlevels,
                           th, nnratio, th_reloc, matched_mp, nmatches, proj);
}

int orbm_project_points(int mode, const float *mp_pos, const float *mp_normal, const float *mp_min_distance,
                        const float *mp_max_distance, int m, const float *Rcw, const float *tcw, const float *Ow,
                        const orbm_camera *cam, float mbf, float viewing_cos_limit, float log_scale_factor,
                        const float *scale_factors, int nlevels, float th, orbm_projected_point *out, orbm_window_query *queries)
{
    if (mode < 0 || mode > 2 || m < 0 || nlevels < 1 || nlevels > 64 || !cam || !Rcw || !tcw || !Ow || !scale_factors || !out ||
        (m && (!mp_pos || !mp_normal || !mp_min_distance || !mp_max_distance)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (m == 0) return ORBX_OK;
    StagedCall sc;
    const size_t o_pos = sc.in(mp_pos, sizeof(float) * 3 * m), o_nrm = sc.in(mp_normal, sizeof(float) * 3 * m),
                 o_min = sc.in(mp_min_distance, sizeof(float) * m), o_max = sc.in(mp_max_distance, sizeof(float) * m),
                 o_sc = sc.in(scale_factors, sizeof(float) * nlevels);
    const size_t o_out = sc.out(sizeof(orbm_projected_point) * (size_t)m), o_q = sc.out(sizeof(WinQuery) * (size_t)m);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    ProjectCam pc;
    pc.fx = cam->fx; pc.fy = cam->fy; pc.cx = cam->cx; pc.cy = cam->cy;
    // Frame::isInFrustum compares with the grid bounds mnMinX..mnMaxY (static floats), KeyFrame::IsInImage likewise
    pc.min_x = cam->grid_min_x; pc.max_x = cam->grid_max_x; pc.min_y = cam->grid_min_y; pc.max_y = cam->grid_max_y;
    pc.mbf = mbf; pc.cos_limit = viewing_cos_limit; pc.log_scale = log_scale_factor; pc.th = th;
    for (int i = 0; i < 9; ++i) pc.R[i] = Rcw[i];
    for (int i = 0; i < 3; ++i) { pc.t[i] = tcw[i]; pc.Ow[i] = Ow[i]; }
    pc.nlevels = nlevels; pc.mode = mode;
    hipLaunchKernelGGL(k_project_points, dim3((m + MT - 1) / MT), dim3(MT), 0, sc.stream(), (const uint8_t *)nullptr, sc.d<const float>(o_pos),
                       sc.d<const float>(o_nrm), sc.d<const float>(o_min), sc.d<const float>(o_max), m, pc, sc.d<const float>(o_sc),
                       sc.d<orbm_projected_point>(o_out), sc.d<WinQuery>(o_q), StageJob{nullptr, nullptr, 0, 0});
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(out, sc.r<orbm_projected_point>(o_out), sizeof(orbm_projected_point) * (size_t)m);
    if (queries) memcpy(queries, sc.r<WinQuery>(o_q), sizeof(WinQuery) * (size_t)m);
    return ORBX_OK;
}

int orbm_search_fuse(const orbm_window_query *queries, const uint8_t *qdesc, int nq, const orbx_keypoint *kps, const uint8_t *desc,
                     int n, const float *uright, const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float max_x,
                     float max_y, int32_t *best, int32_t *idx)
{
    if (nq < 0 || n < 0 || n > 65535 || (nq && (!queries || !qdesc || !best || !idx)) || (n && (!kps || !desc)) ||
        (inv_level_sigma2 && (nlevels < 1 || nlevels > 64)) || !(max_x > min_x) || !(max_y > min_y))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    for (int j = 0; j < n && inv_level_sigma2; ++j)
        if (kps[j].octave < 0 || kps[j].octave >= nlevels) ORBX_FAIL(ORBX_ERR_ARG, "octave out of range");
    ORBX_NEED_DEVICE();
    if (nq == 0) return ORBX_OK;
    SortedFrame sf;
    sort_frame(kps, desc, n, nullptr, uright, min_x, min_y, max_x, max_y, sf);
    if (!uright) for (SeqKp &k : sf.kp) k.uright = -1.0f;
    FrameSrc fs(sf);
    return search_window_core(fs, queries, qdesc, nq, nullptr, uright ? 1 : 0, 256, inv_level_sigma2, nlevels, 1, best, nullptr, nullptr, nullptr, idx);
}

int orbm_frame_search_fuse(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, int nq,
                           const float *inv_level_sigma2, int nlevels, int32_t *best, int32_t *idx)
{
    if (!frame || nq < 0 || (nq && (!queries || !qdesc || !best || !idx)) || (inv_level_sigma2 && (nlevels < 1 || nlevels > 16)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    if (nq == 0) return ORBX_OK;
    FrameSrc fs(frame);
    return search_window_core(fs, queries, qdesc, nq, nullptr, frame->has_uright, 256, inv_level_sigma2, nlevels, 1, best, nullptr, nullptr, nullptr, idx);
}

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
    sort_frame(kps, desc, n, nullptr, uright, min_x, min_y, max_x, max_y, sf);
    FrameSrc fs(sf);
    return run_sequential(0, reinterpret_cast<const WinQuery *>(queries), nullptr, qdesc, qangle, qtakes, nq, fs, n, occupied, uright ? 1 : 0,
                          th_accept, nnratio, ratio_same_level ? ACCEPT_RATIO_SAME_LEVEL : ACCEPT_BEST, check_orientation, match_kp,
                          match_q, nmatches);
}

int orbm_frame_search_projection(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, const float *qangle,
                                 const uint8_t *qtakes, int nq, const uint8_t *occupied, int th_accept, float nnratio, int ratio_same_level,
                                 int check_orientation, int32_t *match_kp, int32_t *match_q, int *nmatches)
{
    if (!frame || nq < 0 || (nq && (!queries || !qdesc || !match_q)) || (frame->n && !match_kp) || !nmatches || (check_orientation && nq && !qangle))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    FrameSrc fs(frame);
    return run_sequential(0, reinterpret_cast<const WinQuery *>(queries), nullptr, qdesc, qangle, qtakes, nq, fs, frame->n, occupied,
                          frame->has_uright, th_accept, nnratio, ratio_same_level ? ACCEPT_RATIO_SAME_LEVEL : ACCEPT_BEST, check_orientation,
                          match_kp, match_q, nmatches);
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
    FrameSrc fs(sf);
    const int rc = search_init_core(fs, n2, kps1, desc1, n1, prev_matched, window_size, nnratio, check_orientation, matches12, nmatches);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < n1; ++i) // :715-718
        if (matches12[i] >= 0) { prev_matched[2 * i] = kps2[matches12[i]].x; prev_matched[2 * i + 1] = kps2[matches12[i]].y; }
    return ORBX_OK;
}

int orbm_frame_search_for_initialization(const orbm_frame *frame2, const orbx_keypoint *kps2, const orbx_keypoint *kps1, const uint8_t *desc1,
                                         int n1, float *prev_matched, int window_size, float nnratio, int check_orientation,
                                         int32_t *matches12, int *nmatches)
{
    if (!frame2 || n1 < 0 || (frame2->n && !kps2) || (n1 && (!kps1 || !desc1 || !prev_matched || !matches12)) || !nmatches)
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    FrameSrc fs(frame2);
    const int rc = search_init_core(fs, frame2->n, kps1, desc1, n1, prev_matched, window_size, nnratio, check_orientation, matches12, nmatches);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < n1; ++i)
        if (matches12[i] >= 0) { prev_matched[2 * i] = kps2[matches12[i]].x; prev_matched[2 * i + 1] = kps2[matches12[i]].y; }
    return ORBX_OK;
}

} // extern "C"

namespace {

// The co-iteration of the two FeatureVectors (ORBmatcher.cc:384-456 / :745-828: equal keys: visit; else lower_bound on the other
// map) as lists: every valid feature of a common node is a query whose candidates are that node's valid members in the other set, one
// copy per node.  inv2: candidate numbers are positions in a resident frame instead of feature indices.
struct BowLists {
    std::vector<int32_t> qidx, cand_off, cand_beg, cand, seg;
    bool disjoint = true;   // a feature sits in one node of a FeatureVector; arrays that break this resolve as one segment
};
void bow_lists(const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1, const uint8_t *valid1, const int32_t *nodes2,
               const int32_t *off2, const int32_t *items2, int nn2, const uint8_t *valid2, int n2, const int *inv2, BowLists &L)
{
    const size_t m1 = nn1 ? (size_t)off1[nn1] : 0, m2 = nn2 ? (size_t)off2[nn2] : 0;
    L.qidx.reserve(m1); L.cand_beg.reserve(m1); L.cand_off.reserve(m1 + 1); L.cand.reserve(m2); L.seg.reserve((size_t)std::min(nn1, nn2) + 1);
    L.cand_off.assign(1, 0); L.seg.assign(1, 0);
    std::vector<uint8_t> seen(n2 ? n2 : 1, 0);
    for (int a = 0, b = 0; a < nn1 && b < nn2;) {
        if (nodes1[a] == nodes2[b]) {
            const size_t c0 = L.cand.size();
            for (int k = off2[b]; k < off2[b + 1]; ++k) { L.disjoint = L.disjoint && !seen[items2[k]]; seen[items2[k]] = 1; }
            for (int k = off2[b]; k < off2[b + 1]; ++k)
                if (!valid2 || valid2[items2[k]]) L.cand.push_back(inv2 ? inv2[items2[k]] : items2[k]);
            const int32_t len = (int32_t)(L.cand.size() - c0);
            bool any = false;
            for (int k = off1[a]; k < off1[a + 1]; ++k) { // every query of the node: the same members, one copy
                if (!valid1[items1[k]]) continue;
                any = true;
                L.qidx.push_back(items1[k]);
                L.cand_beg.push_back((int32_t)c0);
                L.cand_off.push_back(L.cand_off.back() + len);
            }
            if (!any) L.cand.resize(c0);
            else L.seg.push_back((int32_t)L.qidx.size());   // one segment per node
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) ++a;
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) ++b;
        }
    }
}
int bow_check_items(const int32_t *off, const int32_t *items, int nn, int n)
{
    for (int k = 0; k < (nn ? off[nn] : 0); ++k)
        if (items[k] < 0 || items[k] >= n) return -1;
    return 0;
}

} // namespace

extern "C" {

int orbm_search_by_bow(const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1, const uint8_t *valid1,
                       const uint8_t *desc1, const float *angle1, int n1, const int32_t *nodes2, const int32_t *off2,
                       const int32_t *items2, int nn2, const uint8_t *valid2, const uint8_t *desc2, const float *angle2, int n2,
                       int th, int strict_th, float nnratio, int check_orientation, int32_t *match12, int32_t *match21,
                       int *nmatches)
{
    if (n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || (n1 && (!desc1 || !match12 || !valid1)) || (n2 && !desc2) ||
        (nn1 && (!nodes1 || !off1 || !items1)) || (nn2 && (!nodes2 || !off2 || !items2)) || !nmatches ||
        (check_orientation && n1 && n2 && (!angle1 || !angle2)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    if (bow_check_items(off1, items1, nn1, n1) || bow_check_items(off2, items2, nn2, n2)) ORBX_FAIL(ORBX_ERR_ARG, "feature index out of range");
    ORBX_NEED_DEVICE();
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    if (match21) for (int j = 0; j < n2; ++j) match21[j] = -1;
    *nmatches = 0;
    BowLists L;
    bow_lists(nodes1, off1, items1, nn1, valid1, nodes2, off2, items2, nn2, valid2, n2, nullptr, L);
    const int nq = (int)L.qidx.size();
    if (nq == 0 || n2 == 0) return ORBX_OK;
    SortedFrame sf; // the second set as it is: position = feature index
    sf.perm.resize(n2); sf.angle.resize(n2); sf.desc.assign(desc2, desc2 + (size_t)32 * n2);
    for (int j = 0; j < n2; ++j) { sf.perm[j] = j; sf.angle[j] = angle2 ? angle2[j] : 0.f; }
    std::vector<uint8_t> qd((size_t)32 * nq);
    std::vector<float> qa(nq);
    for (int i = 0; i < nq; ++i) { memcpy(&qd[32 * (size_t)i], desc1 + 32 * (size_t)L.qidx[i], 32); qa[i] = angle1 ? angle1[L.qidx[i]] : 0.f; }
    std::vector<int32_t> mk(n2), mq(nq);
    FrameSrc fs(sf);
    // bestDist1 < TH_LOW (:799) == bestDist1 <= TH_LOW - 1
    const int rc = run_sequential(0, nullptr, nullptr, qd.data(), qa.data(), nullptr, nq, fs, n2, nullptr, 0, strict_th ? th - 1 : th, nnratio,
                                  ACCEPT_RATIO, check_orientation, mk.data(), mq.data(), nmatches, nullptr, L.cand_off.data(), L.cand_beg.data(), L.cand.data(),
                                  (int)L.cand.size(), L.disjoint ? L.seg.data() : nullptr, L.disjoint ? (int)L.seg.size() - 1 : 0);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < nq; ++i) // every accepted query blocks its candidate, so slot mq[i] still names i unless rejected
        if (mq[i] >= 0 && mk[mq[i]] == i) { match12[L.qidx[i]] = mq[i]; if (match21) match21[mq[i]] = L.qidx[i]; }
    return ORBX_OK;
}

int orbm_frame_search_by_bow(const orbm_frame *frame1, const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1,
                             const uint8_t *valid1, const orbm_frame *frame2, const int32_t *nodes2, const int32_t *off2,
                             const int32_t *items2, int nn2, const uint8_t *valid2, int th, int strict_th, float nnratio,
                             int check_orientation, int32_t *match12, int32_t *match21, int *nmatches)
{
    if (!frame1 || !frame2 || nn1 < 0 || nn2 < 0 || (frame1->n && (!match12 || !valid1)) || (nn1 && (!nodes1 || !off1 || !items1)) ||
        (nn2 && (!nodes2 || !off2 || !items2)) || !nmatches)
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    const int n1 = frame1->n, n2 = frame2->n;
    if (bow_check_items(off1, items1, nn1, n1) || bow_check_items(off2, items2, nn2, n2)) ORBX_FAIL(ORBX_ERR_ARG, "feature index out of range");
    ORBX_NEED_DEVICE();
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    if (match21) for (int j = 0; j < n2; ++j) match21[j] = -1;
    *nmatches = 0;
    BowLists L;
    bow_lists(nodes1, off1, items1, nn1, valid1, nodes2, off2, items2, nn2, valid2, n2, frame2->inv_host.data(), L);
    const int nq = (int)L.qidx.size();
    if (nq == 0 || n2 == 0) return ORBX_OK;
    std::vector<int32_t> qsrc(nq), mk(n2), mq(nq);
    for (int i = 0; i < nq; ++i) qsrc[i] = frame1->inv_host[L.qidx[i]];
    FrameSrc fs(frame2, true);
    const int rc = run_sequential(0, nullptr, nullptr, nullptr, nullptr, nullptr, nq, fs, n2, nullptr, 0, strict_th ? th - 1 : th, nnratio,
                                  ACCEPT_RATIO, check_orientation, mk.data(), mq.data(), nmatches, nullptr, L.cand_off.data(), L.cand_beg.data(), L.cand.data(),
                                  (int)L.cand.size(), L.disjoint ? L.seg.data() : nullptr, L.disjoint ? (int)L.seg.size() - 1 : 0, frame1, qsrc.data());
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < nq; ++i)
        if (mq[i] >= 0 && mk[mq[i]] == i) { match12[L.qidx[i]] = mq[i]; if (match21) match21[mq[i]] = L.qidx[i]; }
    return ORBX_OK;
}

int orbm_frame_search_for_triangulation(const orbm_frame *kf1, const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1,
                                        const uint8_t *has_mappoint1, const orbm_frame *kf2, const int32_t *nodes2, const int32_t *off2,
                                        const int32_t *items2, int nn2, const uint8_t *has_mappoint2, int only_stereo, const float *F12,
                                        float ex, float ey, const float *scale_factors2, const float *level_sigma2, int nlevels,
                                        int check_orientation, int32_t *match12, int *nmatches)
{
    if (!kf1 || !kf2 || nn1 < 0 || nn2 < 0 || nlevels < 1 || !nmatches || !F12 || !scale_factors2 || !level_sigma2 ||
        (kf1->n && (!match12 || !has_mappoint1)) || (kf2->n && !has_mappoint2) || (nn1 && (!nodes1 || !off1 || !items1)) ||
        (nn2 && (!nodes2 || !off2 || !items2)))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    const int n1 = kf1->n, n2 = kf2->n;
    if (bow_check_items(off1, items1, nn1, n1) || bow_check_items(off2, items2, nn2, n2)) ORBX_FAIL(ORBX_ERR_ARG, "feature index out of range");
    if (n2 && (kf2->min_octave < 0 || kf2->max_octave >= nlevels)) ORBX_FAIL(ORBX_ERR_ARG, "octave out of range");
    *nmatches = 0;
    if (n1 == 0) return ORBX_OK;
    ORBX_NEED_DEVICE();
    // the co-iteration of :881-891 / :1004-1012: every keypoint of a common node in KF1 scans the node's members of KF2, in member
    // order -- the member lists once, as positions in KF2's arrays
    const int nit2 = nn2 ? off2[nn2] : 0;
    std::vector<int32_t> cbeg((size_t)n1, 0), clen((size_t)n1, 0), cand((size_t)std::max(nit2, 1));
    for (int k = 0; k < nit2; ++k) cand[k] = kf2->inv_host[items2[k]];
    for (int a = 0, b = 0; a < nn1 && b < nn2;) {
        if (nodes1[a] == nodes2[b]) {
            if (off2[b + 1] - off2[b] > 65535) ORBX_FAIL(ORBX_ERR_CAPACITY, "more than 65,535 candidates for one keypoint (the tie rule's position field)");
            for (int k = off1[a]; k < off1[a + 1]; ++k) { cbeg[items1[k]] = off2[b]; clen[items1[k]] = off2[b + 1] - off2[b]; }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) ++a;
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) ++b;
        }
    }
    StagedCall sc;
    const size_t o_cb = sc.in(cbeg.data(), sizeof(int) * (size_t)n1), o_cl = sc.in(clen.data(), sizeof(int) * (size_t)n1),
                 o_ca = sc.in(cand.data(), sizeof(int) * cand.size()), o_m1 = sc.in(has_mappoint1, (size_t)n1),
                 o_m2 = sc.in(has_mappoint2, (size_t)n2), o_sc = sc.in(scale_factors2, sizeof(float) * (size_t)nlevels),
                 o_sg = sc.in(level_sigma2, sizeof(float) * (size_t)nlevels), o_o = sc.out(sizeof(int) * 2 * (size_t)n1);
    if (sc.upload()) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation / upload failed");
    TriForm tp;
    for (int i = 0; i < 9; ++i) tp.F12[i] = F12[i];
    tp.ex = ex; tp.ey = ey; tp.only_stereo = only_stereo ? 1 : 0;
    int *om = sc.d<int>(o_o);
    hipLaunchKernelGGL(k_triang_frames, dim3((unsigned)(((size_t)n1 * 16 + MT - 1) / MT)), dim3(MT), 0, sc.stream(), (const SeqKp *)kf1->kp,
                       (const uint4 *)kf1->desc, (const float *)kf1->angle, (const int *)kf1->perm, n1, (const SeqKp *)kf2->kp,
                       (const uint4 *)kf2->desc, (const float *)kf2->angle, (const int *)kf2->perm, sc.d<const int>(o_cb),
                       sc.d<const int>(o_cl), sc.d<const int>(o_ca), sc.d<const uint8_t>(o_m1), sc.d<const uint8_t>(o_m2), tp,
                       sc.d<const float>(o_sc), sc.d<const float>(o_sg), om, reinterpret_cast<float *>(om + n1));
    ORBX_HIP(hipGetLastError());
    if (sc.download()) ORBX_FAIL(ORBX_ERR_HIP, "download failed");
    memcpy(match12, sc.r<int>(o_o), sizeof(int) * (size_t)n1);
    if (check_orientation) {
        const float *rot = reinterpret_cast<const float *>(sc.r<int>(o_o) + n1);
        constexpr int HL = 30;     // HISTO_LENGTH, ORBmatcher.cc:40
        int hist[HL] = {0};
        std::vector<int> bin((size_t)n1, -1);
        const float factor = 1.0f / HL;
        for (int i = 0; i < n1; ++i)
            if (match12[i] >= 0) {
                float r = rot[i];                                      // :994-1001
                if (r < 0.0f) r += 360.0f;
                int b = (int)roundf(r * factor);
                if (b == HL) b = 0;
                if (b < 0 || b > HL) ORBX_FAIL(ORBX_ERR_ARG, "keypoint angles outside [0, 360)");   // (the reference asserts)
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

// ------------------------------------------------------- the projection searches as whole functions, on resident frames

int orbm_search_by_projection_points(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const orbm_points *points,
                                     const uint8_t *occupied, float th, float viewing_cos_limit, int th_high, float nnratio, int32_t *match_kp,
                                     int32_t *match_q, int *nmatches, orbm_projected_point *projected_out, orbm_window_query *queries_out)
{
    if (!cur || bad_view(view) || !Tcw || bad_points(points, true, true, false, view->nlevels) || !nmatches || (points->n && !match_q) ||
        (cur->n && !match_kp))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    PointsPrefix px;
    px.pts = points; px.scale = view->scale_factors; px.frustum = true; px.proj_host = projected_out;
    px.cam.nlevels = view->nlevels;
    ProjectCam &pc = px.pcam;
    pc.fx = view->fx; pc.fy = view->fy; pc.cx = view->cx; pc.cy = view->cy;
    pc.min_x = cur->min_x; pc.max_x = cur->max_x; pc.min_y = cur->min_y; pc.max_y = cur->max_y;
    pc.mbf = view->mbf; pc.cos_limit = viewing_cos_limit; pc.log_scale = view->log_scale_factor; pc.th = th;
    pose_parts(Tcw, pc.R, pc.t);
    neg_Rt_t(pc.R, pc.t, pc.Ow);                 // mOw = -mRcw.t() * mtcw (Frame::UpdatePoseMatrices, src/Frame.cc:236-242)
    pc.nlevels = view->nlevels; pc.mode = ORBM_PROJECT_FRUSTUM;
    if (projected_out) for (int i = 0; i < points->n; ++i) projected_out[i] = {0.f, 0.f, 0.f, 0.f, 0.f, -1, 0};
    FrameSrc fs(cur);
    return run_sequential(0, nullptr, &px, points->desc, nullptr, points->takes, points->n, fs, cur->n, occupied, cur->has_uright, th_high,
                          nnratio, ACCEPT_RATIO_SAME_LEVEL, 0, match_kp, match_q, nmatches, reinterpret_cast<WinQuery *>(queries_out));
}

int orbm_search_by_projection_last(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const float *Tlw, const orbm_points *last,
                                   const uint8_t *occupied, float th, int mono, int th_high, int check_orientation, int32_t *match_kp,
                                   int32_t *match_q, int *nmatches, orbm_window_query *queries_out)
{
    if (!cur || bad_view(view) || !Tcw || !Tlw || bad_points(last, false, false, true, view->nlevels) || !nmatches ||
        (last->n && !match_q) || (cur->n && !match_kp) || (check_orientation && last->n && !last->angle))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    PointsPrefix px;
    px.pts = last; px.scale = view->scale_factors;
    form_cam_common(px.cam, view, cur, FORM_LAST, th);
    float Rlw[9], tlw[3], twc[3], tlc[3];
    pose_parts(Tcw, px.cam.R, px.cam.t);                    // Rcw, tcw (:1539-1540)
    pose_parts(Tlw, Rlw, tlw);
    neg_Rt_t(px.cam.R, px.cam.t, twc);                      // twc = -Rcw.t() * tcw (:1542)
    gemm3(Rlw, twc, 1.0, tlw, 1.0, tlc);                    // tlc = Rlw * twc + tlw (:1547)
    px.cam.forward = tlc[2] > view->mb && !mono;            // :1549-1550
    px.cam.backward = -tlc[2] > view->mb && !mono;
    FrameSrc fs(cur);
    return run_sequential(0, nullptr, &px, last->desc, last->angle, last->takes, last->n, fs, cur->n, occupied, cur->has_uright, th_high, 0.f,
                          ACCEPT_BEST, check_orientation, match_kp, match_q, nmatches, reinterpret_cast<WinQuery *>(queries_out));
}

int orbm_search_by_projection_keyframe(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const orbm_points *kf,
                                       const uint8_t *occupied, float th, int orb_dist, int check_orientation, int32_t *match_kp,
                                       int32_t *match_q, int *nmatches, orbm_window_query *queries_out)
{
    if (!cur || bad_view(view) || !Tcw || bad_points(kf, true, false, false, view->nlevels) || !nmatches || (kf->n && !match_q) ||
        (cur->n && !match_kp) || (check_orientation && kf->n && !kf->angle))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    PointsPrefix px;
    px.pts = kf; px.scale = view->scale_factors;
    form_cam_common(px.cam, view, cur, FORM_KF, th);
    pose_parts(Tcw, px.cam.R, px.cam.t);
    neg_Rt_t(px.cam.R, px.cam.t, px.cam.Ow);                // Ow = -Rcw.t() * tcw (:1679)
    FrameSrc fs(cur);
    // no stereo test in this form, every assignment blocks its slot (:1741-1742)
    return run_sequential(0, nullptr, &px, kf->desc, kf->angle, nullptr, kf->n, fs, cur->n, occupied, 0, orb_dist, 0.f, ACCEPT_BEST,
                          check_orientation, match_kp, match_q, nmatches, reinterpret_cast<WinQuery *>(queries_out));
}

int orbm_search_by_projection_sim3(const orbm_frame *kf, const orbm_view *view, const float *Scw, const orbm_points *points,
                                   const uint8_t *occupied, int th, int th_low, int32_t *match_kp, int32_t *match_q, int *nmatches,
                                   orbm_window_query *queries_out)
{
    if (!kf || bad_view(view) || !Scw || bad_points(points, true, true, false, view->nlevels) || !nmatches || (points->n && !match_q) ||
        (kf->n && !match_kp))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    PointsPrefix px;
    px.pts = points; px.scale = view->scale_factors;
    form_cam_common(px.cam, view, kf, FORM_SIM3, (float)th);
    {   // :499-504: sRcw, scw = sqrt(sRcw.row(0).dot(sRcw.row(0))), Rcw = sRcw / scw, tcw = Scw.col(3) / scw, Ow = -Rcw.t() * tcw
        float sR[9], st[3];
        pose_parts(Scw, sR, st);
        double dot = 0;
        for (int k = 0; k < 3; ++k) dot += (double)sR[k] * (double)sR[k];
        const float scw = (float)sqrt(dot);
        scale_mat(sR, 9, 1. / scw, px.cam.R);
        scale_mat(st, 3, 1. / scw, px.cam.t);
        neg_Rt_t(px.cam.R, px.cam.t, px.cam.Ow);
    }
    FrameSrc fs(kf);
    return run_sequential(0, nullptr, &px, points->desc, nullptr, nullptr, points->n, fs, kf->n, occupied, 0, th_low, 0.f, ACCEPT_BEST, 0,
                          match_kp, match_q, nmatches, reinterpret_cast<WinQuery *>(queries_out));
}

int orbm_search_by_sim3(const orbm_frame *kf1, const orbm_frame *kf2, const orbm_view *view, const float *T1w, const float *T2w, float s12,
                        const float *R12, const float *t12, const orbm_points *points1, const orbm_points *points2, float th, int th_high,
                        int32_t *vnMatch1, int32_t *vnMatch2, int32_t *match12, int *nfound, orbm_window_query *q12_out,
                        orbm_window_query *q21_out)
{
    if (!kf1 || !kf2 || bad_view(view) || !T1w || !T2w || !R12 || !t12 || bad_points(points1, true, false, false, view->nlevels) ||
        bad_points(points2, true, false, false, view->nlevels) || !nfound || points1->n != kf1->n || points2->n != kf2->n ||
        (kf1->n && !match12))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    ORBX_NEED_DEVICE();
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; ++i) { match12[i] = -1; if (vnMatch1) vnMatch1[i] = -1; }
    for (int i = 0; i < n2 && vnMatch2; ++i) vnMatch2[i] = -1;
    *nfound = 0;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    PointsPrefix p12, p21;
    p12.pts = points1; p21.pts = points2; p12.scale = p21.scale = view->scale_factors;
    form_cam_common(p12.cam, view, kf2, FORM_PAIR, th);       // KF1's points into KF2 (:1348-1428)
    form_cam_common(p21.cam, view, kf1, FORM_PAIR, th);       // and back (:1430-1507)
    pose_parts(T1w, p12.cam.R, p12.cam.t);
    pose_parts(T2w, p21.cam.R, p21.cam.t);
    {   // :1320-1323: sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t(), t21 = -sR21 * t12
        float R12t[9];
        scale_mat(R12, 9, (double)s12, p21.cam.R2);
        memcpy(p21.cam.t2, t12, sizeof(float) * 3);
        transpose3(R12, R12t);
        scale_mat(R12t, 9, 1.0 / (double)s12, p12.cam.R2);
        gemm3(p12.cam.R2, t12, -1.0, nullptr, 0.0, p12.cam.t2);
    }
    WorkspaceLease lease;
    Workspace &w = *lease.w;
    w.used = 0;
    const size_t o_a1 = w.carve((size_t)32 * n1), o_a2 = w.carve((size_t)32 * n2);
    const size_t staged = w.used;
    p12.carve(w); p21.carve(w);                    // read by the prefix kernels from the pinned block
    const size_t pin_in = w.used;
    const size_t o_q12 = w.carve(sizeof(WinQuery) * n1), o_q21 = w.carve(sizeof(WinQuery) * n2), o_b1 = w.carve(sizeof(int) * 5 * (size_t)n1),
                 o_b2 = w.carve(sizeof(int) * 5 * (size_t)n2);
    const size_t o_res = w.used;
    const size_t o_v1 = w.carve(sizeof(int) * n1), o_v2 = w.carve(sizeof(int) * n2), o_m = w.carve(sizeof(int) * n1), o_nf = w.carve(sizeof(int));
    const size_t o_qo1 = w.carve(q12_out ? sizeof(WinQuery) * n1 : 1), o_qo2 = w.carve(q21_out ? sizeof(WinQuery) * n2 : 1);
    const size_t res_bytes = w.used - o_res;
    if (w.reserve(w.used, std::max(pin_in, res_bytes))) ORBX_FAIL(ORBX_ERR_HIP, "workspace allocation failed");
    p12.fill(w); p21.fill(w);
    memcpy(w.h<char>(o_a1), points1->desc, (size_t)32 * n1);
    memcpy(w.h<char>(o_a2), points2->desc, (size_t)32 * n2);
    hipStream_t st = w.st;
    ORBX_HIP(hipMemsetAsync(w.d<char>(o_nf), 0, sizeof(int), st));
    if (orbx::stage_ok(w.dev, w.pin, staged)) p12.launch(w, w.d<WinQuery>(o_q12), st, stage_job(w.dev, w.pin, staged));
    else { ORBX_HIP(orbx::stage_in(w.dev, w.pin, staged, st)); p12.launch(w, w.d<WinQuery>(o_q12), st); }
    p21.launch(w, w.d<WinQuery>(o_q21), st);
    FrameSrc f1(kf1), f2(kf2);
    int *b1 = w.d<int>(o_b1), *b2 = w.d<int>(o_b2);
    launch_win_best(st, w.d<WinQuery>(o_q12), w.d<uint4>(o_a1), n1, f2.view(w), nullptr, 0, INT_MAX, nullptr, 0, b1);
    launch_win_best(st, w.d<WinQuery>(o_q21), w.d<uint4>(o_a2), n2, f1.view(w), nullptr, 0, INT_MAX, nullptr, 0, b2);
    hipLaunchKernelGGL(k_sim3_agree, dim3((std::max(n1, n2) + MT - 1) / MT), dim3(MT), 0, st, (const int *)b1, (const int *)(b1 + 4 * n1), n1,
                       (const int *)b2, (const int *)(b2 + 4 * n2), n2, th_high, w.d<int>(o_v1), w.d<int>(o_v2), w.d<int>(o_m), w.d<int>(o_nf));
    ORBX_HIP(hipGetLastError());
    if (q12_out) ORBX_HIP(hipMemcpyAsync(w.d<char>(o_qo1), w.d<char>(o_q12), sizeof(WinQuery) * n1, hipMemcpyDeviceToDevice, st));
    if (q21_out) ORBX_HIP(hipMemcpyAsync(w.d<char>(o_qo2), w.d<char>(o_q21), sizeof(WinQuery) * n2, hipMemcpyDeviceToDevice, st));
    ORBX_HIP(orbx::stage_out(w.pin, w.dev + o_res, res_bytes, st));
    ORBX_HIP(hipStreamSynchronize(st));
    if (vnMatch1) memcpy(vnMatch1, w.pin + (o_v1 - o_res), sizeof(int) * n1);
    if (vnMatch2) memcpy(vnMatch2, w.pin + (o_v2 - o_res), sizeof(int) * n2);
    memcpy(match12, w.pin + (o_m - o_res), sizeof(int) * n1);
    memcpy(nfound, w.pin + (o_nf - o_res), sizeof(int));
    if (q12_out) memcpy(q12_out, w.pin + (o_qo1 - o_res), sizeof(WinQuery) * n1);
    if (q21_out) memcpy(q21_out, w.pin + (o_qo2 - o_res), sizeof(WinQuery) * n2);
    return ORBX_OK;
}

} // extern "C"
