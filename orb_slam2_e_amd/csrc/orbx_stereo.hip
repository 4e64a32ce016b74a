// orbx_stereo.hip -- Frame::ComputeStereoMatches (src/Frame.cc:527-701) on the
// device-resident results of a left and a right orbx_extractor.
//
//   k_stereo_rows     vRowIndices: per image row the right keypoints whose band covers it (:537-554), a CSR table per frame
//   k_stereo_hamming  the row's candidates: octave / disparity gating + best Hamming match (:556-610)
//   k_stereo_refine   11x11 L1 correlation over 11 shifts on the left keypoint's
//                     pyramid level, parabola sub-pixel fit, depth (:612-683)
//   k_stereo_median   median cut 1.5*1.4*median of the correlation distances (:687-700)
//
// Candidate order only matters through "strict <, first index wins", which the
// packed key (dist<<16 | iR) minimum reproduces.  The L1 distances are sums of
// integer-valued floats (exact), so everything up to the parabola is integer work.
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "common.h"
#include "orbx_internal.h"

using namespace orbx_detail;

namespace {

constexpr int ST_T = 256;
struct RightKp { short minr, maxr; int octave; float x; };

__device__ __forceinline__ int hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// vRowIndices (:537-554): for every image row the right keypoints whose band r = 2 * scale[octave], rows floor(y - r) ..
// ceil(y + r), covers it -- as a CSR table per frame, built by ONE workgroup per frame: row counts in LDS (atomics: the
// order inside a row's list is irrelevant, the matcher's key minimum reproduces "first index wins"), scan, fill.  A left
// keypoint then looks at its own row's list only (~50 candidates of 2,000: scanning all right keypoints per left keypoint
// made the row-band match 26 % of a 64-pair batch).
constexpr int ROWS_T = 1024, ROWS_MAX = 4096;   // (orbx_reserve limits an image side to 4,000 px)
__global__ __launch_bounds__(ROWS_T) void k_stereo_rows(const orbx_keypoint *__restrict__ kpR, const int *__restrict__ cntR, int cap,
                                                        const float *__restrict__ scaleFactors, int nRows, int items_cap,
                                                        RightKp *__restrict__ rk, int *__restrict__ rowoff, int *__restrict__ items)
{
    __shared__ int cnt[ROWS_MAX + 1];
    __shared__ int wsum[ROWS_T / 64];
    const int f = blockIdx.x, tid = threadIdx.x, Nr = min(cntR[f], cap);
    for (int r = tid; r <= nRows; r += ROWS_T) cnt[r] = 0;
    __syncthreads();
    for (int j = tid; j < Nr; j += ROWS_T) {
        const orbx_keypoint k = kpR[(size_t)f * cap + j];
        const float r = 2.0f * scaleFactors[k.octave]; // :546
        RightKp o;
        o.maxr = (short)(int)ceilf(k.y + r);
        o.minr = (short)(int)floorf(k.y - r);
        o.octave = k.octave; o.x = k.x;
        rk[(size_t)f * cap + j] = o;
        for (int y = max((int)o.minr, 0); y <= min((int)o.maxr, nRows - 1); ++y) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. nRows] in place: contiguous chunks per thread, wave scan, one hop across waves
    const int per = (nRows + 1 + ROWS_T - 1) / ROWS_T, lo = min(tid * per, nRows + 1), hi = min(lo + per, nRows + 1);
    int sum = 0;
    for (int r = lo; r < hi; ++r) sum += cnt[r];
    const int incl = orbx::wave_incl_scan(sum);
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    int *ro = rowoff + (size_t)f * (nRows + 1);
    for (int r = lo; r < hi; ++r) { const int c = cnt[r]; cnt[r] = base; ro[r] = base; base += c; }   // cnt becomes the fill cursor
    __syncthreads();
    int *it = items + (size_t)f * items_cap;
    for (int j = tid; j < Nr; j += ROWS_T) {
        const RightKp o = rk[(size_t)f * cap + j];
        for (int y = max((int)o.minr, 0); y <= min((int)o.maxr, nRows - 1); ++y) {
            const int pos = atomicAdd(&cnt[y], 1);
            if (pos < items_cap) it[pos] = j;
        }
    }
}

// One wave per left keypoint (4 per workgroup): the lanes stride over vRowIndices[row of the keypoint]; the candidates that
// pass the octave / disparity gates get a Hamming distance; wave minimum of dist << 16 | iR.
__global__ __launch_bounds__(ST_T) void k_stereo_hamming(const orbx_keypoint *__restrict__ kpL, const uint8_t *__restrict__ dL,
                                                         const int *__restrict__ cntL, const RightKp *__restrict__ rkR,
                                                         const uint8_t *__restrict__ dR, const int *__restrict__ rowoff,
                                                         const int *__restrict__ items, int items_cap,
                                                         int cap, int nRows, float maxD, unsigned *__restrict__ best_key)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * (ST_T / 64) + (threadIdx.x >> 6);
    const int N = min(cntL[f], cap);
    if (iL >= N) return;
    const orbx_keypoint k = kpL[(size_t)f * cap + iL];
    const uint4 *A = reinterpret_cast<const uint4 *>(dL + ((size_t)f * cap + iL) * 32);
    const uint4 a0 = A[0], a1 = A[1];
    const int levelL = k.octave;
    const int row = (int)k.y;              // vRowIndices[vL], :569
    const float minU = k.x - maxD;         // :574
    const float maxU = k.x - 0.0f;         // :575 (minD = 0)
    unsigned key = 95u << 16;              // bestDist = TH_HIGH, bestIdxR = 0 (:580-581)
    if (!(row < 0 || row >= nRows || maxU < 0)) { // else no candidates (:571-578)
        const RightKp *rk = rkR + (size_t)f * cap;
        const int *ro = rowoff + (size_t)f * (nRows + 1), *it = items + (size_t)f * items_cap;
        const int k0 = ro[row], k1 = min(ro[row + 1], items_cap);
        for (int q = k0 + lane; q < k1; q += 64) {
            const int j = it[q];
            const RightKp r = rk[j];
            if (r.octave >= levelL - 1 && r.octave <= levelL + 1 && r.x >= minU && r.x <= maxU) {
                const uint4 *Bp = reinterpret_cast<const uint4 *>(dR + ((size_t)f * cap + j) * 32);
                const unsigned k2 = ((unsigned)hamming256(a0, a1, Bp[0], Bp[1]) << 16) | (unsigned)j;
                key = k2 < key ? k2 : key; // dist<bestDist in iR order == min over (dist, iR), :598-602
            }
        }
    }
    key = orbx::wave_min_u32(key);
    if (lane == 0) best_key[(size_t)f * cap + iL] = key;
}

// One wave per left keypoint that found a descriptor match below thOrbDist.
__global__ __launch_bounds__(256) void k_stereo_refine(const orbx_keypoint *__restrict__ kpL, const int *__restrict__ cntL,
                                                       const orbx_keypoint *__restrict__ kpR, int cap,
                                                       const unsigned *__restrict__ best_key,
                                                       const uint8_t *__restrict__ pyrL, const uint8_t *__restrict__ pyrR,
                                                       size_t frame_bytes, const LevelInfo *__restrict__ L,
                                                       const float *__restrict__ scaleFactors, const float *__restrict__ invScaleFactors,
                                                       float maxD, float mbf, float *__restrict__ uRight, float *__restrict__ depth,
                                                       int *__restrict__ sad)
{
    __shared__ uint8_t s_l[4][11 * 11];
    __shared__ uint8_t s_r[4][11 * 21];
    __shared__ int s_p[4][121];
    const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iL = blockIdx.x * 4 + wv;
    const int N = min(cntL[f], cap);
    if (iL >= cap) return;
    const size_t o = (size_t)f * cap + iL;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const int thOrbDist = (95 + 45) / 2; // (TH_HIGH + TH_LOW) / 2, :532
    if (iL < N) {
        const unsigned key = best_key[o];
        if ((int)(key >> 16) < thOrbDist) { // wave-uniform
            const orbx_keypoint kl = kpL[o];
            const int level = kl.octave;
            const float uL = kl.x;
            const float uR0 = kpR[(size_t)f * cap + (key & 0xffffu)].x;
            const float sf = invScaleFactors[level];
            const float scaleduL = roundf(kl.x * sf), scaledvL = roundf(kl.y * sf), scaleduR0 = roundf(uR0 * sf);
            const LevelInfo lv = L[level];
            const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
            if (!(iniu < 0 || endu >= (float)lv.w)) {
                const int xl = (int)scaleduL, yl = (int)scaledvL, xr = (int)scaleduR0;
                const size_t base = (size_t)f * frame_bytes + lv.off + (size_t)(yl - 5 + EDGE) * lv.stride + PADX;
                for (int i = lane; i < 121; i += 64) {
                    const int dy = i / 11, dx = i - 11 * dy;
                    s_l[wv][i] = pyrL[base + (size_t)dy * lv.stride + xl - 5 + dx];
                }
                for (int i = lane; i < 231; i += 64) {
                    const int dy = i / 21, dx = i - 21 * dy;
                    s_r[wv][i] = pyrR[base + (size_t)dy * lv.stride + xr - 10 + dx];
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                // L1 distance for incR = s - 5 (IL, IR minus their centre pixels, :625-645): the 121 (shift s, row dy) row sums over
                // the lanes (two per lane; integers: any summation order), then lanes 0..10 add up their shift's eleven rows
                // (one lane per shift walking all 121 pixels kept 53 of 64 lanes idle for ~700 instructions)
                {
                    const int cL = s_l[wv][5 * 11 + 5];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int t = lane + 64 * h;
                        if (t < 121) {
                            const int sft = t / 11, dy = t - 11 * sft, cR = s_r[wv][5 * 21 + 5 + sft];
                            int acc = 0;
#pragma unroll
                            for (int dx = 0; dx < 11; ++dx) {
                                const int d = ((int)s_l[wv][dy * 11 + dx] - cL) - ((int)s_r[wv][dy * 21 + sft + dx] - cR);
                                acc += d < 0 ? -d : d;
                            }
                            s_p[wv][t] = acc;
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                float dist = 0.0f;
                if (lane < 11) {
                    int acc = 0;
#pragma unroll
                    for (int dy = 0; dy < 11; ++dy) acc += s_p[wv][lane * 11 + dy];
                    dist = (float)acc;
                }
                float vd[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) vd[k] = __shfl(dist, k);
                int bestDist = INT_MAX, bestincR = 0;
#pragma unroll
                for (int k = 0; k < 11; ++k)
                    if (vd[k] < (float)bestDist) { bestDist = (int)vd[k]; bestincR = k - 5; }
                if (!(bestincR == -5 || bestincR == 5)) {
                    float d1 = vd[0], d2 = vd[0], d3 = vd[0];
#pragma unroll
                    for (int k = 1; k < 10; ++k)
                        if (k - 5 == bestincR) { d1 = vd[k - 1]; d2 = vd[k]; d3 = vd[k + 1]; }
                    const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2)); // :666
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = scaleFactors[level] * ((float)scaleduR0 + (float)bestincR + deltaR);
                        float disparity = uL - bestuR;
                        if (disparity >= 0.0f && disparity < maxD) {
                            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
                            out_d = mbf / disparity;
                            out_u = bestuR;
                            out_sad = bestDist;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) { uRight[o] = out_u; depth[o] = out_d; sad[o] = out_sad; }
}

// Median cut (:687-700): median = the (nd/2)-th smallest correlation distance (vDistIdx sorted, element size()/2),
// everything >= 1.5f*1.4f*median is discarded.  One 1024-thread workgroup per frame; the k-th smallest by a two-pass
// radix selection on LDS histograms: an 11 x 11 L1 distance of centred bytes is at most 121 * 510 = 61,710 < 2^16, so
// the high byte's histogram names the bin that holds rank k and the low byte's histogram inside that bin the value
// (the bisection on the value this replaces took 20 rounds of two barriers on four waves: 22 us of a 50-us call).
constexpr int MED_T = 1024;
__device__ __forceinline__ void med_pick(int *hist, int *sel, int tid)
{
    // wave 0: the bin b with prefix(b) <= k < prefix(b) + hist[b]; sel = {b, k - prefix(b)}
    if (tid < 64) {
        const int k = sel[1];
        int h[4], s = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { h[u] = hist[4 * tid + u]; s += h[u]; }
        const int incl = orbx::wave_incl_scan(s);     // inclusive scan of the lanes' sums
        int pre = incl - s;
        if (k >= pre && k < incl) {                   // exactly one lane
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k >= pre && k < pre + h[u]) { sel[0] = 4 * tid + u; sel[1] = k - pre; }
                pre += h[u];
            }
        }
    }
}
__global__ __launch_bounds__(MED_T) void k_stereo_median(const int *__restrict__ cntL, int cap, const int *__restrict__ sad,
                                                         float *__restrict__ uRight, float *__restrict__ depth, int *__restrict__ nvalid)
{
    __shared__ int hist[256];
    __shared__ int sel[2];
    __shared__ int s_nd;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = min(cntL[f], cap);
    const int *sd = sad + (size_t)f * cap;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) s_nd = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < N; i += MED_T) {
        const int v = sd[i];
        if (v >= 0) { atomicAdd(&hist[(v >> 8) & 255], 1); ++mine; }
    }
    if (mine) atomicAdd(&s_nd, mine);
    __syncthreads();
    const int nd = s_nd;
    if (tid == 0 && nvalid) nvalid[f] = nd;
    if (nd == 0) return;
    if (tid == 0) sel[1] = nd / 2;                    // vDistIdx[vDistIdx.size()/2]
    __syncthreads();
    med_pick(hist, sel, tid);
    __syncthreads();
    const int hi = sel[0];
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += MED_T) {
        const int v = sd[i];
        if (v >= 0 && ((v >> 8) & 255) == hi) atomicAdd(&hist[v & 255], 1);
    }
    __syncthreads();
    med_pick(hist, sel, tid);
    __syncthreads();
    const float median = (float)((hi << 8) | sel[0]);
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < N; i += MED_T)
        if (sd[i] >= 0 && !((float)sd[i] < thDist)) { uRight[(size_t)f * cap + i] = -1.0f; depth[(size_t)f * cap + i] = -1.0f; }
}

// one frame's count, mvuRight and mvDepth into the caller's pinned block
__global__ __launch_bounds__(256) void k_stereo_out(const int *__restrict__ count, const float *__restrict__ u, const float *__restrict__ d, int cap,
                                                    int *__restrict__ out_n, float *__restrict__ out_u, float *__restrict__ out_d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *out_n = *count;
    if (i < cap) { out_u[i] = u[i]; out_d[i] = d[i]; }
}

} // namespace

extern "C" {

int orbx_stereo_match(orbx_extractor *left, orbx_extractor *right, float mb, float mbf, void *stream_)
{
    if (!left || !right || !left->d_kps || !right->d_kps) ORBX_FAIL(ORBX_ERR_ARG, "extract on both handles first");
    if (left->last_batch != right->last_batch || left->width != right->width || left->height != right->height ||
        left->nlevels != right->nlevels || left->kcap != right->kcap || left->frame_bytes != right->frame_bytes)
        ORBX_FAIL(ORBX_ERR_ARG, "left / right extractors differ in geometry or batch");
    if (!(mb > 0.f) || !(mbf > 0.f)) ORBX_FAIL(ORBX_ERR_ARG, "bad baseline");
    ORBX_NEED_DEVICE();
    hipStream_t st = stream_ ? (hipStream_t)stream_ : left->stream;
    const int B = left->last_batch, cap = left->kcap;
    if (left->lv[0].h > ROWS_MAX) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "more than 4,096 image rows");
    // a right keypoint's band: floor(y - r) .. ceil(y + r), r = 2 * scale[octave] -> at most 2 ceil(r) + 3 rows
    const int items_cap = cap * (2 * (int)ceilf(2.0f * left->scale[left->nlevels - 1]) + 3);
    if (!left->d_st_key || left->st_batch < B) {
        if (left->d_st_key) { (void)hipFree(left->d_st_rk); (void)hipFree(left->d_st_rowoff); (void)hipFree(left->d_st_items); (void)hipFree(left->d_st_key); (void)hipFree(left->d_uright); (void)hipFree(left->d_depth); (void)hipFree(left->d_st_sad); (void)hipFree(left->d_st_scale); (void)hipFree(left->d_st_nvalid); }
        ORBX_HIP(hipMalloc(&left->d_st_key, sizeof(unsigned) * (size_t)cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_st_rk, sizeof(RightKp) * (size_t)cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_st_rowoff, sizeof(int) * (size_t)(ROWS_MAX + 1) * left->batch));   // (any image height the handle is given later)
        ORBX_HIP(hipMalloc(&left->d_st_items, sizeof(int) * (size_t)items_cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_uright, sizeof(float) * (size_t)cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_depth, sizeof(float) * (size_t)cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_st_sad, sizeof(int) * (size_t)cap * left->batch));
        ORBX_HIP(hipMalloc(&left->d_st_nvalid, sizeof(int) * (size_t)left->batch));
        ORBX_HIP(hipMalloc(&left->d_st_scale, sizeof(float) * 2 * MAXL));
        float sc[2 * MAXL];
        for (int i = 0; i < MAXL; ++i) { sc[i] = left->scale[i]; sc[MAXL + i] = left->inv_scale[i]; }
        ORBX_HIP(hipMemcpyAsync(left->d_st_scale, sc, sizeof(sc), hipMemcpyHostToDevice, st));   // on the call's stream, never the legacy one
        ORBX_HIP(hipStreamSynchronize(st));   // sc lives in this scope
        left->st_batch = left->batch;
    }
    // streams: the results of both extractors must be complete before the matching starts -- ordered on the device (an event
    // recorded behind each extraction that ran on another stream), the host does not wait
    for (orbx_extractor *e : {left, right})
        if (st != e->last_stream && hipStreamQuery(e->last_stream) != hipSuccess) {   // (an idle stream has nothing to wait for: the
            (void)hipGetLastError();                                                  // usual case after two orbx_extract calls)
            if (!e->order_ev) ORBX_HIP(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
            ORBX_HIP(hipEventRecord(e->order_ev, e->last_stream));
            ORBX_HIP(hipStreamWaitEvent(st, e->order_ev, 0));
        }
    left->st_stream = st;
    const float maxD = mbf / mb; // :557-559
    const int nRows = left->lv[0].h;
    hipLaunchKernelGGL(k_stereo_rows, dim3(B), dim3(ROWS_T), 0, st, right->d_kps, right->d_counts, cap, left->d_st_scale, nRows, items_cap,
                       (RightKp *)left->d_st_rk, left->d_st_rowoff, left->d_st_items);
    hipLaunchKernelGGL(k_stereo_hamming, dim3((cap + ST_T / 64 - 1) / (ST_T / 64), B), dim3(ST_T), 0, st, left->d_kps, left->d_desc,
                       left->d_counts, (const RightKp *)left->d_st_rk, right->d_desc, (const int *)left->d_st_rowoff,
                       (const int *)left->d_st_items, items_cap, cap, nRows, maxD, left->d_st_key);
    hipLaunchKernelGGL(k_stereo_refine, dim3((cap + 3) / 4, B), dim3(256), 0, st, left->d_kps, left->d_counts, right->d_kps,
                       cap, left->d_st_key, left->d_pyr, right->d_pyr, left->frame_bytes, left->d_lv, left->d_st_scale,
                       left->d_st_scale + MAXL, maxD, mbf, left->d_uright, left->d_depth, left->d_st_sad);
    hipLaunchKernelGGL(k_stereo_median, dim3(B), dim3(MED_T), 0, st, left->d_counts, cap, left->d_st_sad, left->d_uright,
                       left->d_depth, left->d_st_nvalid);
    ORBX_HIP(hipGetLastError());
    // ... and the next extraction on a handle whose own stream is another one must not overwrite what these kernels still read
    // (keypoints, descriptors, pyramids): it waits for this point of `st` (orbx_extract_batch, reader_pending)
    for (orbx_extractor *e : {left, right})
        if (st != e->last_stream) orbx_detail::reader_done(e, st);
    return ORBX_OK;
}

int orbx_stereo_download(orbx_extractor *left, int frame, float *uRight, float *depth, int cap, int *n)
{
    if (!left || !left->d_uright || frame < 0 || frame >= left->last_batch || !n) ORBX_FAIL(ORBX_ERR_ARG, "no stereo results");
    hipStream_t st = left->st_stream;   // queued behind the stereo match on ITS stream, one wait for that stream only
    // {count, mvuRight[kcap], mvDepth[kcap]} of the frame into a pinned block by the compute queue itself (common.h: no
    // hand-over to the copy engine), ONE synchronisation
    const size_t vec = (sizeof(float) * (size_t)left->kcap + 15) & ~(size_t)15, need = 16 + 2 * vec;
    if (need > left->st_pin_bytes) {
        if (left->h_st_pin) (void)hipHostFree(left->h_st_pin);
        left->h_st_pin = nullptr; left->st_pin_bytes = 0;
        ORBX_HIP(hipHostMalloc((void **)&left->h_st_pin, need, hipHostMallocDefault));
        left->st_pin_bytes = need;
    }
    hipLaunchKernelGGL(k_stereo_out, dim3((unsigned)((left->kcap + 255) / 256)), dim3(256), 0, st, (const int *)left->d_counts + frame,
                       (const float *)left->d_uright + (size_t)frame * left->kcap, (const float *)left->d_depth + (size_t)frame * left->kcap,
                       left->kcap, reinterpret_cast<int *>(left->h_st_pin), reinterpret_cast<float *>(left->h_st_pin + 16),
                       reinterpret_cast<float *>(left->h_st_pin + 16 + vec));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipStreamSynchronize(st));
    int cnt = 0;
    memcpy(&cnt, left->h_st_pin, sizeof(int));
    *n = cnt;
    if (cnt > cap) ORBX_FAIL(ORBX_ERR_CAPACITY, "buffer too small");
    if (cnt > 0) {
        if (uRight) memcpy(uRight, left->h_st_pin + 16, sizeof(float) * cnt);
        if (depth) memcpy(depth, left->h_st_pin + 16 + vec, sizeof(float) * cnt);
    }
    return ORBX_OK;
}

int orbx_stereo_download_batch(orbx_extractor *left, float *uRight, float *depth, int32_t *counts)
{
    if (!left || !left->d_uright || left->last_batch <= 0) ORBX_FAIL(ORBX_ERR_ARG, "no stereo results");
    hipStream_t st = left->st_stream;
    const size_t n = (size_t)left->last_batch * left->kcap;
    if (counts) ORBX_HIP(hipMemcpyAsync(counts, left->d_counts, sizeof(int) * left->last_batch, hipMemcpyDeviceToHost, st));
    if (uRight) ORBX_HIP(hipMemcpyAsync(uRight, left->d_uright, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    if (depth) ORBX_HIP(hipMemcpyAsync(depth, left->d_depth, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}

} // extern "C"
