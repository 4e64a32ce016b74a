// orbx_extract.hip -- ORB extractor (FAST pyramid + octree + rBRIEF) for gfx950.
//
// Replaces ORB_SLAM2::ORBextractor (src/ORBextractor.cc, include/ORBextractor.h).
// One batch = B independent frames; every stage is one launch over all frames
// (and, where there is no dependency, all pyramid levels):
//
//   k_pyr_level0    copyMakeBorder of the input            (:1135)
//   k_pyr_resize    resize level l from l-1 + border       (:1128-1131)   x (nlevels-1)
//   k_fast_cells    per 30-px cell FAST-9/16 score + NMS + threshold fallback (:789-837)
//   k_octree        DistributeOctTree as scan-based rounds (:539-763)
//   k_blur          GaussianBlur 7x7 sigma 2               (:1093-1094)
//   k_describe      IC_Angle + steered BRIEF + output      (:77-147, :845-860, :1103-1111)
//
// HBM layout: per frame one pyramid block; level l is a padded image of
// (h_l + 38) rows x stride_l bytes, inner pixel (x, y) at off_l + (y+19)*stride_l + 32 + x
// (19-px REFLECT_101 border as in mvImagePyramid; 32-byte left pad keeps inner
// rows dword/16-B aligned).  All integer work is bit-exact by construction; the
// float steps (fastAtan2, rBRIEF rotation) use fixed IEEE op sequences (orbx_math.h,
// built with -ffp-contract=off).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "orbx_internal.h"
#include "orbx_math.h"

using namespace orbx_detail;

namespace {

__constant__ signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// ------------------------------------------------------------------ pyramid
// Both pyramid kernels are latency-bound (two dependent global loads per output),
// so every thread produces 4 pixels (one dword store) in PYR_ROWS rows and keeps
// all of their loads in flight at once.  The 19-px REFLECT_101 border is
// recomputed through the reflected coordinate instead of copied afterwards.
// Rows per thread, round 4 (same-box A/B of the pipelined step, three rounds each): 4 -> 0.2495 ms, 6 -> 0.2474, 8 -> 0.2447-0.2471,
// 10 -> 0.2536, 16 -> 0.2568.  A resize launch alone is no faster with 8 (8.7 against 8.1 us: half the waves, each twice as long) -- but
// the chain then holds half the wave slots while it waits for memory, and the other contexts' kernels get them.
constexpr int PYR_ROWS = 8;

#ifdef ORBX_PHASE_TIMING
// development aid (never in the product build): shader-clock time per kernel phase, one record per workgroup (plain
// stores: atomics on shared counters serialise in one L2 channel and the measurement measures itself)
__device__ unsigned long long g_phase_rec[4 * 65536 * 16];   // per workgroup: 8 phase times (shader clock), [13] = end and [15] = start in the 100 MHz wall clock, [14] = HW_ID
#define ORBX_PH_INIT(K) unsigned long long *const ph_rec = g_phase_rec + ((K) * 65536 + ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 65535u)) * 16; \
    if (threadIdx.x == 0) { ph_rec[15] = wall_clock64(); ph_rec[14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } \
    __builtin_amdgcn_s_waitcnt(0); unsigned long long ph_t = clock64()   /* the wall clock's own latency stays outside the first phase */
#define ORBX_PH(i, cond) do { if (cond) { const unsigned long long ph_n = clock64(); ph_rec[(i) & 7] = ph_n - ph_t; ph_t = ph_n; } } while (0)
#define ORBX_PH_END(cond) do { if (cond) ph_rec[13] = wall_clock64(); } while (0)   // s_memrealtime is slow: once, at the very end
#define ORBX_PHA(i, cond) do { if (cond) { const unsigned long long ph_n = clock64(); ph_rec[(i) & 7] += ph_n - ph_t; ph_t = ph_n; } } while (0)   // accumulating (loops)
#else
#define ORBX_PH_INIT(K) do {} while (0)
#define ORBX_PH(i, cond) do {} while (0)
#define ORBX_PH_END(cond) do {} while (0)
#define ORBX_PHA(i, cond) do {} while (0)
#endif

// XCD-aware (frame, item) mapping for grids of (nitems, nframes) workgroups (speed only,
// never correctness): workgroups are dealt round-robin over the 8 XCDs in dispatch
// order (x fastest), each XCD has its own L2; handing every XCD whole frames lets the
// halos / patches of one frame hit in one L2 instead of being fetched by all eight.
__device__ __forceinline__ void xcd_frame_item(int &frame, int &item)
{
    const int nb = gridDim.x, B = gridDim.y, id = blockIdx.x + nb * blockIdx.y;
    if ((B & 7) == 0) {
        const int x = id & 7, s = id >> 3;
        frame = (s / nb) * 8 + x;
        item = s % nb;
    } else {
        frame = blockIdx.y;
        item = blockIdx.x;
    }
}

// Level 0: copyMakeBorder(image, temp, 19,19,19,19, BORDER_REFLECT_101), ORBextractor.cc:1135.
__global__ __launch_bounds__(64) void k_pyr_level0(const uint8_t *__restrict__ img, int img_stride,
                                                   size_t img_frame_stride, uint8_t *__restrict__ pyr,
                                                   size_t frame_bytes, LevelInfo lv, int aligned)
{
    const int xw = blockIdx.x * 64 + threadIdx.x;
    if (xw * 4 >= lv.stride) return;
    const int f = blockIdx.z, px0 = xw * 4 - PADX;
    const bool inner = aligned && px0 >= 0 && px0 + 4 <= lv.w;
    int sx[4];
    uint32_t keep = 0; // bytes inside [-EDGE, w+EDGE); the others (row padding) read a clamped column and are masked to 0
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int px = px0 + b;
        sx[b] = reflect101(min(max(px, -EDGE), lv.w + EDGE - 1), lv.w);
        keep |= (px >= -EDGE && px < lv.w + EDGE) ? 0xffu << (8 * b) : 0u;
    }
    uint32_t out[PYR_ROWS];
#pragma unroll
    for (int r = 0; r < PYR_ROWS; ++r) {
        const int row = blockIdx.y * PYR_ROWS + r;
        out[r] = 0;
        if (row < lv.h + 2 * EDGE) {
            const uint8_t *src = img + (size_t)f * img_frame_stride + (size_t)reflect101(row - EDGE, lv.h) * img_stride;
            if (inner) {
                out[r] = *reinterpret_cast<const uint32_t *>(src + px0);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) out[r] |= (uint32_t)src[sx[b]] << (8 * b);
                out[r] &= keep;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PYR_ROWS; ++r) {
        const int row = blockIdx.y * PYR_ROWS + r;
        if (row < lv.h + 2 * EDGE)
            *reinterpret_cast<uint32_t *>(pyr + (size_t)f * frame_bytes + lv.off + (size_t)row * lv.stride + xw * 4) = out[r];
    }
}

// The same for images whose width is a multiple of 8 (and whose rows are dword-aligned), without a divergent wave: the
// padded level is (h + 38) rows x pp = w / 8 interior pieces of 8 bytes -- every one a straight copy from a (reflected)
// source row, numbered row-major and dealt out 64 x 4 per wave (8 bytes per lane and load: the texture addresser's sweet
// spot) -- plus nb border dwords per row (the 19 px either side), which are put together byte by byte through the reflected
// column by waves of their own.  The dword form above had the border lanes in the first and last wave of every row group:
// both paths executed by two waves of three.  inv_pp / inv_nb = ceil(2^32 / pp), ceil(2^32 / nb): exact quotients.
__global__ __launch_bounds__(64) void k_pyr_level0_lin(const uint8_t *__restrict__ img, int img_stride, size_t img_frame_stride,
                                                       uint8_t *__restrict__ pyr, size_t frame_bytes, LevelInfo lv, int pp,
                                                       unsigned inv_pp, int nblk_int, int nb, unsigned inv_nb, int nleft)
{
    int f, bx;
    xcd_frame_item(f, bx);          // grid (workgroups per frame, frames): a frame's pyramid is made by one XCD, launch after launch
    const int lane = threadIdx.x, rows = lv.h + 2 * EDGE;
    const uint8_t *src = img + (size_t)f * img_frame_stride;
    uint8_t *dst = pyr + (size_t)f * frame_bytes + lv.off;
    if (bx < nblk_int) {
        uint2 v[4];
        int row[4], pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = min(bx * 256 + u * 64 + lane, rows * pp - 1);   // past the end: the last piece again
            row[u] = (int)(((unsigned long long)(unsigned)p * inv_pp) >> 32);
            pc[u] = p - row[u] * pp;
            __builtin_memcpy(&v[u], src + (uint32_t)(reflect101(row[u] - EDGE, lv.h) * img_stride + 8 * pc[u]), 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            *reinterpret_cast<uint2 *>(dst + (uint32_t)(row[u] * lv.stride + PADX + 8 * pc[u])) = v[u];
    } else {
        const int q = (bx - nblk_int) * 64 + lane;
        if (q >= rows * nb) return;
        const int row = (int)(((unsigned long long)(unsigned)q * inv_nb) >> 32), k = q - row * nb;
        const int xw = k < nleft ? ((PADX - EDGE) >> 2) + k : ((PADX + lv.w) >> 2) + (k - nleft);   // dword of the padded row
        const uint8_t *s = src + (uint32_t)(reflect101(row - EDGE, lv.h) * img_stride);
        uint32_t out = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int px = xw * 4 - PADX + b;
            const uint32_t val = s[reflect101(min(max(px, -EDGE), lv.w + EDGE - 1), lv.w)];
            out |= (px >= -EDGE && px < lv.w + EDGE) ? val << (8 * b) : 0u;   // row padding stays 0
        }
        *reinterpret_cast<uint32_t *>(dst + (uint32_t)(row * lv.stride + xw * 4)) = out;
    }
}

// Level l from level l-1: cv::resize INTER_LINEAR 8U fixed point
// + REFLECT_101 border.  xt[dx] = {sx, a0 | a1<<16}, yt[dy] = {sy0, sy1, b0, b1}:
// OpenCV's coefficient tables, built on the host.
//
// Interior dword groups (4 pixels with 0 <= x < w, no reflection): the source columns of such a group are
// non-decreasing and span at most 8 bytes (the host checks it per level), so each source row is ONE unaligned
// 8-byte load and a pixel's pair (S[sx], S[sx+1]) is one v_perm_b32 into two u16 halves, times (a0, a1) one
// v_dot2_u32_u16.  The groups that touch the border take their pixels one by one through the reflected
// coordinate (nxf = 0: every group does).
// (b (r >> 4)) >> 16 = floor((b 2^12) (r & ~15) / 2^32): both factors are below 2^24 (b <= 2048, r <= 255 * 2048), so
// it is one v_mul_hi_u32_u24 after the mask instead of shift, multiply, shift.  bs = b << 12.
// (written as asm: from the masked 64-bit product the compiler makes v_mul_hi_u32, a quarter-rate instruction -- 64 of them were
// a fifth of the kernel's VALU time)
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b)   // (a[23:0] * b[23:0]) >> 32
{
    uint32_t d;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a[23:0] * b[23:0], low 32 bits, as ONE full-rate instruction: the compiler folds __mul24 of values whose range it cannot see back into
// v_mul_lo_u32 (quarter rate)
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t resize_vertical(int r0, int r1, uint32_t bs0, uint32_t bs1)
{
    const uint32_t t0 = mulhi_u24(bs0, (uint32_t)r0 & 0xfffff0u);
    const uint32_t t1 = mulhi_u24(bs1, (uint32_t)r1 & 0xfffff0u);
    const uint32_t v = (t0 + t1 + 2) >> 2;
    return v > 255u ? 255u : v;
}
typedef unsigned short pyr_u16x2 __attribute__((ext_vector_type(2)));
// Thread -> work.  A row of level 1 has 133 interior groups + 11 border groups: one wave per 64 groups of a row left the
// third and the border wave of every row 80-90 % empty (45-55 % of all lanes idle).  Now: workgroups 0 .. nff*nrg-1 are full
// waves of 64 consecutive interior groups of ONE row group (contiguous 256-byte rows in and out: tiling these 16 x 4
// instead was slower although it saved instructions -- the kernel lives on the memory pipe); the REST of every row
// (the last ngi % 64 interior groups and the border groups, `ntail` in all) is gathered tw groups x 64/tw row groups per
// wave and goes through the coordinate-by-coordinate path, which serves interior groups as well.  Dwords of pure row
// padding (outside the 19-px border) are not written: the workspace is zeroed once.
__global__ __launch_bounds__(64) void k_pyr_resize(uint8_t *__restrict__ pyr, size_t frame_bytes, LevelInfo src,
                                                   LevelInfo dst, const int2 *__restrict__ xt,
                                                   const int4 *__restrict__ yt, int nff, int nint, int nrg, int ntail, int tw_shift, int ntw, int rpw, int tail_window)
{
    int f, bx;
    xcd_frame_item(f, bx);
    const int tid = threadIdx.x;
    ORBX_PH_INIT(2);
#ifdef ORBX_PHASE_TIMING
    if (tid == 0) { ph_rec[12] = dst.w; ph_rec[11] = (bx < nff * ((nrg + rpw - 1) / rpw)) ? 1 : 0; }
#endif
    const int xlo = (PADX - EDGE) >> 2, xhi = (PADX + dst.w + EDGE - 1) >> 2; // first / last dword that holds border or image bytes
    const int nrgw = (nrg + rpw - 1) / rpw;   // a full wave takes rpw consecutive row groups, one after the other
    const bool fast = bx < nff * nrgw;
    int xw, rowg;
    if (fast) {
        const int rw = bx / nff;
        const int g = (bx - rw * nff) * 64 + tid;
        if (g >= nint) return;   // the last wave of a row may be partly filled (nint = interior groups the full-wave path takes)
        xw = PADX / 4 + g;
        rowg = rw * rpw;
    } else {
        // tail tile: 2^tw_shift groups x 64 >> tw_shift row groups; ntw > 1 (a tail wider than a wave) only with tw_shift = 6
        const int tb = bx - nff * nrgw, tr = tb / ntw;
        const int col = ((tb - tr * ntw) << tw_shift) + (tid & ((1 << tw_shift) - 1));
        rowg = (tr << (6 - tw_shift)) + (tid >> tw_shift);
        if (col >= ntail) return;
        const int nleft = PADX / 4 - xlo;                                  // border groups on the left
        xw = col < nleft ? xlo + col : PADX / 4 + nint + (col - nleft);      // left border, leftover interior, right border
    }
    if (xw > xhi || rowg * PYR_ROWS >= dst.h + 2 * EDGE) return;
    const uint8_t *base = pyr + (size_t)f * frame_bytes + src.off + PADX;
    uint32_t out[PYR_ROWS];
    if (fast) {
        // A full wave works on one row group at a time: the four row-table entries are wave-uniform and come through the
        // scalar cache; the column table is read once and serves all rpw row groups of the wave.  The kernel's time is a
        // wave's life (three dependent memory round trips, 4-6 us) times the number of wave generations, and a level's
        // wave count is what the host sets through rpw so that one generation holds them all.
        const int px0 = xw * 4 - PADX;
        const int4 xa = *reinterpret_cast<const int4 *>(xt + px0), xb = *reinterpret_cast<const int4 *>(xt + px0 + 2);
        const int sx0 = xa.x;
        const uint32_t coef[4] = {(uint32_t)xa.y, (uint32_t)xa.w, (uint32_t)xb.y, (uint32_t)xb.w};
        // selector {byte o, 0, byte o+1, 0} of the 8-byte window, o = sx - sx0
        const uint32_t sel[4] = {0x0c010c00u, 0x0c010c00u + (uint32_t)(xa.z - sx0) * 0x00010001u,
                                 0x0c010c00u + (uint32_t)(xb.x - sx0) * 0x00010001u, 0x0c010c00u + (uint32_t)(xb.z - sx0) * 0x00010001u};
        const int rg0 = __builtin_amdgcn_readfirstlane(rowg);
        for (int j = 0; j < rpw; ++j) {
            const int rg = rg0 + j;
            if (rg * PYR_ROWS >= dst.h + 2 * EDGE) break;
            int4 yy[PYR_ROWS];
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                const int row = rg * PYR_ROWS + r;
                yy[r] = yt[reflect101((row < dst.h + 2 * EDGE ? row : 0) - EDGE, dst.h)];
            }
            uint2 w0[PYR_ROWS], w1[PYR_ROWS];
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                __builtin_memcpy(&w0[r], base + (uint32_t)((yy[r].x + EDGE) * src.stride + sx0), 8);
                __builtin_memcpy(&w1[r], base + (uint32_t)((yy[r].y + EDGE) * src.stride + sx0), 8);
            }
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                out[r] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(w0[r].y, w0[r].x, sel[b])),
                                                               __builtin_bit_cast(pyr_u16x2, coef[b]), 0u, false);
                    const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(w1[r].y, w1[r].x, sel[b])),
                                                               __builtin_bit_cast(pyr_u16x2, coef[b]), 0u, false);
                    out[r] |= resize_vertical(r0, r1, (uint32_t)yy[r].z << 12, (uint32_t)yy[r].w << 12) << (8 * b);
                }
            }
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                const int row = rg * PYR_ROWS + r;
                if (row < dst.h + 2 * EDGE)
                    *reinterpret_cast<uint32_t *>(pyr + (size_t)f * frame_bytes + dst.off + (uint32_t)(row * dst.stride + xw * 4)) = out[r];
            }
        }
        ORBX_PH(1, tid == 0);
        ORBX_PH_END(tid == 0);
        return;
    }
    {
        int4 yy[PYR_ROWS];
#pragma unroll
        for (int r = 0; r < PYR_ROWS; ++r) {
            const int row = rowg * PYR_ROWS + r;
            yy[r] = yt[reflect101((row < dst.h + 2 * EDGE ? row : 0) - EDGE, dst.h)];
        }
        // no per-pixel branches: pixels outside [-EDGE, w+EDGE) (row padding) take the column of a clamped coordinate and
        // are masked out of the stored dword
        const int px0 = xw * 4 - PADX;
        int sxs[4], a0[4], a1[4];
        uint32_t keep = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int px = px0 + b;
            const bool in = px >= -EDGE && px < dst.w + EDGE;
            const int2 xx = xt[reflect101(min(max(px, -EDGE), dst.w + EDGE - 1), dst.w)];
            sxs[b] = xx.x; a0[b] = xx.y & 0xffff; a1[b] = xx.y >> 16;
            keep |= in ? 0xffu << (8 * b) : 0u;
        }
        if (tail_window) {
            // The four pixels of a border group are reflections of interior pixels, so their source columns lie as close together as
            // an interior group's (in mirrored order): one 8-byte window from the smallest of them serves all four, as in the
            // full waves (the host checks the span per level) -- 8 loads per lane instead of 64 single bytes, which made a tail
            // wave six times as expensive for the texture addresser as a full one.
            const int smin = min(min(sxs[0], sxs[1]), min(sxs[2], sxs[3]));
            uint32_t sel[4], coef[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                sel[b] = 0x0c010c00u + (uint32_t)(sxs[b] - smin) * 0x00010001u;   // {byte o, 0, byte o+1, 0}
                coef[b] = (uint32_t)a0[b] | ((uint32_t)a1[b] << 16);
            }
            uint2 w0[PYR_ROWS], w1[PYR_ROWS];
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                __builtin_memcpy(&w0[r], base + (uint32_t)((yy[r].x + EDGE) * src.stride + smin), 8);
                __builtin_memcpy(&w1[r], base + (uint32_t)((yy[r].y + EDGE) * src.stride + smin), 8);
            }
#pragma unroll
            for (int r = 0; r < PYR_ROWS; ++r) {
                out[r] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(w0[r].y, w0[r].x, sel[b])),
                                                               __builtin_bit_cast(pyr_u16x2, coef[b]), 0u, false);
                    const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(w1[r].y, w1[r].x, sel[b])),
                                                               __builtin_bit_cast(pyr_u16x2, coef[b]), 0u, false);
                    out[r] |= resize_vertical(r0, r1, (uint32_t)yy[r].z << 12, (uint32_t)yy[r].w << 12) << (8 * b);
                }
                out[r] &= keep;
            }
        } else
#pragma unroll
        for (int r = 0; r < PYR_ROWS; ++r) {
            const uint8_t *S0 = base + (size_t)(yy[r].x + EDGE) * src.stride;
            const uint8_t *S1 = base + (size_t)(yy[r].y + EDGE) * src.stride;
            out[r] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r0 = __mul24(S0[sxs[b]], a0[b]) + __mul24(S0[sxs[b] + 1], a1[b]);
                const int r1 = __mul24(S1[sxs[b]], a0[b]) + __mul24(S1[sxs[b] + 1], a1[b]);
                out[r] |= resize_vertical(r0, r1, (uint32_t)yy[r].z << 12, (uint32_t)yy[r].w << 12) << (8 * b);
            }
            out[r] &= keep;
        }
    }
    ORBX_PH(1, tid == 0 && out[0] != 0x12345678u);   // column table, source rows, arithmetic
#pragma unroll
    for (int r = 0; r < PYR_ROWS; ++r) {
        const int row = rowg * PYR_ROWS + r;
        if (row < dst.h + 2 * EDGE)
            *reinterpret_cast<uint32_t *>(pyr + (size_t)f * frame_bytes + dst.off + (uint32_t)(row * dst.stride + xw * 4)) = out[r];
    }
    ORBX_PH(2, tid == 0);
    ORBX_PH_END(tid == 0);
}

// Dword load at any byte address: gfx950 under amdhsa runs in unaligned-access mode (the compiler itself turns an
// align-1 4-byte copy into one global_load_dword).
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// --------------------------------------------------------------------- FAST
// Circle pixels p_k, k = 0..15: the 16-pixel Bresenham circle of radius 3 in OpenCV's makeOffsets order, starting at
// (0, 3) and running through (3, 0), (0, -3), (-3, 0).
// Necessary condition for a 9-arc (OpenCV's opposite-pair pre-test): a 9-arc contains
// one pixel of every opposite pair (k, k+8), so with d_k = v - p_k a dark arc needs
// min_k max(d_k, d_k+8) > th and a bright one max_k min(d_k, d_k+8) < -th (both at
// once cannot be a corner: a 9-arc holds both pixels of one pair).
// Evaluated for 4 horizontally adjacent pixels per lane, on packed u16 pairs
// (v_perm_b32 / v_pk_min_u16 / v_pk_max_u16; there is no byte-wise min/max).  With
// d_k = v - p_k:  min_k max(d_k, d_k+8) > th  <=>  v > max_k min(p_k, p_k+8) + th  (dark)
//                max_k min(d_k, d_k+8) < -th <=>  min_k max(p_k, p_k+8) > v + th  (bright)
// so the differences are never formed.  p = dword-aligned address of the 4 centre pixels
// in the LDS tile; the 7 rows x 12 bytes around it are read as dwords and the circle
// pixels of lanes-pixels (0,2) / (1,3) are pulled out as (even, odd) u16 pairs.
// Returns one byte per pixel: 1 = dark arc possible, 2 = bright, 0 = neither.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t pk1(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
template <int DX>
__device__ __forceinline__ void fast_pick(const uint32_t r[3], u16x2 &E, u16x2 &O)
{
    constexpr int s = 4 + DX;                  // byte of pixel 0 in the 12-byte window r[0] | r[1] | r[2]
    constexpr int base = (DX < 0) ? s : s - 4; // ... inside the 8-byte pair handed to v_perm
    const uint32_t hi = (DX < 0) ? r[1] : r[2], lo = (DX < 0) ? r[0] : r[1];
    constexpr uint32_t selE = base | 0x0c00u | ((base + 2) << 16) | 0x0c000000u;       // {b, 0, b+2, 0}
    constexpr uint32_t selO = (base + 1) | 0x0c00u | ((base + 3) << 16) | 0x0c000000u; // {b+1, 0, b+3, 0}
    E = pk2(__builtin_amdgcn_perm(hi, lo, selE));
    O = pk2(__builtin_amdgcn_perm(hi, lo, selO));
}
// Both u16 halves: a > b ? 1 : 0, as v_pk_sub_u16 with clamp + v_pk_min_u16.  Written in asm because the compiler
// rewrites min(sub_sat(a, b), 1) into two 16-bit compares, two selects and a v_perm (six instructions instead of two);
// the operands come from VALU instructions, so there is no software-visible hazard to pad.
__device__ __forceinline__ uint32_t pk_gt01(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(0x00010001u));
    return d;
}
// number of set bits of a wave mask below this lane
__device__ __forceinline__ int mask_rank(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// PAIRS: bit k = opposite pair (k, k+8) takes part.  Returns bytes 0..3 = pixels 0..3: dark possible | bright possible << 1
// over the chosen pairs.  With all eight pairs both bits at once cannot be a corner (see above); with a subset either
// polarity is still open, so the codes of two subsets are ANDed first and the rule applied to the result
// (fast_resolve).  Rows whose pairs are not chosen are never loaded (the loads fold away at compile time).
template <int TS, unsigned PAIRS>
__device__ __forceinline__ uint32_t fast_pretest4(const uint8_t *p, uint32_t th2)
{
    uint32_t r[7][3];
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p + dy * TS);
        r[dy + 3][0] = q[-1]; r[dy + 3][1] = q[0]; r[dy + 3][2] = q[1];
    }
    u16x2 aE, aO, bE, bO, loE, loO, hiE, hiO;
    constexpr int FIRSTK = __builtin_ctz(PAIRS);
#define ORBX_PAIR(K, DYA, DXA)                                                                       \
    if constexpr ((PAIRS >> (K)) & 1u) {                                                             \
        fast_pick<DXA>(r[3 + (DYA)], aE, aO);                                                        \
        fast_pick<-(DXA)>(r[3 - (DYA)], bE, bO);                                                     \
        if constexpr ((K) == FIRSTK) {                                                               \
            loE = __builtin_elementwise_min(aE, bE); loO = __builtin_elementwise_min(aO, bO);        \
            hiE = __builtin_elementwise_max(aE, bE); hiO = __builtin_elementwise_max(aO, bO);        \
        } else {                                                                                     \
            loE = __builtin_elementwise_max(loE, __builtin_elementwise_min(aE, bE));                 \
            loO = __builtin_elementwise_max(loO, __builtin_elementwise_min(aO, bO));                 \
            hiE = __builtin_elementwise_min(hiE, __builtin_elementwise_max(aE, bE));                 \
            hiO = __builtin_elementwise_min(hiO, __builtin_elementwise_max(aO, bO));                 \
        }                                                                                            \
    }
    ORBX_PAIR(0, 3, 0)   // circle pixels 0 / 8
    ORBX_PAIR(1, 3, 1)   // 1 / 9
    ORBX_PAIR(2, 2, 2)   // 2 / 10
    ORBX_PAIR(3, 1, 3)   // 3 / 11
    ORBX_PAIR(4, 0, 3)   // 4 / 12
    ORBX_PAIR(5, -1, 3)  // 5 / 13
    ORBX_PAIR(6, -2, 2)  // 6 / 14
    ORBX_PAIR(7, -3, 1)  // 7 / 15
#undef ORBX_PAIR
    u16x2 vE, vO;
    fast_pick<0>(r[3], vE, vO);
    const u16x2 t = pk2(th2);
    // per half "a > b" as 0 / 1: saturating difference clamped to 1
    const uint32_t dE = pk_gt01(pk1(vE), pk1(loE + t)), dO = pk_gt01(pk1(vO), pk1(loO + t));
    const uint32_t gE = pk_gt01(pk1(hiE), pk1(vE + t)), gO = pk_gt01(pk1(hiO), pk1(vO + t));
    return (dE | (gE << 1)) | ((dO | (gO << 1)) << 8);
}
// dark and bright both possible over all eight pairs cannot be a corner -> 0
__device__ __forceinline__ uint32_t fast_resolve(uint32_t code) { return code & ~((code & (code >> 1) & 0x01010101u) * 3u); }
constexpr unsigned FAST_PAIRS_A = 0x11u;   // compass pairs 0/8 and 4/12 on every pixel; the other six on what is left, pixel by pixel

// cornerScore<16>: (largest arc-minimum of e_k over the 16 circular 9-arcs) - 1
// where e = d (dark) or -d (bright), d_k = v - p_k; it is a corner at threshold th iff that
// arc-minimum exceeds th, and the score does not depend on th.
// Computed on packed halves.  X[k] (k = 0..7) holds e_k in its low and e_(k+8) in its high 16 bits, every value
// biased to 0x4100 + e: the bit patterns 0x4001..0x41ff are positive normal f16 numbers, and for those the float order IS
// the integer order -- so gfx950's three-operand packed v_pk_minimum3_f16 / v_pk_maximum3_f16 serve as integer min3 /
// max3 on two values at once (no NaN, no zero, no denormal among the operands; nothing is rounded: min / max only
// select).  X[k+8] is X[k] with its halves swapped, which the op_sel modifiers do for free.  min over the 9-arc
// starting at k = min3 of three 3-arcs; 16 + 4 instructions instead of 64.  Returns max_k min(e_k .. e_(k+8)) + 0x4100.
template <int S1, int S2> __device__ __forceinline__ uint32_t pk_min3_h(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    if constexpr (S1 && S2) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if constexpr (S2) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3_h(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t fast_arc_best_packed(const uint32_t X[8])
{
    uint32_t m3[8], m9[8];
    m3[0] = pk_min3_h<0, 0>(X[0], X[1], X[2]); m3[1] = pk_min3_h<0, 0>(X[1], X[2], X[3]);
    m3[2] = pk_min3_h<0, 0>(X[2], X[3], X[4]); m3[3] = pk_min3_h<0, 0>(X[3], X[4], X[5]);
    m3[4] = pk_min3_h<0, 0>(X[4], X[5], X[6]); m3[5] = pk_min3_h<0, 0>(X[5], X[6], X[7]);
    m3[6] = pk_min3_h<0, 1>(X[6], X[7], X[0]); m3[7] = pk_min3_h<1, 1>(X[7], X[0], X[1]);
    m9[0] = pk_min3_h<0, 0>(m3[0], m3[3], m3[6]); m9[1] = pk_min3_h<0, 0>(m3[1], m3[4], m3[7]);
    m9[2] = pk_min3_h<0, 1>(m3[2], m3[5], m3[0]); m9[3] = pk_min3_h<0, 1>(m3[3], m3[6], m3[1]);
    m9[4] = pk_min3_h<0, 1>(m3[4], m3[7], m3[2]); m9[5] = pk_min3_h<1, 1>(m3[5], m3[0], m3[3]);
    m9[6] = pk_min3_h<1, 1>(m3[6], m3[1], m3[4]); m9[7] = pk_min3_h<1, 1>(m3[7], m3[2], m3[5]);
    const uint32_t a = pk_max3_h(m9[0], m9[1], m9[2]), b = pk_max3_h(m9[3], m9[4], m9[5]);
    const uint32_t c = pk_max3_h(m9[6], m9[7], a), m = pk_max3_h(b, c, c);
    return max(m & 0xffffu, m >> 16);
}

// One wave per 30-px cell (ORBextractor.cc:789-837): cv::FAST(cell, iniThFAST, nms=true),
// rerun with minThFAST only if the cell came back empty; candidates are written
// in FAST raster order, coordinates relative to (minBorderX, minBorderY).
// Packed candidate: y<<20 | x<<8 | score.
//
// Phases: (1) cell tile -> LDS as dwords; then, as the reference does, one pass at iniThFAST and -- only if that
// pass emits nothing (a wave-uniform branch; on textured frames almost every cell has a corner at iniThFAST) -- a
// second one at minThFAST:
// (2a) every pixel, 4 adjacent pixels per lane on packed u16: the opposite-pair pre-test on the two compass
// pairs only (0/8 and 4/12: the vertical pair needs no byte shuffling across dwords); the 4-pixel groups that still
// hold a possible corner (18 % of them at level 0, 67 % at level 7 on the synthetic frames) are compacted in raster
// order into an LDS group queue; (2b) the queued groups' live pixels go to the pixel queue, and the other six pairs are tested
// with a lane per pixel (a group-wise test on packed halves worked on all four pixels of a group of which one or two were
// alive), survivors (4 % ... 28 % of the pixels) compacted in place; (3) survivors only: arc score -> score map (0 below
// the pass's threshold, which is all the non-max test needs: a neighbour that is no corner at this threshold scores
// less than any corner); (4) 3x3 strict NMS and ordered emission, one loop.
// TS / SS (tile and score-map strides) are compile-time so that the 16 circle
// offsets fold into ds_read immediates.
template <int TS, int SS>
__global__ __launch_bounds__(64) void k_fast_cells(const uint8_t *__restrict__ pyr, size_t frame_bytes,
                                                   const LevelInfo *__restrict__ L,
                                                   const CellInfo *__restrict__ cells, int *__restrict__ cell_count,
                                                   int cells_per_frame, uint32_t *__restrict__ cands,
                                                   size_t cands_per_frame, int iniTh, int minTh,
                                                   int tile_bytes, int sc_bytes, int queue_bytes)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint8_t *tile = smem;
    // LDS per wave decides how many waves a CU holds (the kernel speeds up with occupancy well beyond 20 waves per CU, and what
    // it leaves free is what the other pipeline contexts' kernels can use): the group queue is dead once stage (2b) has read
    // it and the score map is born after that, so they share one region (sc_bytes = the larger of the two)
    uint8_t *sc = smem + tile_bytes;
    uint32_t *gqueue = reinterpret_cast<uint32_t *>(smem + tile_bytes);
    unsigned short *queue = reinterpret_cast<unsigned short *>(smem + tile_bytes + sc_bytes);
    (void)queue_bytes;
    int c, f;
    xcd_frame_item(f, c);
    c = cells_per_frame - 1 - c;   // dispatch order: the small levels' cells (most survivors, longest waves) first, level 0 last -- a shorter tail
    const int lane = threadIdx.x;
    ORBX_PH_INIT(0);
    const FastCell ci = cells[c];   // everything the wave needs about its cell: no second dependent table read
    const int cw = ci.cw, ch = ci.ch, zw = cw - 6, zh = ch - 6;
    int total = 0;
    if (zw > 0 && zh > 0) {
        // (1) tile: pixel (x, y) of the cell lives at tile[y*TS + 5 + x], so that zone pixel 0 (cell pixel 3) sits
        // on the dword boundary at column 8 whatever the cell's position: the pre-test's 4-pixel groups then cover a
        // zone row with ceil(zw/4) groups.  Dword loads at byte addresses (see load_u32_unaligned).
        // LW lanes side by side on a tile row (ndw <= LW by the choice of TS), 64 / LW rows per step: a lane's column never
        // changes.  All loads of a chunk of NB steps are issued before the first is stored, and none of them sits behind a
        // branch (a lane without work in a step reads the cell's last row / last dword again): the wave spends ONE memory
        // latency here, not one per step -- this phase was 44 % of a wave's life when every load was followed by its
        // store.  44 rows = one chunk covers every cell of a 640 x 480 frame.  (Several cells per wave with the next
        // cell's tile in flight during the current one's work: kernel 3-20 % slower, the tail grows; dropped.)
        constexpr int LW = TS > 64 ? 32 : 16, RPI = 64 / LW, NB = TS > 64 ? 16 : 11;
        const int ndw = (cw + 5 + 3) >> 2;
        const uint8_t *img = pyr + (size_t)f * frame_bytes + ci.img_off;
        {
            const int xw = lane & (LW - 1), yr = lane / LW;
            const uint32_t xoff = 4u * (uint32_t)min(xw, ndw - 1);
            uint8_t *dst = tile + yr * TS + 4 * xw;
            const int ylim = xw < ndw ? ch - yr : 0;      // rows y0 = 0, RPI, 2 RPI, ... < ylim are this lane's
            for (int yb = 0; yb < ch; yb += NB * RPI) {
                uint32_t v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    v[i] = load_u32_unaligned(img + ((uint32_t)__mul24(min(yb + yr + i * RPI, ch - 1), ci.stride) + xoff));
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (yb + i * RPI < ylim) *reinterpret_cast<uint32_t *>(dst + i * RPI * TS) = v[i];
                dst += NB * RPI * TS;
            }
        }
        __syncthreads();
        ORBX_PH(0, lane == 0);   // tile load
        constexpr int zc0 = 8;                        // tile column of zone pixel 0
        const uint8_t *t0 = tile + 3 * TS + zc0;      // zone pixel (0,0)
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int g0 = zc0 >> 2, ngx = ci.ngx, ngrp = ngx * zh;
        const int qstep = ci.qstep, rstep = ci.rstep;         // lane + 64 -> (gx + rstep, y + qstep), one carry
        unsigned short *const dump = queue + zw * zh;         // one spare slot per lane for rejected pixels
        uint32_t *out = cands + (size_t)f * cands_per_frame + ci.cand_off;
        for (int pass = 0; pass < 2; ++pass) {
            const int th = pass ? minTh : iniTh;
            const uint32_t th2 = (uint32_t)th | ((uint32_t)th << 16);
            // (2a) compass pairs on every group; group queue entry = code | y << 2 | gx << 10 (the code uses bits 0-1 of
            // every byte).  A group = the 4 pixels of one tile dword; groups overlapping the zone row, in raster order.
            int ngq = 0;
            {
                int y = (lane * ci.inv_ngx) >> 16, gx = lane - y * ngx;
                for (int p0 = 0; p0 < ngrp; p0 += 64) {
                    uint32_t code = 0;
                    // a row's last group may reach past the zone: those pixels are masked in (2b), on the queued groups only
                    if (p0 + lane < ngrp) code = fast_pretest4<TS, FAST_PAIRS_A>(tile + __mul24(y + 3, TS) + 4 * (g0 + gx), th2);
                    const unsigned long long b = __ballot(code != 0);
                    if (code) gqueue[ngq + mask_rank(b)] = code | ((uint32_t)y << 2) | ((uint32_t)gx << 10);
                    ngq += __popcll(b);
                    gx += rstep; y += qstep;
                    if (gx >= ngx) { gx -= ngx; ++y; }
                }
            }
            __syncthreads();
            ORBX_PH(1, lane == 0);   // stage A
            // (2b) the other six pairs, per PIXEL: of a queued group's four pixels one or two are still alive, and a lane that tests a
            // whole group on packed halves works on all four (151 instructions per 64 groups against 61 + 45 per 64 pixels).
            // First the queued groups' alive pixels (compass code != 0, inside the zone) go to the pixel queue in raster order, entry
            // y<<6 | x | compass code<<12 ...
            int np = 0;
            for (int q0 = 0; q0 < ngq; q0 += 64) {
                uint32_t code = 0;
                int ent = 0;
                if (q0 + lane < ngq) {
                    const uint32_t e = gqueue[q0 + lane];
                    const int y = (e >> 2) & 63, gx = (e >> 10) & 63;
                    ent = (y << 6) + 4 * gx;                                                   // zone x of the group's pixel 0
                    code = e & 0x03030303u & (0xffffffffu >> (8 * max(0, 4 * gx + 4 - zw)));   // pixels at zone x >= zw are outside
                }
                if (__ballot(code != 0)) {
                    // alive pixels of this lane: bytes 0..3 are 0..3 -> one bit per pixel, n = how many
                    const uint32_t nz = (code | (code >> 1)) & 0x01010101u;
                    const int n = __popc(nz);
                    const unsigned long long c0 = __ballot(n & 1), c1 = __ballot(n & 2), c2 = __ballot(n & 4);
                    int pos = np + mask_rank(c0) + 2 * mask_rank(c1) + 4 * mask_rank(c2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { // branch-free: dead pixels go to the lane's dump slot
                        const int ac = (code >> (8 * j)) & 3;
                        unsigned short *dst = ac ? queue + pos : dump + lane;
                        *dst = (unsigned short)((ent + j) | (ac << 12));
                        pos += ac != 0;
                    }
                    np += __popcll(c0) + 2 * __popcll(c1) + 4 * __popcll(c2);
                }
            }
            // ... then a lane per pixel: max over the six pairs of min(p_k, p_k+8) and min of their max, the two threshold tests, ANDed with
            // the compass code; both polarities possible over all eight pairs cannot be a corner.  Survivors are compacted IN PLACE
            // (entry y<<6 | x | polarity<<12): a write lands at or below the position its lane read in this round, and the wave reads
            // before it writes.
            int nq = 0;
            for (int q0 = 0; q0 < np; q0 += 64) {
                int pol = 0, ent = 0;
                if (q0 + lane < np) {
                    const int e = queue[q0 + lane], x = e & 63, y = (e >> 6) & 63, ac = e >> 12;
                    ent = e & 0xfff;
                    const uint8_t *t = t0 + y * TS + x;
                    const int v = t[0];
                    const int a1 = t[3 * TS + 1], b1 = t[-3 * TS - 1], a2 = t[2 * TS + 2], b2 = t[-2 * TS - 2], a3 = t[TS + 3], b3 = t[-TS - 3];
                    const int a5 = t[-TS + 3], b5 = t[TS - 3], a6 = t[-2 * TS + 2], b6 = t[2 * TS - 2], a7 = t[-3 * TS + 1], b7 = t[3 * TS - 1];
                    const int lo = max(max(max(min(a1, b1), min(a2, b2)), min(a3, b3)), max(max(min(a5, b5), min(a6, b6)), min(a7, b7)));
                    const int hi = min(min(min(max(a1, b1), max(a2, b2)), max(a3, b3)), min(min(max(a5, b5), max(a6, b6)), max(a7, b7)));
                    const bool dark = (ac & 1) && v > lo + th, bright = (ac & 2) && hi > v + th;
                    pol = dark == bright ? 0 : dark ? 1 : 2;
                }
                const unsigned long long bm = __ballot(pol != 0);
                if (pol) queue[nq + mask_rank(bm)] = (unsigned short)(ent | (pol << 12));
                nq += __popcll(bm);
            }
            __syncthreads();
            ORBX_PH(2, lane == 0);   // stage B
            // the score map takes over the group queue's region: all zero, then (3) writes the survivors' scores
            for (int i = lane; i < ((zh + 2) * SS + 3) / 4; i += 64) reinterpret_cast<uint32_t *>(sc)[i] = 0;
            __syncthreads();
            // (3) arc score of the survivors
            for (int q0 = 0; q0 < nq; q0 += 64) {
                if (q0 + lane < nq) {
                    const int e = queue[q0 + lane], x = e & 63, y = (e >> 6) & 63;
                    // e_k = v - p_k (dark) or p_k - v (bright) = (p_k ^ m) - (v ^ m) with m = 0xff / 0; pairs (k, k+8) packed
                    // and biased in one xor-add each: halves (p ^ m) + (0x4100 - (v ^ m)) lie in 0x4001..0x41ff, no carry
                    const uint8_t *t = t0 + y * TS + x;
                    const uint32_t m1 = (e >> 12) == 1 ? 0xffu : 0u, m2 = m1 | (m1 << 16);
                    const uint32_t c1 = 0x4100u - ((uint32_t)t[0] ^ m1), c2 = c1 | (c1 << 16);
                    uint32_t X[8];
                    X[0] = (((uint32_t)t[3 * TS] | ((uint32_t)t[-3 * TS] << 16)) ^ m2) + c2;
                    X[1] = (((uint32_t)t[3 * TS + 1] | ((uint32_t)t[-3 * TS - 1] << 16)) ^ m2) + c2;
                    X[2] = (((uint32_t)t[2 * TS + 2] | ((uint32_t)t[-2 * TS - 2] << 16)) ^ m2) + c2;
                    X[3] = (((uint32_t)t[TS + 3] | ((uint32_t)t[-TS - 3] << 16)) ^ m2) + c2;
                    X[4] = (((uint32_t)t[3] | ((uint32_t)t[-3] << 16)) ^ m2) + c2;
                    X[5] = (((uint32_t)t[-TS + 3] | ((uint32_t)t[TS - 3] << 16)) ^ m2) + c2;
                    X[6] = (((uint32_t)t[-2 * TS + 2] | ((uint32_t)t[2 * TS - 2] << 16)) ^ m2) + c2;
                    X[7] = (((uint32_t)t[-3 * TS + 1] | ((uint32_t)t[3 * TS - 1] << 16)) ^ m2) + c2;
                    const int best = (int)fast_arc_best_packed(X) - 0x4100;
                    sc[(y + 1) * SS + x + 1] = (uint8_t)(best > th ? best - 1 : 0);
                }
            }
            __syncthreads();
            ORBX_PH(3, lane == 0);   // zero + arc score
            // (4) 3x3 strict non-max suppression on the survivors and (5) emission in queue (= raster) order, pass by pass of 64
            for (int q0 = 0; q0 < nq; q0 += 64) {
                bool ismax = false;
                uint32_t word = 0;
                if (q0 + lane < nq) {
                    const int e = queue[q0 + lane], x = e & 63, y = (e >> 6) & 63;
                    const uint8_t *q = sc + (y + 1) * SS + x + 1;
                    const int s = q[0];
                    // strictly above all eight neighbours (scores are >= 0, so this also says s > 0); no short-circuit
                    // branches: eight LDS bytes, three v_max3, one compare
                    const int m = max(max(max(q[-1], q[1]), max(q[-SS - 1], q[-SS])), max(max(q[-SS + 1], q[SS - 1]), max(q[SS], q[SS + 1])));
                    ismax = s > m;
                    word = ((uint32_t)(ci.dy + y + 3) << 20) | ((uint32_t)(ci.dx + x + 3) << 8) | (uint32_t)s;
                }
                const unsigned long long b = __ballot(ismax);
                const int pos = total + __popcll(b & lt);
                if (ismax && pos < ci.cap) out[pos] = word;
                total += __popcll(b);
            }
            ORBX_PH(4, lane == 0);   // NMS + emission
            if (total > 0 || minTh == iniTh) break; // empty at iniThFAST: once more at minThFAST (:820-824)
            __syncthreads();                        // the queues are rewritten
        }
    }
    if (lane == 0) cell_count[(size_t)f * cells_per_frame + c] = total;
    ORBX_PH(5, lane == 0);       // epilogue
    ORBX_PH(6, lane == 0);       // (count of waves x timer cost)
    ORBX_PH_END(lane == 0);
}

// ------------------------------------------------------------------- octree
// Exclusive scan of a[0..n) (and, if b != nullptr, of b[0..n)) in place by the
// whole 256-thread block; totals returned through ta / tb.  Each thread owns a
// contiguous chunk, wave-level shuffle scan, one cross-wave hop through LDS:
// three barriers in all.
template <int T>
__device__ void block_excl_scan2(int *a, int *b, int n, int *part, int &ta, int &tb)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __syncthreads();
    const int per = (n + T - 1) / T;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sa = 0, sb = 0;
    for (int i = lo; i < hi; ++i) { sa += a[i]; if (b) sb += b[i]; }
    // inclusive over the wave (this kernel is a chain of ~20 such scans, each waited for)
    const int ia = orbx::wave_incl_scan(sa), ib = b ? orbx::wave_incl_scan(sb) : 0;
    if (lane == 63) { part[w] = ia; part[T / 64 + w] = ib; }
    __syncthreads();
    int ba = 0, bb = 0, ga = 0, gb = 0;
#pragma unroll
    for (int i = 0; i < T / 64; ++i) {
        if (i < w) { ba += part[i]; bb += part[T / 64 + i]; }
        ga += part[i]; gb += part[T / 64 + i];
    }
    int ra = ba + ia - sa, rb = bb + ib - sb;
    for (int i = lo; i < hi; ++i) {
        const int va = a[i];
        a[i] = ra; ra += va;
        if (b) { const int vb = b[i]; b[i] = rb; rb += vb; }
    }
    ta = ga; tb = gb;
    __syncthreads();
}
template <int T>
__device__ int block_excl_scan(int *a, int n, int *part)
{
    int ta, tb;
    block_excl_scan2<T>(a, nullptr, n, part, ta, tb);
    return ta;
}

// Children of node box (x0,x1,y0,y1): ExtractorNode::DivideNode, ORBextractor.cc:481-509.
__device__ __forceinline__ void child_box(int x0, int x1, int y0, int y1, int q, int &cx0, int &cx1, int &cy0, int &cy1)
{
    const int mx = x0 + (int)ceilf((float)(x1 - x0) / 2);
    const int my = y0 + (int)ceilf((float)(y1 - y0) / 2);
    cx0 = (q & 1) ? mx : x0;
    cx1 = (q & 1) ? x1 : mx;
    cy0 = (q & 2) ? my : y0;
    cy1 = (q & 2) ? y1 : my;
}

// DistributeOctTree (ORBextractor.cc:539-763) for one (level, frame) per block.
// Each pass of the reference's list surgery is one "round": which nodes split,
// where their children land in the list, and where the untouched nodes move are
// all prefix sums (tests/octree_model.py is the same formulation in Python and is
// checked against the literal list-based oracle).  Equal-size ties in the
// largest-first phase use creation order.
// Keys (candidates) live in LDS when the level has at most `kcap` of them, else in
// the per-frame HBM workspace (flat pointers serve both).
template <int T, int KPT, int KB>
__global__ __launch_bounds__(T) void k_octree(const LevelInfo *__restrict__ L, const CellInfo *__restrict__ cells,
                                                  const int *__restrict__ cell_count, int cells_per_frame,
                                                  const uint32_t *__restrict__ cands, size_t cands_per_frame,
                                                  uint32_t *__restrict__ kpos_all, unsigned short *__restrict__ knode_all,
                                                  uint8_t *__restrict__ kq_all, size_t keys_per_frame,
                                                  uint32_t *__restrict__ sel_all, int sel_per_frame,
                                                  int *__restrict__ level_count, int *__restrict__ level_ncand,
                                                  int nlevels, int NC, int maxcells, int kcap, int kshift)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int *p = reinterpret_cast<int *>(smem);
    int *part = p;          p += 2 * (T / 64);
    int *s_cell = p;        p += (maxcells + 1 + 3) & ~3;
    int *s_coff = p;        p += (maxcells + 1 + 3) & ~3; // each cell's slot in the candidate buffer
    int *bx[2] = {p, p + NC};  p += 2 * NC; // x0 | x1<<16
    int *by[2] = {p, p + NC};  p += 2 * NC; // y0 | y1<<16
    int *cnt[2] = {p, p + NC}; p += 2 * NC;
    // narrow per-node fields share dwords (NC is a multiple of 4): 16 NC dwords of node state in all
    unsigned short *seq[2] = {reinterpret_cast<unsigned short *>(p), reinterpret_cast<unsigned short *>(p) + NC}; p += NC; // < NC
    int *cc = p;            p += 4 * NC;
    uint8_t *nne = reinterpret_cast<uint8_t *>(p), *eexp = nne + NC, *split = nne + 2 * NC; p += NC; // 0..4, 0..4, flag
    unsigned short *order = reinterpret_cast<unsigned short *>(p), *bstart = order + NC;      p += NC; // node / list positions
    int *ebase = p;         p += NC;
    int *a1 = p;            p += NC;
    int *a2 = p;            p += NC;
    uint32_t *l_kpos = reinterpret_cast<uint32_t *>(p);                     p += kcap;
    unsigned short *l_knode = reinterpret_cast<unsigned short *>(p);        p += (kcap + 1) / 2;
    uint8_t *l_kq = reinterpret_cast<uint8_t *>(p);
    __shared__ int s_nproc;

    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    ORBX_PH_INIT(3);
#ifdef ORBX_PHASE_TIMING
    if (tid == 0) { for (int i = 0; i < 8; ++i) ph_rec[i] = 0; ph_rec[12] = l; }
#endif
    const LevelInfo lv = L[l];
    const int N = lv.N;

    // gather the level's candidates in cell row-major order (vToDistributeKeys)
    // (cand_off goes to LDS too, so that the gather below has one dependent global load per candidate, not two)
    for (int c = tid; c < lv.ncells; c += T) {
        const int n = cell_count[(size_t)f * cells_per_frame + lv.cell_base + c];
        const CellInfo ci = cells[lv.cell_base + c];
        s_cell[c] = n < ci.cap ? n : ci.cap;
        s_coff[c] = ci.cand_off;
    }
    const int M = block_excl_scan<T>(s_cell, lv.ncells, part);
    if (tid == 0) s_cell[lv.ncells] = M;
    const bool in_lds = M <= kcap;
    uint32_t *kpos = in_lds ? l_kpos : kpos_all + (size_t)f * keys_per_frame + lv.key_base;
    unsigned short *knode = in_lds ? l_knode : knode_all + (size_t)f * keys_per_frame + lv.key_base;
    uint8_t *kq = in_lds ? l_kq : kq_all + (size_t)f * keys_per_frame + lv.key_base;
    uint32_t *kpos_out = kpos_all + (size_t)f * keys_per_frame + lv.key_base; // candidates stay readable for staged tests
    for (int i = tid; i < NC; i += T) cnt[0][i] = 0;
    __syncthreads();
    // Up to KPT x 256 candidates (every level of a usual frame) the keys never leave the registers: thread t owns the keys
    // t, t + 256, ...; a key's packed position never changes and its node is rewritten by its owner only.  A round's two passes
    // over the keys were 60 % of a workgroup's life as loops over LDS / HBM arrays -- every iteration a chain of dependent
    // reads (node of the key -> state of the node) on a workgroup of four waves; with the keys in registers all of a thread's
    // node-state reads are in flight together.
    const bool regs = M <= KPT * T;
    uint32_t kp[KPT];
    int kn[KPT], kqv[KPT];
    if (regs) {
        // cell of key k = largest c with s_cell[c] <= k: the bisections of a thread's keys run in lockstep, one LDS read of each
        // in flight per step (one after the other they were 8 x 9 dependent reads)
        // (KB keys at a time here and in the passes below: batching all eight took 187 VGPRs, and what the kernel holds on
        // a CU for 60 us is what the other contexts' kernels cannot have -- the step lost 4 % to it)
#pragma unroll
        for (int u = 0; u < KPT; ++u) { kp[u] = 0; kn[u] = 0; kqv[u] = 0; }
#pragma unroll
        for (int h = 0; h < KPT; h += KB) {
            if (h * T >= M) break;
            int lo[KB], hi[KB];
#pragma unroll
            for (int v = 0; v < KB; ++v) { lo[v] = 0; hi[v] = lv.ncells; }
            for (int span = lv.ncells; span > 1; span = (span + 1) >> 1) {
#pragma unroll
                for (int v = 0; v < KB; ++v) {
                    const int mid = (lo[v] + hi[v]) >> 1;
                    if (hi[v] - lo[v] > 1) { if (s_cell[mid] <= min(tid + (h + v) * T, M - 1)) lo[v] = mid; else hi[v] = mid; }
                }
            }
#pragma unroll
            for (int v = 0; v < KB; ++v) {
                const int k = tid + (h + v) * T;
                if (k < M) kp[h + v] = cands[(size_t)f * cands_per_frame + s_coff[lo[v]] + (k - s_cell[lo[v]])];
            }
        }
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const int k = tid + u * T;
            if (k < M) {
                kpos_out[k] = kp[u];
                const float x = (float)((kp[u] >> 8) & 0xfffu);
                kn[u] = (int)(x / lv.hX); // vpIniNodes[kp.pt.x/hX], :569
                atomicAdd(&cnt[0][kn[u]], 1);
            }
        }
    } else
    for (int k0 = tid; k0 < M; k0 += 4 * T) { // four candidates per thread in flight: the loads are a dependent chain each
        uint32_t pk4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * T;
            pk4[u] = 0;
            if (k < M) {
                int lo = 0, hiC = lv.ncells; // largest c with s_cell[c] <= k
                while (hiC - lo > 1) {
                    const int mid = (lo + hiC) >> 1;
                    if (s_cell[mid] <= k) lo = mid; else hiC = mid;
                }
                pk4[u] = cands[(size_t)f * cands_per_frame + s_coff[lo] + (k - s_cell[lo])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * T;
            if (k < M) {
                const uint32_t pk = pk4[u];
                kpos[k] = pk;
                if (in_lds) kpos_out[k] = pk;
                const float x = (float)((pk >> 8) & 0xfffu);
                const int root = (int)(x / lv.hX); // vpIniNodes[kp.pt.x/hX], :569
                knode[k] = (unsigned short)root;
                atomicAdd(&cnt[0][root], 1);
            }
        }
    }
    __syncthreads();
    ORBX_PHA(0, tid == 0);   // cells, scan, candidate gather
#ifdef ORBX_PHASE_TIMING
    if (tid == 0) ph_rec[11] = M;
#endif
    // roots (:552-563), empty ones erased (:574-585)
    for (int i = tid; i < lv.nIni; i += T) a1[i] = cnt[0][i] > 0 ? 1 : 0;
    int S = block_excl_scan<T>(a1, lv.nIni, part);
    for (int i = tid; i < lv.nIni; i += T) {
        if (cnt[0][i] > 0) {
            const int x0 = (int)(lv.hX * (float)i), x1 = (int)(lv.hX * (float)(i + 1));
            const int pos = a1[i];
            bx[1][pos] = x0 | (x1 << 16);
            by[1][pos] = 0 | (lv.H << 16);
            cnt[1][pos] = cnt[0][i];
            seq[1][pos] = 0;
        }
    }
    if (regs) {
#pragma unroll
        for (int u = 0; u < KPT; ++u) { if (u * T >= M) break; kn[u] = tid + u * T < M ? a1[kn[u]] : 0; }
    } else
    for (int k = tid; k < M; k += T) knode[k] = (unsigned short)a1[knode[k]];
    __syncthreads();
    ORBX_PHA(1, tid == 0);   // roots
    int cur = 1, mode = 1;

    while (true) {
        int *cx = bx[cur], *cy = by[cur], *cn = cnt[cur];
        unsigned short *cs = seq[cur];
        for (int s = tid; s < S; s += T) {
            cc[4 * s] = cc[4 * s + 1] = cc[4 * s + 2] = cc[4 * s + 3] = 0;
            split[s] = 0;
        }
        __syncthreads();
        // children counts of every expandable node (DivideNode, :511-526)
        if (regs) {
#pragma unroll
            for (int h = 0; h < KPT; h += KB) {
                if (h * T >= M) break;
                int ncn[KB], ncx[KB], ncy[KB];   // the node's state, read for KB of the thread's keys at once
#pragma unroll
                for (int v = 0; v < KB; ++v) { ncn[v] = cn[kn[h + v]]; ncx[v] = cx[kn[h + v]]; ncy[v] = cy[kn[h + v]]; }
#pragma unroll
                for (int v = 0; v < KB; ++v) {
                    const int u = h + v;
                    if (tid + u * T < M && ncn[v] > 1) {
                        const float x = (float)((kp[u] >> 8) & 0xfffu), y = (float)(kp[u] >> 20);
                        const int x0 = ncx[v] & 0xffff, x1 = ncx[v] >> 16, y0 = ncy[v] & 0xffff, y1 = ncy[v] >> 16;
                        const int mx = x0 + (int)ceilf((float)(x1 - x0) / 2);
                        const int my = y0 + (int)ceilf((float)(y1 - y0) / 2);
                        kqv[u] = (x < (float)mx ? 0 : 1) + (y < (float)my ? 0 : 2);
                        atomicAdd(&cc[4 * kn[u] + kqv[u]], 1);
                    }
                }
            }
        } else
        for (int k = tid; k < M; k += T) {
            const int s = knode[k];
            if (cn[s] > 1) {
                const uint32_t pk = kpos[k];
                const float x = (float)((pk >> 8) & 0xfffu), y = (float)(pk >> 20);
                const int x0 = cx[s] & 0xffff, x1 = cx[s] >> 16, y0 = cy[s] & 0xffff, y1 = cy[s] >> 16;
                const int mx = x0 + (int)ceilf((float)(x1 - x0) / 2);
                const int my = y0 + (int)ceilf((float)(y1 - y0) / 2);
                const int q = (x < (float)mx ? 0 : 1) + (y < (float)my ? 0 : 2);
                kq[k] = (uint8_t)q;
                atomicAdd(&cc[4 * s + q], 1);
            }
        }
        __syncthreads();
        ORBX_PHA(2, tid == 0);   // children counts (pass over the keys)
        // per node: non-empty / expandable children; scanned together with the candidate flag
        for (int s = tid; s < S; s += T) {
            int a = 0, b = 0;
            const bool cand = cn[s] > 1;
            if (cand) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a += cc[4 * s + q] > 0;
                    b += cc[4 * s + q] > 1;
                }
            }
            nne[s] = a;
            eexp[s] = b;
            a1[s] = cand ? 1 : 0;      // -> candidate index in list order
            a2[s] = a | (b << 16);     // -> prefix of (non-empty, expandable) children in list order
        }
        int E, PE;
        block_excl_scan2<T>(a1, a2, S, part, E, PE);
        ORBX_PHA(3, tid == 0);   // node flags + scans
        if (E == 0) break;
        int F, Etot, nns;
        if (mode == 1) {
            // every expandable node splits, in list order (:606-665)
            F = PE & 0xffff; Etot = PE >> 16;
            for (int s = tid; s < S; s += T) {
                if (cn[s] > 1) {
                    split[s] = 1;
                    bstart[s] = F - ((a2[s] & 0xffff) + nne[s]); // children pushed to the front, last processed first
                    ebase[s] = a2[s] >> 16;
                }
                a1[s] = s - a1[s]; // rank among the nodes that stay
            }
            nns = S - E;
            __syncthreads();
        } else {
            // largest first, later-created first among equals (:684-732): rank = number of expandable nodes with a
            // larger (size, creation seq) key.  Keys (size << kshift | seq, seq < NC <= 2^kshift; 0 for the others) are packed first so that
            // the S x S comparison reads one LDS dword per four nodes (it was 45 % of the kernel as a scalar loop).
            unsigned *okey = reinterpret_cast<unsigned *>(ebase); // free until the splits are numbered below
            if (kshift) {
                for (int s = tid; s < ((S + 3) & ~3); s += T) okey[s] = (s < S && cn[s] > 1) ? ((unsigned)cn[s] << kshift) | cs[s] : 0u;
                __syncthreads();
                for (int s = tid; s < S; s += T) {
                    const unsigned k0 = okey[s];
                    if (k0) {
                        int r = 0;
                        for (int s2 = 0; s2 < S; s2 += 4) {
                            const uint4 k4 = *reinterpret_cast<const uint4 *>(okey + s2);
                            r += (k4.x > k0) + (k4.y > k0) + (k4.z > k0) + (k4.w > k0);
                        }
                        order[r] = (unsigned short)s;
                    }
                }
            } else {
                // a level with 2^21 candidate slots or more (frames beyond ~8 M pixels): size and sequence do not fit one dword,
                // the pairs are compared as they are (the plain loop the packed keys replaced: slower, exact)
                for (int s = tid; s < S; s += T) {
                    const int c0 = cn[s], q0 = cs[s];
                    if (c0 > 1) {
                        int r = 0;
                        for (int s2 = 0; s2 < S; ++s2) {
                            const int c2 = cn[s2];
                            r += c2 > 1 && (c2 > c0 || (c2 == c0 && (int)cs[s2] > q0));
                        }
                        order[r] = (unsigned short)s;
                    }
                }
            }
            __syncthreads(); // okey (= ebase) is rewritten below
            if (tid == 0) s_nproc = E;
            __syncthreads();
            for (int r = tid; r < E; r += T) a2[r] = nne[order[r]] | (eexp[order[r]] << 16);
            int dummy;
            block_excl_scan2<T>(a2, nullptr, E, part, PE, dummy);
            for (int r = tid; r < E; r += T) // stop at the first split that reaches N leaves (:730-731)
                if (S + (a2[r] & 0xffff) - r + nne[order[r]] - 1 >= N) atomicMin(&s_nproc, r + 1);
            __syncthreads();
            const int nproc = s_nproc;
            // totals over the processed prefix
            const int lastr = nproc - 1, lasts = order[lastr];
            F = (a2[lastr] & 0xffff) + nne[lasts];
            Etot = (a2[lastr] >> 16) + eexp[lasts];
            for (int r = tid; r < nproc; r += T) {
                const int s = order[r];
                split[s] = 1;
                bstart[s] = F - ((a2[r] & 0xffff) + nne[s]);
                ebase[s] = a2[r] >> 16;
            }
            __syncthreads();
            for (int s = tid; s < S; s += T) a1[s] = split[s] ? 0 : 1;
            nns = block_excl_scan<T>(a1, S, part); // a1[s] = rank among the nodes that stay
        }
        const int S2 = F + nns;
        int *nx = bx[cur ^ 1], *ny = by[cur ^ 1], *nn = cnt[cur ^ 1];
        unsigned short *ns = seq[cur ^ 1];
        ORBX_PHA(4, tid == 0);   // which nodes split (mode 1: all; mode 2: ranking, largest first)
        for (int s = tid; s < S; s += T) {
            if (split[s]) {
                const int x0 = cx[s] & 0xffff, x1 = cx[s] >> 16, y0 = cy[s] & 0xffff, y1 = cy[s] >> 16;
                int e = ebase[s];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cc[4 * s + q];
                    if (c > 0) {
                        int after = 0;
                        for (int q2 = q + 1; q2 < 4; ++q2) after += cc[4 * s + q2] > 0;
                        const int pos = bstart[s] + after; // list shows n4,n3,n2,n1
                        int c0, c1, d0, d1;
                        child_box(x0, x1, y0, y1, q, c0, c1, d0, d1);
                        nx[pos] = c0 | (c1 << 16);
                        ny[pos] = d0 | (d1 << 16);
                        nn[pos] = c;
                        ns[pos] = c > 1 ? e++ : 0;
                    }
                }
            } else {
                const int pos = F + a1[s];
                nx[pos] = cx[s]; ny[pos] = cy[s]; nn[pos] = cn[s]; ns[pos] = cs[s];
            }
        }
        if (regs) {
#pragma unroll
            for (int h = 0; h < KPT; h += KB) {
                if (h * T >= M) break;
                int nsp[KB], nbs[KB], na1[KB];
                int4 ncc[KB];
#pragma unroll
                for (int v = 0; v < KB; ++v) {
                    const int s = kn[h + v];
                    nsp[v] = split[s]; nbs[v] = bstart[s]; na1[v] = a1[s];
                    ncc[v] = *reinterpret_cast<const int4 *>(cc + 4 * s);
                }
#pragma unroll
                for (int v = 0; v < KB; ++v) {
                    const int u = h + v, q = kqv[u];
                    const int after = (q < 1 && ncc[v].y > 0) + (q < 2 && ncc[v].z > 0) + (q < 3 && ncc[v].w > 0);   // later non-empty children
                    kn[u] = tid + u * T < M ? (nsp[v] ? nbs[v] + after : F + na1[v]) : 0;
                }
            }
        } else
        for (int k = tid; k < M; k += T) {
            const int s = knode[k];
            int pos;
            if (split[s]) {
                const int q = kq[k];
                int after = 0;
                for (int q2 = q + 1; q2 < 4; ++q2) after += cc[4 * s + q2] > 0;
                pos = bstart[s] + after;
            } else {
                pos = F + a1[s];
            }
            knode[k] = (unsigned short)pos;
        }
        __syncthreads();
        ORBX_PHA(5, tid == 0);   // new node list + keys to their new nodes (pass over the keys)
#ifdef ORBX_PHASE_TIMING
        if (tid == 0) ph_rec[7] += 1;
#endif
        const int prevS = S;
        S = S2;
        cur ^= 1;
        if (S >= N || S == prevS) break;          // :669-672, :734-735
        if (mode == 1 && S + 3 * Etot > N) mode = 2; // :673
    }
    __syncthreads();
    // best response per leaf, first candidate wins ties (:741-760)
    int *best = a2;
    for (int s = tid; s < S; s += T) best[s] = 0;
    __syncthreads();
    if (regs) {
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            if (tid + u * T < M)
                atomicMax(reinterpret_cast<unsigned *>(&best[kn[u]]), ((kp[u] & 0xffu) << 24) | (0xffffffu - (unsigned)(tid + u * T)));
    } else
    for (int k = tid; k < M; k += T)
        atomicMax(reinterpret_cast<unsigned *>(&best[knode[k]]), ((kpos[k] & 0xffu) << 24) | (0xffffffu - (unsigned)k));
    __syncthreads();
    uint32_t *sel = sel_all + (size_t)f * sel_per_frame + lv.sel_base;
    const uint32_t *kfin = regs ? kpos_out : kpos;   // (register keys: the gather left the packed positions in the frame's workspace)
    for (int s = tid; s < S; s += T) sel[s] = kfin[0xffffffu - ((unsigned)best[s] & 0xffffffu)];
    if (tid == 0) {
        level_count[f * nlevels + l] = S;
        level_ncand[f * nlevels + l] = M;
    }
    ORBX_PHA(6, tid == 0);   // best response per leaf, output
    ORBX_PH_END(tid == 0);
}

// --------------------------------------------------------------------- blur


// Levels 1 .. n-1 of the pyramid in ONE launch -- the form SMALL batches take (orbx_extract_batch: <= 6 frames, the live tracker's one frame
// per call: 22 us against the seven dependent per-level launches' 52; in the pipelined 64-frame step it was slower than they are,
// profiles/r04_notes.md, and is not used there).  A workgroup carries one spatial tile through all levels in LDS: it loads
// its rectangle of level 0, computes from it its rectangle of level 1 (the same fixed-point arithmetic, 4 pixels per lane from an
// 8-byte source window -- taken from three aligned LDS dwords), stores the part of the padded level 1 it owns (border pixels through
// the reflected coordinate, as the per-level kernel does), computes level 2 from level 1, and so on.  The rectangles (host table,
// plan_frame) are what the tile owns at a level plus what its deeper levels read: neighbours overlap by a few pixels per level and
// recompute them (1.3-1.5 x the arithmetic of the exact chain), and nobody waits for anybody.
__device__ __forceinline__ int chain_pitch(int nw) { return ((nw + 15) & ~15) + 16; }
__global__ __launch_bounds__(256) void k_pyr_chain(uint8_t *__restrict__ pyr, size_t frame_bytes, const LevelInfo *__restrict__ L, int nlevels,
                                                   const ChainTile *__restrict__ tiles, const uint4 *__restrict__ tabs,
                                                   const int2 *__restrict__ tab_span, int ldsA, int ldsB)
{
    extern __shared__ __align__(16) uint8_t chain_lds[];
    const int tid = threadIdx.x, f = blockIdx.y;
    ORBX_PH_INIT(2);
#ifdef ORBX_PHASE_TIMING
    if (tid == 0) { ph_rec[12] = 0; ph_rec[11] = 1; ph_rec[1] = ph_rec[2] = ph_rec[3] = 0; }
#endif
    const ChainTile &T = tiles[blockIdx.x];
    uint8_t *const fp = pyr + (size_t)f * frame_bytes;
    // the tile's coefficient tables, prepared by the host as one block (plan_frame): for every level the rows {sy0 | sy1 << 16,
    // b0 | b1 << 16} and then the columns {sx, a0 | a1 << 16} of its rectangle, source coordinates relative to the source rectangle.
    // The rectangles and the levels' geometry are staged too (read per level through the scalar cache they were a dependent memory
    // round trip at the top of every level)
    uint4 *const s_rect = reinterpret_cast<uint4 *>(chain_lds + ldsA + ldsB);          // [MAXL] ChainRect, [MAXL] {w, h, stride, off}
    int2 *const s_tab = reinterpret_cast<int2 *>(s_rect + 2 * MAXL);
    const int2 span = tab_span[blockIdx.x];      // first 16-byte unit, units
    if (tid < MAXL) s_rect[tid] = reinterpret_cast<const uint4 *>(&T)[tid];
    else if (tid < 2 * MAXL) s_rect[tid] = *reinterpret_cast<const uint4 *>(&L[tid - MAXL]);
    auto rect_at = [&](int l) {
        uint4 v = s_rect[l];
        v.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.x); v.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y);
        v.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.z); v.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w);
        return __builtin_bit_cast(ChainRect, v);
    };
    struct Geo { int w, h, stride, off; };
    auto geo_at = [&](int l) {
        uint4 v = s_rect[MAXL + l];
        Geo g;
        g.w = __builtin_amdgcn_readfirstlane((int)v.x); g.h = __builtin_amdgcn_readfirstlane((int)v.y);
        g.stride = __builtin_amdgcn_readfirstlane((int)v.z); g.off = __builtin_amdgcn_readfirstlane((int)v.w);
        return g;
    };
    {   // level 0: the rectangle in 16-byte pieces (nx0 % 4 == 0: dword-aligned in memory; the columns past nw up to the pitch are read
        // too: they lie inside the padded row), all of a lane's pieces in flight together; the tables travel meanwhile
        const ChainRect r0 = T.r[0];
        const LevelInfo l0 = L[0];
        const int p16 = chain_pitch(r0.nw) >> 4, n = p16 * r0.nh;
        const uint8_t *src = fp + l0.off + PADX + r0.nx0 + (size_t)(r0.ny0 + EDGE) * l0.stride;
        for (int i = tid; i < span.y; i += 256) reinterpret_cast<uint4 *>(s_tab)[i] = tabs[span.x + i];
        for (int i0 = tid; i0 < n; i0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(i0 + 256 * k, n - 1), row = i / p16, c = i - row * p16;
                __builtin_memcpy(&v[k], src + (uint32_t)(row * l0.stride + 16 * c), 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + 256 * k < n) reinterpret_cast<uint4 *>(chain_lds)[i0 + 256 * k] = v[k];
        }
    }
    __syncthreads();
    int nytot = 0;
    for (int l = 1; l < nlevels; ++l) nytot += rect_at(l).nh;
    const int2 *const s_yt = s_tab, *const s_xt = s_tab + ((nytot + 1) & ~1);
    ORBX_PH(0, tid == 0);    // tables + level 0 in LDS
    int yo = 0, xo = 0;
    ChainRect rp = rect_at(0);
    for (int l = 1; l < nlevels; ++l) {
        const ChainRect rc = rect_at(l);
        const Geo lv = geo_at(l);
        const uint8_t *src = chain_lds + ((l - 1) & 1 ? ldsA : 0);
        uint8_t *dst = chain_lds + (l & 1 ? ldsA : 0);
        const int pp = chain_pitch(rp.nw), pc = chain_pitch(rc.nw);
        uint8_t *const gout = fp + lv.off;
        // ---- compute the rectangle: lane = a group of 4 columns x a band of consecutive rows (G groups side by side, G a power of two,
        // 256 / G bands).  A source row's horizontal pass (4 pixels) stays in registers: consecutive output rows share source rows
        // (sy1 of one is sy0 of the next four times out of five), so a row is read and interpolated 1.2 times instead of 2.  A dword
        // that the tile owns and that holds four interior pixels goes to memory straight from the registers.
        const int ngrp = (rc.nw + 3) >> 2;
        int gs = 0;
        while ((1 << gs) < ngrp) ++gs;
        const int g = tid & ((1 << gs) - 1), band = tid >> gs, nband = 256 >> gs;
        const int rb = (rc.nh + nband - 1) / nband, r_lo = band * rb, r_hi = min(r_lo + rb, (int)rc.nh);
        if (g < ngrp && r_lo < r_hi) {
            const int2 *xg = s_xt + xo + 4 * g;
            const int2 x0 = xg[0], x1 = xg[1], x2 = xg[2], x3 = xg[3];
            const uint32_t coef[4] = {(uint32_t)x0.y, (uint32_t)x1.y, (uint32_t)x2.y, (uint32_t)x3.y};
            const int col0 = x0.x, a4 = col0 >> 2, o = col0 & 3;
            const uint32_t sel[4] = {0x0c010c00u, 0x0c010c00u + (uint32_t)(x1.x - col0) * 0x00010001u,
                                     0x0c010c00u + (uint32_t)(x2.x - col0) * 0x00010001u, 0x0c010c00u + (uint32_t)(x3.x - col0) * 0x00010001u};
            const int2 *yl = s_yt + yo;
            struct H4 { int v[4]; };
            auto horiz = [&](int srow) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(src + srow * pp) + a4;
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                const uint32_t wx = __builtin_amdgcn_alignbyte(d1, d0, o), wy = __builtin_amdgcn_alignbyte(d2, d1, o);
                H4 h;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    h.v[b] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(wy, wx, sel[b])),
                                                         __builtin_bit_cast(pyr_u16x2, coef[b]), 0u, false);
                return h;
            };
            // the dword of this group in the padded row, and whether the tile stores it from here
            const int gx = rc.nx0 + 4 * g, xw = (PADX + gx) >> 2;
            const bool own_x = xw >= rc.oxw0 && xw < rc.oxw1 && gx + 3 < lv.w;
            int prev = -1;
            H4 hp = {{0, 0, 0, 0}};
            for (int r = r_lo; r < r_hi; ++r) {
                const int2 yy = yl[r];
                const int s0 = yy.x & 0xffff, s1 = (int)((uint32_t)yy.x >> 16);
                const H4 h0 = s0 == prev ? hp : horiz(s0);
                const H4 h1 = s1 == s0 ? h0 : horiz(s1);
                hp = h1; prev = s1;
                const uint32_t bs0 = ((uint32_t)yy.y & 0xffffu) << 12, bs1 = ((uint32_t)yy.y >> 16) << 12;
                uint32_t out = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) out |= resize_vertical(h0.v[b], h1.v[b], bs0, bs1) << (8 * b);
                *reinterpret_cast<uint32_t *>(dst + r * pc + 4 * g) = out;
                const int row = rc.ny0 + r + EDGE;
                if (own_x && row >= rc.or0 && row < rc.or1) *reinterpret_cast<uint32_t *>(gout + (uint32_t)(row * lv.stride + 4 * xw)) = out;
            }
        }
        __syncthreads();
        ORBX_PHA(1, tid == 0);   // compute (all levels)
        // ---- what is left of the tile's share of the padded level: the border (reflected coordinates) -- rows above / below the image
        // in full, beside the image the dwords that are not four interior pixels.  Lane = a dword of the padded row
        const int ndw = rc.oxw1 - rc.oxw0, nrow = rc.or1 - rc.or0;
        const int ilo = PADX >> 2, ihi = (PADX + lv.w) >> 2;           // dwords [ilo, ihi) hold four interior pixels each
        const bool side = rc.oxw0 < ilo || rc.oxw1 > ihi, cap = rc.or0 < EDGE || rc.or1 > lv.h + EDGE;
        if (ndw > 0 && nrow > 0 && (side || cap)) {
            int ws = 0;
            while ((1 << ws) < ndw) ++ws;
            const int c = tid & ((1 << ws) - 1), r1st = tid >> ws, wpass = 256 >> ws;
            if (c < ndw) {
                const int xw = rc.oxw0 + c, px0 = xw * 4 - PADX;
                const bool interior = xw >= ilo && xw < ihi;
                int cx[4]; uint32_t keep = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int px = min(max(px0 + b, -EDGE), lv.w + EDGE - 1);
                    px = px < 0 ? -px : (px >= lv.w ? 2 * (lv.w - 1) - px : px);       // one fold is enough: |px| <= EDGE < w
                    cx[b] = px - rc.nx0;
                    keep |= (px0 + b >= -EDGE && px0 + b < lv.w + EDGE) ? 0xffu << (8 * b) : 0u;
                }
                for (int r = r1st; r < nrow; r += wpass) {
                    const int row = rc.or0 + r, yy0 = row - EDGE;
                    const bool inrow = yy0 >= 0 && yy0 < lv.h;
                    if (interior && inrow) continue;                                   // stored from the registers above
                    const int y = yy0 < 0 ? -yy0 : (yy0 >= lv.h ? 2 * (lv.h - 1) - yy0 : yy0);
                    const uint8_t *sr = dst + (y - rc.ny0) * pc;
                    uint32_t v;
                    if (interior) v = *reinterpret_cast<const uint32_t *>(sr + cx[0]);   // nx0 % 4 == 0: an aligned dword
                    else v = ((uint32_t)sr[cx[0]] | ((uint32_t)sr[cx[1]] << 8) | ((uint32_t)sr[cx[2]] << 16) | ((uint32_t)sr[cx[3]] << 24)) & keep;
                    *reinterpret_cast<uint32_t *>(gout + (uint32_t)(row * lv.stride + 4 * xw)) = v;
                }
            }
        }
        // (no barrier here: the next level reads `dst`, which nobody writes any more, and writes the other buffer, which nobody
        // reads any more -- its last readers passed the barrier above)
        yo += rc.nh; xo += (rc.nw + 3) & ~3;
        rp = rc;
        ORBX_PHA(2, tid == 0);   // border stores (all levels; thread 0's share)
    }
    ORBX_PH_END(tid == 0);
}

// GaussianBlur(7x7, sigma 2, REFLECT_101) in 8.8 fixed point: horizontal 7 taps exact in
// u16, vertical 7 taps exact in u32, min((v + 2^15) >> 16, 255).  The padded pyramid already holds the
// reflected border, so no index clamping.
//
// Both passes are products with a banded Toeplitz matrix of the taps, in exact integers on the i8 matrix
// cores (v_mfma_i32_32x32x32_i8): one wave blurs one 32 x 32 input tile,
//   H''[y'][x] = sum_k (P[y'][k] - 128) T[k - x - 1] + 128          (A = pixel rows as loaded, B = Toeplitz)
//   V^T[x][y]  = sum_y' H[y'][x] T[y' - y]                            (A = the accumulator of the first product)
// H'' fits 16 signed bits for tap sums up to 257 and goes into the second product as two i8 planes
// (H'' >> 8 and (H'' & 255) - 128, eight v_perm_b32 + xor per 16 values); the constant terms of both offsets
// ride in the accumulator seeds.  The first result has its column on the lane and its rows in the lane's 16
// registers, which is exactly an A (or B) operand of the next MFMA -- no LDS, no lane movement -- and since a
// dot product pairs operand bytes by position, the Toeplitz fragments (a 2-KB table built on the host) are
// simply written in the order in which the lane holds its rows; the hardware's K order never enters.  Taking
// the accumulator as A transposes the result: a lane ends with 16 x-consecutive outputs of ONE row, four per
// dword store.  A tile yields 24 x 26 valid outputs (7 taps of support inside 32 inputs, dword-aligned in x).
constexpr int BLUR_TW = 24, BLUR_TH = 26;
typedef int blur_v4i __attribute__((ext_vector_type(4)));
typedef int blur_v16i __attribute__((ext_vector_type(16)));
// A wave blurs a strip of BLUR_TPW x-adjacent tiles (96 x 26 outputs) and stages it in LDS, so that the rows
// leave as 16-byte pieces of 96 contiguous bytes instead of 4-byte pieces in 32 different lines.
constexpr int BLUR_TPW = 4, BLUR_SW = BLUR_TPW * BLUR_TW, BLUR_LROW = 112;   // strip width; LDS row stride (7 x 16 B)
constexpr int BLUR_IROW = 144;                                                 // LDS row stride of the 128-byte input window
// WIDE: tap sum 257 (blur_variant 1): H' = sum (P - 128) T needs a +128 bias to fit 16 signed bits, and the
// result can exceed 255.  Otherwise the first product starts from 0 and nothing saturates.
template <bool WIDE>
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                              size_t frame_bytes, size_t blur_frame_bytes, const LevelInfo *__restrict__ L,
                                              const BlurTile *__restrict__ tiles, int ntiles,
                                              const uint4 *__restrict__ frag, int seed2)
{
    __shared__ __align__(16) uint8_t stage[4][BLUR_TH * BLUR_LROW];
    __shared__ __align__(16) uint8_t win[4][32 * BLUR_IROW];
    int f, wg;
    xcd_frame_item(f, wg);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ti = wg * 4 + wave;
    if (ti >= ntiles) return;
    const int r = lane & 31, h = lane >> 5;
    blur_v16i c1, c2, z;
#pragma unroll
    for (int g = 0; g < 16; ++g) { c1[g] = WIDE ? 128 : 0; c2[g] = seed2; z[g] = 0; }
    const BlurTile bt = tiles[ti];   // wave-uniform index: a scalar load
    struct { int w, h, stride; } lv = {bt.w, bt.h, bt.stride};
    const size_t base = (size_t)f * frame_bytes + bt.off + PADX;
    // the Toeplitz fragments (per-lane constants) are requested HERE, beside the window: left to the compiler they sank below the
    // window's LDS stores, a memory round trip of their own in front of the first MFMA
    uint4 f1 = frag[2 * lane], f2 = frag[2 * lane + 1];
    uint8_t *const st = stage[wave], *const wn = win[wave];
    // input window: rows y0-3 .. y0+28, columns x0-16 .. x0+111 as 8 aligned 16-byte pieces per row (whole 128-byte
    // lines per row instead of 32 bytes of 32 different lines per load).  Rows past the padded level and pieces past
    // the padded row are clamped: they only reach outputs that are not stored.
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int i = lane + 64 * s4, row = i >> 3, pc = i & 7;
        const int yin = min(bt.y0 - 3 + row, lv.h + EDGE - 1), xo = min(PADX + bt.x0 - 16 + 16 * pc, lv.stride - 16);
        *reinterpret_cast<uint4 *>(wn + row * BLUR_IROW + 16 * pc) =
            *reinterpret_cast<const uint4 *>(pyr + (base - PADX) + (uint32_t)(__mul24(yin + EDGE, lv.stride) + xo));   // (24-bit factors: a full-rate multiply)
    }
    asm volatile("" : "+v"(f1.x), "+v"(f1.y), "+v"(f1.z), "+v"(f1.w), "+v"(f2.x), "+v"(f2.y), "+v"(f2.z), "+v"(f2.w));   // pins the loads above this point
    const blur_v4i b1 = {(int)f1.x, (int)f1.y, (int)f1.z, (int)f1.w}, b2 = {(int)f2.x, (int)f2.y, (int)f2.z, (int)f2.w};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < BLUR_TPW; ++k) {
        const int x0 = bt.x0 + BLUR_TW * k;
        if (x0 >= lv.w) break;
        // tile k: input columns x0-4 .. x0+27 = window bytes 12 + 24k ..; this lane's 16 of them
        const uint32_t *q = reinterpret_cast<const uint32_t *>(wn + r * BLUR_IROW + 12 + BLUR_TW * k + 16 * h);
        const uint4 p = make_uint4(q[0], q[1], q[2], q[3]);
        const blur_v4i a1 = {(int)(p.x ^ 0x80808080u), (int)(p.y ^ 0x80808080u), (int)(p.z ^ 0x80808080u), (int)(p.w ^ 0x80808080u)};
        const blur_v16i H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c1, 0, 0, 0);
        blur_v4i ah, al;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // [b0(g), b0(g+1), b1(g), b1(g+1)] of two registers, then the low / high byte planes of four
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)H[4 * d + 1], (uint32_t)H[4 * d], 0x05010400u);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)H[4 * d + 3], (uint32_t)H[4 * d + 2], 0x05010400u);
            al[d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
            ah[d] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
        }
        const blur_v16i VL = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, b2, c2, 0, 0, 0);
        const blur_v16i VH = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, b2, z, 0, 0, 0);
        // lane = output row r; registers 4q .. 4q+3 = columns 8q + 4h + (0..3) of the tile
        if (r < BLUR_TH) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = ((uint32_t)VH[4 * q + e] << 8) + (uint32_t)VL[4 * q + e];   // v + 2^15: the result is byte 2
                    if (WIDE) o[e] = o[e] < 0x00ffffffu ? o[e] : 0x00ffffffu;
                }
                *reinterpret_cast<uint32_t *>(st + r * BLUR_LROW + BLUR_TW * k + 8 * q + 4 * h) =
                    __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u) | __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // rows out as 16-byte pieces, 6 per row, into the strip layout (orbx_internal.h): piece c of a row is one row of strip
    // (PADX + x) / 16 -- a lane's piece and the pieces of the rows above and below it in the same strip are contiguous
    uint8_t *const bdst = blur + (size_t)f * blur_frame_bytes + bt.boff;
#pragma unroll
    for (int s = 0; s < (BLUR_TH * (BLUR_SW / 16) + 63) / 64; ++s) {
        const int i = lane + 64 * s, c = i / BLUR_TH, row = i - c * BLUR_TH;   // rows fastest: four neighbouring lanes = 64 contiguous bytes of a strip
        const int x = bt.x0 + 16 * c, y = bt.y0 + row;
        if (c < BLUR_SW / 16 && y < lv.h && x < lv.w)
            *reinterpret_cast<uint4 *>(bdst + (uint32_t)(__mul24((PADX + x) >> 4, bt.bcol) + (y + EDGE) * 16)) =
                *reinterpret_cast<const uint4 *>(st + mul_u24((uint32_t)row, BLUR_LROW) + 16 * c);
    }
}

// (Measured in round 3 and dropped: k_octree and k_blur as ONE launch -- both follow k_fast_cells / the pyramid and precede k_describe,
// neither needs the other.  66-68 us for the pair instead of 54 + 27 alone, bit-exact, and the pipelined step 3 % SLOWER: the fused
// kernel allocates the octree's 121 VGPRs and 30 KB of LDS for every blur workgroup too.  Even the refactoring that made the two
// bodies callable from one kernel -- arguments through structs, the blur's LDS as a dynamic block -- cost the step 5.5 % with every
// kernel's own time unchanged (0.2737 against 0.2587 ms, three A/B rounds on one box, tools/ab.sh): with static LDS the compiler
// knows k_blur's occupancy limit and schedules for it.  What a kernel pins while it runs is what the other contexts pay for.)

// ----------------------------------------------------- orientation + rBRIEF
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

// Sum over the wave, returned wave-uniform: four row_shr steps leave each 16-lane row's total in its last lane,
// row_bcast:15 / row_bcast:31 carry it across the rows into lane 63 (six DPP adds, no LDS traffic).
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// N independent wave sums, step by step across all of them: a DPP instruction needs its source two VALU slots old, and
// with N chains interleaved those slots hold the other chains' steps instead of s_nop.
template <int N> __device__ __forceinline__ void wave_sum_n(int (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x111, 0xf, 0xf, true);   // row_shr:1
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x112, 0xf, 0xf, true);   // row_shr:2
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x114, 0xf, 0xf, true);   // row_shr:4
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x118, 0xf, 0xf, true);   // row_shr:8
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_readlane(v[i], 63);
}

// 16 keypoints per 256-thread workgroup (4 per wave).  Phase 0: one lane per
// keypoint finds its level and pixel; phase 1: IC_Angle moments (:77-104) on the
// unblurred level, one wave per keypoint at a time; phase 2: one lane per
// keypoint evaluates fastAtan2 and the exact sin/cos ONCE (they are wave-uniform
// values -- evaluating them in every lane of a keypoint's wave would cost 64x the
// instructions); phase 3: computeOrbDescriptor (:108-147) on the blurred level
// and the output record (:845-855 octave/size, :1103-1109 pt *= scale).
// IC_Angle weights for v_dot4c_i32_i8: the 31x31 window is read as 31 rows x 8 dwords starting at (x-15, y-15); dword
// d = row*8 + w holds u = 4w-15 .. 4w-12 of row v = row - 15.  Inside the circular patch (|u| <= umax[|v|],
// ORBextractor.cc:454-469) the U weight of a byte is u and its V weight is v, outside both are 0.  The pixels enter as
// p - 128 (one xor with 0x80808080 makes them signed bytes): the patch is symmetric in u and in v, so the weights of all
// 248 dwords sum to zero and sum(u (p - 128)) = sum(u p) = m10 exactly, likewise m01 -- two accumulating dot products per
// dword and nothing else.
struct MomentWeights { uint32_t u[248], v[248]; };
constexpr MomentWeights make_moment_weights()
{
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    MomentWeights m{};
    for (int d = 0; d < 248; ++d) {
        const int v = d / 8 - 15, av = v < 0 ? -v : v;
        uint32_t wu = 0, wv = 0;
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * (d % 8) + k - 15, au = u < 0 ? -u : u;
            if (au <= umax[av]) { wu |= (uint32_t)(u & 0xff) << (8 * k); wv |= (uint32_t)(v & 0xff) << (8 * k); }
        }
        m.u[d] = wu; m.v[d] = wv;
    }
    return m;
}
__device__ const MomentWeights c_momw = make_moment_weights();

constexpr int DESC_KPB = 16, DESC_PD = 1;
// Per-level constants of k_describe, passed by value (kernel-argument memory: scalar loads, nothing to wait for behind a
// table pointer).  Workgroup `item` of a frame serves chunk item - chunk_base[level] of the level with
// chunk_base[level] <= item < chunk_base[level + 1]: 16 consecutive entries of that level's selected-keypoint slots.
struct DescLevels {
    int off[MAXL], stride[MAXL], sel_base[MAXL], patch[MAXL], chunk_base[MAXL + 1];
    int boff[MAXL], bcol[MAXL];   // the level in the blurred buffer's strip layout
    float scale[MAXL];
};
__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                  size_t frame_bytes, size_t blur_frame_bytes, const DescLevels D, int nlevels,
                                                  const uint32_t *__restrict__ sel_all, int sel_per_frame,
                                                  const int *__restrict__ level_count,
                                                  orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                                  int *__restrict__ counts, int cap, int trig_variant, uint32_t *__restrict__ mirror, int mirror_desc)
{
    // mirror (one frame per call only): the caller's pinned result block {count, 12 bytes | kps[cap] | desc[cap][32]} as dwords, desc at
    // dword mirror_desc -- every record is stored there as well (posted writes), so that no copy kernel runs behind this one
    __shared__ uint32_t s_coff[DESC_KPB];   // patch centre, byte offset inside the frame's pyramid
    __shared__ int s_valid[DESC_KPB], s_out[DESC_KPB], s_m10[DESC_KPB], s_m01[DESC_KPB];
    __shared__ uint32_t s_pk[DESC_KPB];
    __shared__ uint32_t s_boff[DESC_KPB];   // blurred window: byte offset of (strip of its first aligned piece, first aligned row) in the frame's strips
    __shared__ int s_bx[DESC_KPB];          // bits 0-3 (x - 18) mod 16: bit 3 = which half of the strip the first piece is, bits 0-2 = shift inside it; bits 4-5: (y - 18) mod 4
    __shared__ float s_angle[DESC_KPB], s_cos[DESC_KPB], s_sin[DESC_KPB];
    __shared__ uint32_t s_patch[4][40 * 12 + 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ORBX_PH_INIT(1);
    int f, item;
    xcd_frame_item(f, item);
    // the workgroup's level and chunk follow from its index alone, so the keypoint words and the level counts (which
    // only decide validity and the OUTPUT position) are read side by side: one memory latency, not two
    int level = 0;
    for (int l = 1; l < nlevels; ++l) level += item >= D.chunk_base[l];
    const int stride = D.stride[level];
    // the level counts of the frame: wave-uniform addresses, so they come through the scalar cache (as per-lane loads of
    // 16 lanes they were 8 of a workgroup's ~100 texture-addresser instructions)
    int first = 0, mine = 0, total = 0;
    {
        const int *lc = level_count + __builtin_amdgcn_readfirstlane(f) * nlevels;
        for (int l = 0; l < nlevels; ++l) {
            const int c = lc[l];
            first += l < level ? c : 0;
            mine = l == level ? c : mine;
            total += c;
        }
    }
    if (tid < DESC_KPB) {
        const int j = (item - D.chunk_base[level]) * DESC_KPB + tid;   // slot within the level
        const uint32_t pk = sel_all[(size_t)f * sel_per_frame + D.sel_base[level] + j];   // (slots past the level's count: never used)
        if (item == 0 && tid == 0) { counts[f] = total < cap ? total : cap; if (mirror) mirror[0] = (uint32_t)(total < cap ? total : cap); }
        const int x = (int)((pk >> 8) & 0xfffu) + MIN_BORDER, y = (int)(pk >> 20) + MIN_BORDER;
        s_valid[tid] = j < mine && first + j < cap;
        s_out[tid] = first + j;
        s_pk[tid] = pk;
        s_coff[tid] = (uint32_t)(D.off[level] + (y + EDGE) * stride + PADX + x);
        const int X = PADX + x - 18;                 // window origin in the padded row (>= PADX - 2)
        const int Y = y + EDGE - 18;                 // window's first row in the padded level
        s_boff[tid] = (uint32_t)(D.boff[level] + (X >> 4) * D.bcol[level] + (Y & ~3) * 16);
        s_bx[tid] = (X & 15) | ((Y & 3) << 4);
    }
    __syncthreads();
    ORBX_PH(8, tid == 0);    // keypoint lookup
    if (!s_valid[0]) return; // a level's slots fill from 0: nothing in this chunk
    const uint8_t *const fpyr = pyr + (size_t)f * frame_bytes, *const fblur = blur + (size_t)f * blur_frame_bytes;
    // IC_Angle (:77-104): every lane owns 4 dwords of the 31x31 window (fixed per lane, so are
    // their weights) and reads them straight from the unblurred level; two dot4 per dword.
    // No branches on "slot in use": an unused slot reads the window of slot 0 (in use, see above) and its results are
    // dropped, so the loads of all four keypoints of a wave are in flight together.
    uint32_t *patch = s_patch[wv];
    // The 37x37 blurred window of steered BRIEF (|offset| <= 18) is staged in LDS as 40 rows of 12 dwords: the 8-byte ALIGNED pieces
    // that cover columns x-18 .. x+18 (six per row; the window starts (x - 18) mod 8 bytes into the first), from the 4-ALIGNED row at
    // or above y-18.  Aligned pieces never straddle a strip of the blurred buffer, and in a strip consecutive rows are 16 bytes
    // apart: four lanes that take the same piece of four consecutive aligned rows read one 64-byte span -- the unit the texture
    // addresser works in -- so a wave-wide load is 16 of those instead of ~35 (row-major: the four lanes of a group sat in two rows).
    // Every lane owns four fixed pieces (piece d + 64 jj: row d % 40, column d / 40); which strip a column falls into depends on
    // whether the window starts in the first or the second half of its strip -- two offset sets per lane, chosen per keypoint by a
    // wave-uniform select.  The windows are loaded PD keypoints ahead of their use, the first ones together with the moment
    // windows: they depend on the keypoint's position only
    constexpr int PR = 18, PW = 12, PROWS = 40, PD = DESC_PD, NPL = 4; // patch radius; LDS row = 12 dwords; staged rows; prefetch distance; loads per lane
    const int bcol = D.bcol[level];
    uint32_t poffA[NPL], poffB[NPL]; int pslot[NPL];
#pragma unroll
    for (int jj = 0; jj < NPL; ++jj) {
        const int d = lane + 64 * jj, dd = min(d, PROWS * 6 - 1);   // pieces past the window repeat its last one
        const int col = dd / PROWS, row = dd - PROWS * col;
        poffA[jj] = (uint32_t)(row * 16 + __mul24((8 * col) >> 4, bcol) + ((8 * col) & 15));
        poffB[jj] = (uint32_t)(row * 16 + __mul24((8 + 8 * col) >> 4, bcol) + ((8 + 8 * col) & 15) - 8);   // (the base below already carries the 8)
        pslot[jj] = (int)mul_u24((uint32_t)row, PW) + 2 * col;
    }
    uint2 nxt[PD][NPL];
    auto load_patch = [&](int j, uint2 (&r)[NPL]) {   // unused slots read slot 0's window (no branch in front of a load)
        const int kp = wv * (DESC_KPB / 4) + j, kpv = s_valid[kp] ? kp : 0;
        const int bx = __builtin_amdgcn_readfirstlane(s_bx[kpv]);
        const uint8_t *win = fblur + (uint32_t)__builtin_amdgcn_readfirstlane((int)s_boff[kpv]) + (bx & 8);   // first aligned piece of row y-18
#pragma unroll
        for (int jj = 0; jj < NPL; ++jj) __builtin_memcpy(&r[jj], win + ((bx & 8) ? poffB[jj] : poffA[jj]), 8);
    };
    // IC_Angle window, 31 rows x 32 bytes from (x-15, y-15): lane d (+ 64 jj) owns the 8-byte piece d % 4 of row d / 4 (dwords
    // 2 (d % 4) and + 1 of that row): two 8-byte loads per keypoint instead of four dwords; pieces past row 30 re-read row 0 with
    // weights 0
    int wU[4], wV[4];
    uint32_t moff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int d = lane + 64 * jj, row = d >> 2, piece = d & 3;
        const bool in = row < 31;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int w = in ? 8 * row + 2 * piece + k : 0;
            wU[2 * jj + k] = in ? (int)c_momw.u[w] : 0; wV[2 * jj + k] = in ? (int)c_momw.v[w] : 0;
        }
        moff[jj] = (uint32_t)(__mul24(in ? row : 0, stride) + 8 * piece);
    }
    {
        uint32_t px[DESC_KPB / 4][4];
#pragma unroll
        for (int j = 0; j < DESC_KPB / 4; ++j) {
            const int kp = wv * (DESC_KPB / 4) + j, kpv = s_valid[kp] ? kp : 0;
            // wave-uniform value read from LDS: tell the compiler, so the loads take a scalar base + 32-bit lane offset
            const uint8_t *win = fpyr + (uint32_t)__builtin_amdgcn_readfirstlane((int)s_coff[kpv]) - 15 - (ptrdiff_t)15 * stride; // (x-15, y-15)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) __builtin_memcpy(&px[j][2 * jj], win + moff[jj], 8);
        }
#pragma unroll
        for (int j = 0; j < PD; ++j) load_patch(j, nxt[j]);
        int mom[2 * (DESC_KPB / 4)];   // m10, m01 of the wave's four keypoints: eight independent sums
#pragma unroll
        for (int j = 0; j < DESC_KPB / 4; ++j) {
            int m10 = 0, m01 = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int ps = (int)(px[j][jj] ^ 0x80808080u);
                m10 = __builtin_amdgcn_sdot4(ps, wU[jj], m10, false);
                m01 = __builtin_amdgcn_sdot4(ps, wV[jj], m01, false);
            }
            mom[2 * j] = m10; mom[2 * j + 1] = m01;
        }
        wave_sum_n<2 * (DESC_KPB / 4)>(mom);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < DESC_KPB / 4; ++j) { s_m10[wv * (DESC_KPB / 4) + j] = mom[2 * j]; s_m01[wv * (DESC_KPB / 4) + j] = mom[2 * j + 1]; }
        }
    }
    __syncthreads();
    ORBX_PH(9, tid == 0);    // moments
    if (tid < DESC_KPB && s_valid[tid]) {
        const float angle = orbx_fast_atan2((float)s_m01[tid], (float)s_m10[tid]);
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float a, b;
        if (trig_variant == 0) orbx_sincos_glibc_f32(angle * factorPI, &b, &a); // a = cos, b = sin
        else orbx_sincos_f32(angle * factorPI, &b, &a);
        s_angle[tid] = angle; s_cos[tid] = a; s_sin[tid] = b;
    }
    __syncthreads();
    ORBX_PH(10, tid == 0);   // atan2 + sincos
    // steered BRIEF: lane i evaluates tests 4i..4i+3 of its wave's current keypoint, sampling the staged window.
    // Rotation on packed f32 (v_pk_mul_f32 / v_pk_add_f32, the same IEEE operations as the scalar form):
    // (x', y') = (px a - py b, px b + py a) = (px, px) * (a, b) + (-py, py) * (b, a); adding 1.5 * 2^23 rounds both to
    // the nearest-even integer (= cvRound) and leaves 0x4B400000 + value in the bits, which feed the address
    // multiply-add directly (its 24-bit multiplicand is 0x400000 + y').
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 PX[8], PY[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const signed char *pp = c_pattern + 16 * lane + 4 * t;
        PX[2 * t] = f32x2{(float)pp[0], (float)pp[0]}; PY[2 * t] = f32x2{-(float)pp[1], (float)pp[1]};
        PX[2 * t + 1] = f32x2{(float)pp[2], (float)pp[2]}; PY[2 * t + 1] = f32x2{-(float)pp[3], (float)pp[3]};
    }
    unsigned dacc = 0;   // lane 8j + m: dword m of the descriptor of the wave's keypoint j
#pragma unroll
    for (int j = 0; j < DESC_KPB / 4; ++j) {
        const int kp = wv * (DESC_KPB / 4) + j;
        const float a = s_cos[kp], b = s_sin[kp];
#pragma unroll
        for (int jj = 0; jj < NPL; ++jj) *reinterpret_cast<uint2 *>(patch + pslot[jj]) = nxt[j % PD][jj];
        if (j + PD < DESC_KPB / 4) load_patch(j + PD, nxt[j % PD]);
        if (!s_valid[kp]) continue;
        // patch centre, minus what the magic-number bits add: (0x400000 * 40 + 0x4B400000) mod 2^32
        const int bxs = __builtin_amdgcn_readfirstlane(s_bx[kp]);
        const uint8_t *pc = reinterpret_cast<const uint8_t *>(patch) + (PR + (bxs >> 4)) * (PW * 4) + PR + (bxs & 7);
        const f32x2 ab = {a, b}, ba = {b, a}, magic = {12582912.0f, 12582912.0f};
        unsigned nib = 0, tv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x2 w = (PX[t] * ab + PY[t] * ba) + magic;
            const unsigned off = __umul24(__float_as_uint(w.y), PW * 4) + __float_as_uint(w.x) - (0x400000u * (PW * 4) + 0x4B400000u);
            tv[t] = pc[(int)off];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) nib |= (unsigned)(tv[2 * t] < tv[2 * t + 1]) << t;
        // lanes 8j .. 8j+7 hold the eight nibbles of dword j: DPP row_shl (lane i reads lane i+n of its 16-lane row)
        const unsigned byte = nib | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xf, 0xf, true) << 4); // even lanes
        const unsigned w = byte | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x102, 0xf, 0xf, true) << 8) |
                           ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x104, 0xf, 0xf, true) << 16) |
                           ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x106, 0xf, 0xf, true) << 24);
        // lane 8m holds dword m of this keypoint's descriptor: hand it to lane 8j + m, where it waits for the wave's one store
        const unsigned wv_ = (unsigned)__shfl((int)w, 8 * (lane & 7));
        if ((lane >> 3) == j) dacc = wv_;
        ORBX_PH(11 + j, tid == 0);   // BRIEF of the wave's keypoint j (incl. waiting for its patch)
    }
    // Output records of the wave's four keypoints, which sit side by side in the frame's arrays (a level's slots fill from
    // 0, so the valid ones are a prefix): ONE store of 32 lanes x 4 bytes for the descriptors and ONE of 28 lanes x 4 bytes
    // for the keypoint structs (7 dwords each: x, y, size, angle, response, octave, class_id), instead of a store per
    // keypoint and per struct -- the stores go through the texture addresser like the loads, and the kernel is bound by it
    {
        const int kp0 = wv * (DESC_KPB / 4);
        const size_t o0 = (size_t)f * cap + s_out[kp0];
        if (lane < 32 && s_valid[kp0 + (lane >> 3)]) reinterpret_cast<uint32_t *>(desc + o0 * 32)[lane] = dacc;
        if (mirror && lane < 32 && s_valid[kp0 + (lane >> 3)]) mirror[mirror_desc + o0 * 8 + lane] = dacc;
        if (lane < 28) {
            const int jk = lane / 7, fld = lane - 7 * jk, kp = kp0 + jk;
            if (s_valid[kp]) {
                const uint32_t pk = s_pk[kp];
                float x = (float)((int)((pk >> 8) & 0xfffu) + MIN_BORDER), y = (float)((int)(pk >> 20) + MIN_BORDER);
                if (level != 0) { x *= D.scale[level]; y *= D.scale[level]; }   // :1103-1109
                const uint32_t val = fld == 0 ? __float_as_uint(x) : fld == 1 ? __float_as_uint(y) :
                                     fld == 2 ? __float_as_uint((float)D.patch[level]) : fld == 3 ? __float_as_uint(s_angle[kp]) :
                                     fld == 4 ? __float_as_uint((float)(pk & 0xffu)) : fld == 5 ? (uint32_t)level : 0xffffffffu;
                reinterpret_cast<uint32_t *>(kps + o0)[lane] = val;
                if (mirror) mirror[4 + o0 * 7 + lane] = val;
            }
        }
    }
    ORBX_PH(15, tid == 0);
    ORBX_PH_END(tid == 0);
}

} // namespace

// ======================================================================= host

namespace {

// slots of a level in the selected-keypoint array: DistributeOctTree's worst case (see orbx_create) + 1
inline int level_slots(const LevelInfo &lv) { return std::max(lv.N + 4, 4 * lv.nIni); }

void drop_graph(orbx_extractor *ex)
{
    if (ex->g_exec) (void)hipGraphExecDestroy(ex->g_exec);
    if (ex->g_graph) (void)hipGraphDestroy(ex->g_graph);
    ex->g_exec = nullptr; ex->g_graph = nullptr; ex->g_w = ex->g_h = ex->g_stride = 0;
}

void free_workspace(orbx_extractor *ex)
{
    drop_graph(ex);
    void *ptrs[] = {ex->d_chain, ex->d_chain_tabs, ex->d_chain_span, ex->d_pyr, ex->d_blur, ex->d_lv, ex->d_cells, ex->d_tiles, ex->d_blur_frag, ex->d_xt, ex->d_yt, ex->d_cell_count,
                    ex->d_level_count, ex->d_level_ncand, ex->d_counts, ex->d_cands, ex->d_kpos, ex->d_sel,
                    ex->d_knode, ex->d_kq, ex->d_desc, ex->d_kps};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    ex->d_chain = nullptr; ex->d_chain_tabs = nullptr; ex->d_chain_span = nullptr; ex->d_pyr = ex->d_blur = nullptr; ex->d_lv = nullptr; ex->d_cells = nullptr; ex->d_tiles = nullptr; ex->d_blur_frag = nullptr;
    ex->d_xt = nullptr; ex->d_yt = nullptr; ex->d_cell_count = ex->d_level_count = ex->d_level_ncand = ex->d_counts = nullptr;
    ex->d_cands = ex->d_kpos = ex->d_sel = nullptr; ex->d_knode = nullptr; ex->d_kq = ex->d_desc = nullptr; ex->d_kps = nullptr;
    ex->width = ex->height = ex->batch = 0;
}

inline int cv_round_f(float v) { return (int)lrintf(v); }

// OpenCV resize coefficient tables for one axis.
void resize_axis(int dn, int sn, std::vector<int> &ofs, std::vector<int> &c0, std::vector<int> &c1)
{
    const double inv_scale = (double)dn / sn, scale = 1. / inv_scale;
    ofs.resize(dn); c0.resize(dn); c1.resize(dn);
    for (int d = 0; d < dn; ++d) {
        float fv = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(fv);
        fv -= s;
        ofs[d] = s;
        const int a0 = cv_round_f((1.f - fv) * 2048.f), a1 = cv_round_f(fv * 2048.f);
        c0[d] = a0 > 32767 ? 32767 : a0;
        c1[d] = a1 > 32767 ? 32767 : a1;
    }
}

// One frame's results into the caller's pinned block: {count, 0, 0, 0}, the keypoint records, the descriptors -- written by the
// queue that produced them (posted writes over PCIe), 16 bytes per lane.
__global__ __launch_bounds__(256) void k_result_out(const int *__restrict__ counts, const uint4 *__restrict__ kps, const uint4 *__restrict__ desc,
                                                    int kps16, int desc16, uint4 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) out[0] = make_uint4((unsigned)counts[0], 0u, 0u, 0u);
    if (i < kps16) out[1 + i] = kps[i];
    else if (i < kps16 + desc16) out[1 + i] = desc[i - kps16];
}

std::mutex g_reg_mu;
std::vector<orbx_extractor *> g_reg;   // live handles (orbx_create .. orbx_destroy)

} // namespace

namespace orbx_detail {

orbx_extractor *order_after_producer(const void *ptr, hipStream_t st)
{
    const uint8_t *q = (const uint8_t *)ptr;
    orbx_extractor *hit = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (orbx_extractor *ex : g_reg) {
            if (!ex->d_desc || !ex->batch) continue;
            const size_t B = (size_t)ex->batch, cap = (size_t)ex->kcap;
            const uint8_t *d = ex->d_desc, *k = (const uint8_t *)ex->d_kps, *c = (const uint8_t *)ex->d_counts;
            if ((q >= d && q < d + B * cap * 32) || (q >= k && q < k + B * cap * sizeof(orbx_keypoint)) || (q >= c && q < c + B * sizeof(int))) {
                hit = ex;
                break;
            }
        }
    }
    if (!hit || !hit->last_batch || hit->last_stream == st) return nullptr;
    if (!hit->order_ev && hipEventCreateWithFlags(&hit->order_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    // everything queued on the producer's stream so far -- its last batch was queued by an earlier call -- precedes `st`'s next work
    if (hipEventRecord(hit->order_ev, hit->last_stream) != hipSuccess || hipStreamWaitEvent(st, hit->order_ev, 0) != hipSuccess) return nullptr;
    return hit;
}

void reader_done(orbx_extractor *ex, hipStream_t st)
{
    if (!ex) return;
    if (!ex->reader_ev && hipEventCreateWithFlags(&ex->reader_ev, hipEventDisableTiming) != hipSuccess) return;
    if (hipEventRecord(ex->reader_ev, st) != hipSuccess) return;
    ex->reader_stream = st;
    ex->reader_pending = true;
}

} // namespace orbx_detail

extern "C" {

const char *orbx_last_error(void) { return orbx::last_error().c_str(); }
int orbx_abi_version(void) { return 135; } // 135: + orbm_profile_read_bow (addition only); 134: + fem_cg_one_launch_stats (addition only); 133: + fem_plan_single_cg (addition only); 132: + fem_cg_preconditioner, fem_cg_coarse_matrix (additions only); 131: + orbm_frame_search_by_bow, orbm_frame_search_for_triangulation; ORBM_ALLPAIRS_MFMA now names the second matrix-core kernel (additions only); 130: + orbm_frame_*, the whole-function searches, orbx_params.trig_variant (a struct field: rebuild callers); 120: + fem_plan, fem_plan_selfcheck, orbm_sorted_frame, orbx_stereo_download_batch, orbm_debug_* (additions only); 110: + orbm_project_points, fem_create_batch, fem_batch_offsets

int orbx_create(const orbx_params *prm, orbx_extractor **out)
{
    if (!prm || !out) ORBX_FAIL(ORBX_ERR_ARG, "null argument");
    if (prm->nlevels < 1 || prm->nlevels > MAXL || prm->nfeatures < 0 || !(prm->scale_factor > 1.0f) || prm->blur_variant < 0 ||
        prm->blur_variant > 1 || prm->trig_variant < 0 || prm->trig_variant > 1)
        ORBX_FAIL(ORBX_ERR_ARG, "bad extractor parameters");
    orbx_extractor *ex = new orbx_extractor();
    ex->prm = *prm;
    const int nl = ex->nlevels = prm->nlevels;
    // ORBextractor::ORBextractor, ORBextractor.cc:415-446
    ex->scale[0] = 1.0f; ex->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        ex->scale[i] = ex->scale[i - 1] * prm->scale_factor;
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        ex->inv_scale[i] = 1.0f / ex->scale[i];
        ex->inv_sigma2[i] = 1.0f / ex->sigma2[i];
    }
    const float factor = 1.0f / prm->scale_factor;
    float nDesired = prm->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        ex->nfeat[l] = cv_round_f(nDesired);
        sum += ex->nfeat[l];
        nDesired *= factor;
    }
    ex->nfeat[nl - 1] = prm->nfeatures - sum > 0 ? prm->nfeatures - sum : 0;
    for (int l = 0; l < nl; l++)
        if (ex->nfeat[l] > OCT_MAXN) {
            delete ex;
            ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "per-level feature quota above 2047 is not supported");
        }
    if (prm->blur_variant == 1) { ex->taps[0] = 18; ex->taps[1] = 34; ex->taps[2] = 49; ex->taps[3] = 55; }
    else { ex->taps[0] = 18; ex->taps[1] = 34; ex->taps[2] = 48; ex->taps[3] = 56; }
    // DistributeOctTree returns at most N + 3 keypoints for a quota N >= 1 (the last split may overshoot by three) -- and up to
    // four per INITIAL node whatever N is: every initial node is split before the first "enough nodes" test
    // (ORBextractor.cc:575-669).  The number of initial nodes is round(width / height) of the level (:543), known once the
    // frame size is (orbx_reserve raises the capacity if 4 nIni exceeds N + 3 on some level: tiny quotas on wide frames)
    ex->kcap_params = prm->nfeatures + 3 * nl;
    for (int l = 0; l < nl; l++) ex->kcap_params += ex->nfeat[l] == 0 ? 1 : 0;
    ex->kcap = ex->kcap_params;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_reg.push_back(ex);
    }
    *out = ex;
    return ORBX_OK;
}

int orbx_destroy(orbx_extractor *ex)
{
    if (!ex) return ORBX_OK;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_reg.erase(std::remove(g_reg.begin(), g_reg.end(), ex), g_reg.end());
    }
    if (ex->reader_pending) (void)hipEventSynchronize(ex->reader_ev);   // a consumer on another stream may still read the results
    if (ex->order_ev) (void)hipEventDestroy(ex->order_ev);
    if (ex->reader_ev) (void)hipEventDestroy(ex->reader_ev);
    free_workspace(ex);
    if (ex->d_in) (void)hipFree(ex->d_in);
    void *stp[] = {ex->d_st_key, ex->d_st_rk, ex->d_st_rowoff, ex->d_st_items, ex->d_uright, ex->d_depth, ex->d_st_scale, ex->d_st_sad, ex->d_st_nvalid};
    for (void *q : stp)
        if (q) (void)hipFree(q);
    if (ex->h_pin) (void)hipHostFree(ex->h_pin);
    if (ex->h_st_pin) (void)hipHostFree(ex->h_st_pin);
    if (ex->stream) (void)hipStreamDestroy(ex->stream);
    delete ex;
    return ORBX_OK;
}

int orbx_get_levels(const orbx_extractor *ex) { return ex ? ex->nlevels : ORBX_ERR_ARG; }
#define ORBX_GETTER(name, field, type)                                   \
    int name(const orbx_extractor *ex, type *out)                        \
    {                                                                    \
        if (!ex || !out) ORBX_FAIL(ORBX_ERR_ARG, "null argument");       \
        for (int i = 0; i < ex->nlevels; i++) out[i] = ex->field[i];     \
        return ORBX_OK;                                                  \
    }
ORBX_GETTER(orbx_get_scale_factors, scale, float)
ORBX_GETTER(orbx_get_inv_scale_factors, inv_scale, float)
ORBX_GETTER(orbx_get_level_sigma2, sigma2, float)
ORBX_GETTER(orbx_get_inv_level_sigma2, inv_sigma2, float)
ORBX_GETTER(orbx_get_features_per_level, nfeat, int32_t)
int orbx_keypoint_capacity(const orbx_extractor *ex) { return ex ? ex->kcap : ORBX_ERR_ARG; }

int orbx_level_size(const orbx_extractor *ex, int level, int *w, int *h)
{
    if (!ex || level < 0 || level >= ex->nlevels || !ex->width) ORBX_FAIL(ORBX_ERR_ARG, "bad level / nothing reserved");
    if (w) *w = ex->lv[level].w;
    if (h) *h = ex->lv[level].h;
    return ORBX_OK;
}

// Everything orbx_reserve decides about a frame size on the HOST: level geometry, the FAST cell grid, the octree's initial nodes,
// slot / node / result capacities, resize coefficient tables, blur tiles, LDS budgets.  No device call (orbx_plan runs it
// without a GPU: the CPU tests and the host sanitizer pass check its capacities against the oracle's literal algorithm).
static int plan_frame(orbx_extractor *ex, int width, int height, std::vector<int2> &xt, std::vector<int4> &yt)
{
    const int nl = ex->nlevels;
    xt.clear(); yt.clear();
    ex->cells.clear(); ex->tiles.clear();
    size_t off = 0, cand_off = 0, key_off = 0;
    int sel_off = 0, maxcw = 0, maxch = 0, kneed = 0;
    size_t boff_run = 0;
    ex->maxcells = 0;
    for (int l = 0; l < nl; l++) {
        LevelInfo &lv = ex->lv[l];
        memset(&lv, 0, sizeof(lv));
        // ComputePyramid, ORBextractor.cc:1119-1121
        lv.w = cv_round_f((float)width * ex->inv_scale[l]);
        lv.h = cv_round_f((float)height * ex->inv_scale[l]);
        if (lv.w > 4000 || lv.h > 4000) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "image larger than 4000 px");
        lv.stride = (PADX + lv.w + EDGE + 63) & ~63;
        lv.off = (int)off;
        off += (size_t)lv.stride * (lv.h + 2 * EDGE);
        off = (off + 255) & ~(size_t)255;
        // the same level in the blurred buffer's strip layout: stride / 16 strips of ((rows + 7) & ~7) x 16 bytes
        ex->bcol[l] = ((lv.h + 2 * EDGE + 7) & ~7) * 16;
        ex->boff[l] = (int)boff_run;
        boff_run += (size_t)(lv.stride / 16) * ex->bcol[l];
        lv.scale = ex->scale[l];
        lv.patch = (int)(31 * ex->scale[l]); // :845
        lv.N = ex->nfeat[l];
        // ComputeKeyPointsOctTree grid, :773-787
        const int maxBorderX = lv.w - EDGE + 3, maxBorderY = lv.h - EDGE + 3;
        const float fw = (float)(maxBorderX - MIN_BORDER), fh = (float)(maxBorderY - MIN_BORDER);
        const int nCols = (int)(fw / 30.f), nRows = (int)(fh / 30.f);
        if (nCols < 1 || nRows < 1) ORBX_FAIL(ORBX_ERR_ARG, "image too small for the 30-px FAST cell grid at some level");
        const int wCell = (int)ceilf(fw / nCols), hCell = (int)ceilf(fh / nRows);
        lv.W = maxBorderX - MIN_BORDER;
        lv.H = maxBorderY - MIN_BORDER;
        lv.nIni = (int)roundf((float)lv.W / lv.H); // :543
        if (lv.nIni < 1) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "aspect ratio below 0.5 (reference divides by zero)");
        lv.hX = (float)lv.W / lv.nIni;
        lv.cell_base = (int)ex->cells.size();
        lv.key_base = (int)key_off;
        for (int i = 0; i < nRows; i++) { // :789-806
            const float iniY = (float)(MIN_BORDER + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(MIN_BORDER + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                CellInfo c;
                memset(&c, 0, sizeof(c));
                c.level = (short)l;
                c.x0 = (short)(int)iniX; c.y0 = (short)(int)iniY;
                c.cw = (short)((int)maxX - (int)iniX); c.ch = (short)((int)maxY - (int)iniY);
                c.dx = (short)(j * wCell); c.dy = (short)(i * hCell);
                const int zw = c.cw - 6, zh = c.ch - 6;
                c.cap = (zw > 0 && zh > 0) ? ((zw + 1) / 2) * ((zh + 1) / 2) : 0; // 3x3 strict NMS bound
                if (zw > 63 || zh > 63) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "FAST cell larger than 63 px");
                c.ngx = (short)(zw > 0 ? (zw + 3) >> 2 : 1);
                c.qstep = (short)(64 / c.ngx); c.rstep = (short)(64 - c.qstep * c.ngx);
                c.inv_ngx = (65536 + c.ngx - 1) / c.ngx;
                c.stride = lv.stride;
                c.img_off = lv.off + (c.y0 + EDGE) * lv.stride + PADX + (c.x0 - 5);
                c.cand_off = (int)cand_off;
                cand_off += c.cap;
                key_off += c.cap;
                if (c.cw > maxcw) maxcw = c.cw;
                if (c.ch > maxch) maxch = c.ch;
                ex->cells.push_back(c);
            }
        }
        lv.ncells = (int)ex->cells.size() - lv.cell_base;
        if (lv.ncells > ex->maxcells) ex->maxcells = lv.ncells;
        lv.sel_base = sel_off;
        sel_off += level_slots(lv);
        kneed += std::max(lv.N + 3, 4 * lv.nIni);
        // blur tiles
        for (int y0 = 0; y0 < lv.h; y0 += BLUR_TH)
            for (int x0 = 0; x0 < lv.w; x0 += BLUR_SW) {
                BlurTile t; t.x0 = (short)x0; t.y0 = (short)y0; t.w = (short)lv.w; t.h = (short)lv.h; t.off = lv.off; t.stride = lv.stride;
                t.boff = ex->boff[l]; t.bcol = ex->bcol[l]; t.pad[0] = t.pad[1] = 0;
                ex->tiles.push_back(t);
            }
        // resize tables from level l-1
        if (xt.size() & 1) xt.push_back(make_int2(0, 0));   // every level's table 16-byte aligned (int4 reads in k_pyr_resize)
        lv.xtab = (int)xt.size(); lv.ytab = (int)yt.size();
        if (l > 0) {
            const LevelInfo &sv = ex->lv[l - 1];
            std::vector<int> o, c0, c1;
            resize_axis(lv.w, sv.w, o, c0, c1);
            for (int d = 0; d < lv.w; d++) {
                int s = o[d], a0 = c0[d], a1 = c1[d];
                if (s < 0) { s = 0; a0 = 2048; a1 = 0; }              // fx = 0, sx = 0
                if (s >= sv.w - 1) { s = sv.w - 1; a0 = 2048; a1 = 0; } // fx = 0, sx = w-1
                xt.push_back(make_int2(s, a0 | (a1 << 16)));
            }
            // interior dword groups read their source columns as one 8-byte window: sx non-decreasing, span <= 8 bytes
            bool window_ok = true;
            for (int d = 0; d + 3 < lv.w; d += 4) {
                const int2 *g = &xt[lv.xtab + d];
                for (int k = 1; k < 4; ++k) window_ok = window_ok && g[k].x >= g[k - 1].x;
                window_ok = window_ok && g[3].x + 1 - g[0].x <= 7;
            }
            ex->resize_nxi[l] = window_ok ? 1 : 0;   // the 8-byte-window fast path may serve this level's interior groups
            {   // ... and the tail tiles' groups (border groups: reflected columns; row padding: clamped ones) if every dword group of
                // the padded row keeps its four source columns within 7 bytes
                bool ok = true;
                for (int xw = (PADX - EDGE) >> 2; xw <= (PADX + lv.w + EDGE - 1) >> 2; ++xw) {
                    int lo = 1 << 30, hi = -1;
                    for (int b = 0; b < 4; ++b) {
                        int px = xw * 4 - PADX + b;
                        px = px < -EDGE ? -EDGE : px > lv.w + EDGE - 1 ? lv.w + EDGE - 1 : px;
                        if (lv.w == 1) px = 0; else while (px < 0 || px >= lv.w) px = px < 0 ? -px : 2 * (lv.w - 1) - px;   // reflect101
                        const int sx = xt[lv.xtab + px].x;
                        lo = sx < lo ? sx : lo; hi = sx > hi ? sx : hi;
                    }
                    ok = ok && hi - lo <= 6;
                }
                ex->resize_tailwin[l] = ok ? 1 : 0;
            }
            resize_axis(lv.h, sv.h, o, c0, c1);
            for (int d = 0; d < lv.h; d++) {
                const int s = o[d];
                const int s0 = s < 0 ? 0 : (s < sv.h ? s : sv.h - 1);
                const int s1 = s + 1 < 0 ? 0 : (s + 1 < sv.h ? s + 1 : sv.h - 1);
                yt.push_back(make_int4(s0, s1, c0[d], c1[d]));
            }
        }
    }
    ex->frame_bytes = off;
    ex->blur_frame_bytes = (boff_run + 255) & ~(size_t)255;
    ex->cands_per_frame = cand_off;
    ex->keys_per_frame = key_off;
    ex->cells_per_frame = (int)ex->cells.size();
    ex->sel_per_frame = sel_off;
    ex->kcap = std::max(ex->kcap_params, kneed);
    int maxN = 0;
    for (int l = 0; l < nl; l++) {
        if (ex->lv[l].N > maxN) maxN = ex->lv[l].N;
        if (4 * ex->lv[l].nIni - 4 > maxN) maxN = 4 * ex->lv[l].nIni - 4;   // the first pass leaves up to 4 nIni nodes
    }
    ex->NC = (maxN + 8 + 3) & ~3; // multiple of 4: the node arrays stay 16-byte aligned (int4 reads in k_octree)
    // tile row = 5 spare bytes + the cell + the packed pre-test's right-hand dword; zone <= 63 (6-bit queue
    // coordinates)
    if (maxcw + 12 > 80 || maxch > 69) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "FAST cell larger than the LDS tile");
    // three instantiations by the widest cell: tile stride 52 / score-map stride 40 for cells up to 44 px (zone <= 38: its
    // ten 4-pixel groups end at tile column 51 and its score row, with the two border columns, at 39) -- the 36-43 px
    // cells of every usual image size; 64 / 64; 80 / 80.  LDS per wave sets how many waves a CU holds.
    ex->TS = maxcw <= 44 ? 52 : (maxcw + 12 <= 64) ? 64 : 80;
    ex->SS = maxcw <= 44 ? 40 : ex->TS;
    ex->tile_bytes = (ex->TS * maxch + 15) & ~15;
    const int gq_bytes = 4 * (((maxcw - 6 + 3) >> 2) * (maxch - 6));                               // group queue
    ex->sc_bytes = (std::max(ex->SS * (maxch - 6 + 2), gq_bytes) + 15) & ~15;                      // score map, sharing the group queue's region
    ex->queue_bytes = (2 * (maxcw - 6) * (maxch - 6) + 2 * 64 + 15) & ~15;                         // pixel queue + the pre-test's dump slots
    ex->fast_lds = ex->tile_bytes + ex->sc_bytes + ex->queue_bytes;
    // Keys of a level live in LDS up to this many, else in HBM.  Small on purpose: while k_octree runs (60-80 us of serial
    // rounds) its workgroups pin their LDS on every CU and keep the LDS-hungry kernels of the other pipeline contexts
    // (FAST, blur) off it.  With room for 4096 keys (60 KB per workgroup) the kernel alone is 20 % faster and the
    // 64-frame step 4-5 % slower than with 1536 (39 KB; levels 0-2 of a 640 x 480 frame then keep their keys in HBM).
    // (Since the keys of levels up to 2048 candidates live in registers, the LDS slots only serve larger levels' arrays -- but giving
    // them up (kcap = 0, 28 KB per workgroup) made the step 1 % SLOWER: measured 0.266 against 0.263 ms; kept.)
    ex->oct_kcap = 1536;
    {   // k_octree ranks nodes by size << kshift | creation seq in one dword where that fits (seq < NC <= 2^kshift, kshift = 11 up to
        // a per-level quota of 2,037 features, else 12; a level's candidate slots -- a good quarter of its FAST zone: 3 x 3 strict
        // NMS -- below 2^(32 - kshift)), and by the plain pair otherwise (kshift = 0: frames beyond ~8 M pixels, e.g. 3840 x 2160).
        // What remains a limit: 24 bits for a candidate's index within its level, int offsets per frame.
        size_t worst = 0;
        for (int l = 0; l < nl; l++) {
            const size_t next = l + 1 < nl ? (size_t)ex->lv[l + 1].key_base : ex->keys_per_frame;
            worst = std::max(worst, next - (size_t)ex->lv[l].key_base);
        }
        const int ks = ex->NC <= 2048 ? 11 : 12;
        ex->oct_kshift = worst < ((size_t)1 << (32 - ks)) ? ks : 0;
        if (worst >= ((size_t)1 << 24) || ex->keys_per_frame >= ((size_t)1 << 31) || ex->cands_per_frame >= ((size_t)1 << 31))
            ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "more than 2^24 FAST candidate slots in one pyramid level");
    }
    // ---- the pyramid chain in one launch (k_pyr_chain): tiles and their rectangles per level
    ex->chain.clear(); ex->chain_tiles = 0; ex->chain_ldsA = ex->chain_ldsB = 0;
    {
        bool ok = nl >= 2 && nl <= MAXL;
        for (int l = 1; l < nl && ok; ++l) ok = ex->resize_nxi[l] != 0 && ex->lv[l].w >= 8 && ex->lv[l].h >= 8;   // the 8-byte source window must hold at every level
        const auto refl = [](int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p; return p; };
        if (ok) {
            const LevelInfo &l1 = ex->lv[1];
            // tiles of 48 x 36 px at level 1 (15 x 11 of them on 640 x 480): one frame's launch 22.3 us; 96 x 72: 26.8, 32 x 36: 22.6 -- a
            // workgroup's seven dependent levels are the floor, not the tile's area (ORBX_CHAIN_TW / _TH: development knobs)
            const int ctw = getenv("ORBX_CHAIN_TW") ? std::max(16, atoi(getenv("ORBX_CHAIN_TW"))) : 48, cth = getenv("ORBX_CHAIN_TH") ? std::max(16, atoi(getenv("ORBX_CHAIN_TH"))) : 36;
            const int nx = std::max(1, (l1.w + 2 * EDGE + ctw - 1) / ctw), ny = std::max(1, (l1.h + 2 * EDGE + cth - 1) / cth);
            int maxA = 0, maxB = 0;
            for (int j = 0; j < ny && ok; ++j)
                for (int i = 0; i < nx && ok; ++i) {
                    ChainTile T;
                    memset(&T, 0, sizeof(T));
                    int ax0 = 0, ax1 = -1, ay0 = 0, ay1 = -1;        // what the deeper level reads of this one (inner coordinates, inclusive; empty: ax1 < ax0)
                    for (int l = nl - 1; l >= 0; --l) {
                        const LevelInfo &lv = ex->lv[l];
                        int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
                        ChainRect &rc = T.r[l];
                        if (l >= 1) {
                            // ownership follows the INNER image: tile i owns the columns [w i / nx, w (i + 1) / nx) (cut at multiples of 4, so
                            // that the cuts are dword boundaries of the padded row), the first and last tile the 19-px border beside them --
                            // the tiles of all levels then cover the same part of the scene and a level's rectangle is little more than what
                            // the tile owns of it
                            const int xlo = (PADX - EDGE) >> 2, xhi = (PADX + lv.w + EDGE - 1) >> 2, R = lv.h + 2 * EDGE;
                            const auto bx = [&](int k) { return (int)((long long)k * lv.w / nx) & ~3; };
                            const auto by = [&](int k) { return (int)((long long)k * lv.h / ny); };
                            rc.oxw0 = (short)(i == 0 ? xlo : (PADX + bx(i)) >> 2); rc.oxw1 = (short)(i == nx - 1 ? xhi + 1 : (PADX + bx(i + 1)) >> 2);
                            rc.or0 = (short)(j == 0 ? 0 : EDGE + by(j)); rc.or1 = (short)(j == ny - 1 ? R : EDGE + by(j + 1));
                            for (int px = rc.oxw0 * 4 - PADX; px < rc.oxw1 * 4 - PADX; ++px) {
                                if (px < -EDGE || px >= lv.w + EDGE) continue;
                                const int x = refl(px, lv.w);
                                x0 = std::min(x0, x); x1 = std::max(x1, x);
                            }
                            for (int row = rc.or0; row < rc.or1; ++row) {
                                const int y = refl(row - EDGE, lv.h);
                                y0 = std::min(y0, y); y1 = std::max(y1, y);
                            }
                            if (x1 < x0 || y1 < y0) { x0 = y0 = 1 << 30; x1 = y1 = -1; rc.oxw1 = rc.oxw0; rc.or1 = rc.or0; }   // owns nothing here
                        }
                        if (ax1 >= ax0) { x0 = std::min(x0, ax0); x1 = std::max(x1, ax1); y0 = std::min(y0, ay0); y1 = std::max(y1, ay1); }
                        if (x1 < x0) { x0 = x1 = 0; y0 = y1 = 0; }      // nothing owned, nothing needed below: a 1 x 1 rectangle keeps the kernel uniform
                        x0 &= ~3;
                        rc.nx0 = (short)x0; rc.ny0 = (short)y0; rc.nw = (short)(x1 - x0 + 1); rc.nh = (short)(y1 - y0 + 1);
                        if (rc.nw > 1024 || rc.nh > 4096) ok = false;
                        const int bytes = (((rc.nw + 15) & ~15) + 16) * rc.nh + 16;
                        if (l & 1) maxB = std::max(maxB, bytes); else maxA = std::max(maxA, bytes);
                        if (l >= 1) {      // what computing this rectangle reads of level l - 1
                            const int2 *xtab = &xt[lv.xtab];
                            const int4 *ytab = &yt[lv.ytab];
                            const int wsrc = ex->lv[l - 1].w;
                            ax0 = xtab[x0].x; ax1 = std::min(xtab[std::min(x1, lv.w - 1)].x + 1, wsrc - 1);
                            ay0 = std::min(ytab[y0].x, ytab[y0].y); ay1 = std::max(ytab[y1].x, ytab[y1].y);
                        }
                    }
                    ex->chain.push_back(T);
                }
            maxA = (maxA + 15) & ~15; maxB = (maxB + 15) & ~15;
            // the coefficient tables of every tile as one block of 16-byte units: per level the rows, then per level the columns
            int maxT = 0;
            ex->chain_tabs.clear(); ex->chain_span.clear();
            for (const ChainTile &T : ex->chain) {
                const size_t first = ex->chain_tabs.size();
                std::vector<int2> rows;
                for (int l = 1; l < nl; ++l)
                    for (int r = 0; r < T.r[l].nh; ++r) {
                        const int4 y = yt[ex->lv[l].ytab + T.r[l].ny0 + r];
                        const int s0 = y.x - T.r[l - 1].ny0, s1 = y.y - T.r[l - 1].ny0;
                        if (s0 < 0 || s1 < s0 || s1 > 65535 || y.z < 0 || y.z > 65535 || y.w < 0 || y.w > 65535) ok = false;
                        rows.push_back(make_int2(s0 | (s1 << 16), y.z | (y.w << 16)));
                    }
                if (rows.size() & 1) rows.push_back(make_int2(0, 0));
                for (size_t k = 0; k < rows.size(); k += 2)
                    ex->chain_tabs.push_back(make_uint4((unsigned)rows[k].x, (unsigned)rows[k].y, (unsigned)rows[k + 1].x, (unsigned)rows[k + 1].y));
                std::vector<int2> cols;
                for (int l = 1; l < nl; ++l)
                    for (int c = 0; c < ((T.r[l].nw + 3) & ~3); ++c) {
                        int2 x = xt[ex->lv[l].xtab + std::min(T.r[l].nx0 + c, ex->lv[l].w - 1)];     // (columns past the level: junk nobody reads)
                        x.x -= T.r[l - 1].nx0;
                        cols.push_back(x);
                    }
                for (size_t k = 0; k < cols.size(); k += 2)      // (rounded-up widths are multiples of 4: an even count)
                    ex->chain_tabs.push_back(make_uint4((unsigned)cols[k].x, (unsigned)cols[k].y, (unsigned)cols[k + 1].x, (unsigned)cols[k + 1].y));
                ex->chain_span.push_back(make_int2((int)first, (int)(ex->chain_tabs.size() - first)));
                maxT = std::max(maxT, (int)(ex->chain_tabs.size() - first) * 16);
            }
            if (ok && maxA + maxB + maxT <= 64 * 1024) { ex->chain_tiles = nx * ny; ex->chain_ldsA = maxA; ex->chain_ldsB = maxB; ex->chain_ldsT = maxT; }
            else ex->chain.clear();
        }
    }
    ex->oct_lds = (int)sizeof(int) * (2 * (OCT_TW / 64) + 2 * ((ex->maxcells + 1 + 3) & ~3) + 16 * ex->NC + ex->oct_kcap + (ex->oct_kcap + 1) / 2 + (ex->oct_kcap + 3) / 4) + 64;
    if (ex->oct_lds > 160 * 1024) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "octree node pool exceeds LDS");

    return ORBX_OK;
}

int orbx_reserve(orbx_extractor *ex, int width, int height, int batch)
{
    if (!ex || width <= 0 || height <= 0 || batch <= 0) ORBX_FAIL(ORBX_ERR_ARG, "bad reserve arguments");
    ORBX_NEED_DEVICE();
    if (ex->width == width && ex->height == height && ex->batch >= batch) return ORBX_OK;
    if (!ex->stream) ORBX_HIP(hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    free_workspace(ex);
    std::vector<int2> xt;
    std::vector<int4> yt;
    {
        const int rc = plan_frame(ex, width, height, xt, yt);
        if (rc != ORBX_OK) return rc;
    }

    // every fill and upload below runs on the handle's own stream and is waited for there: a blocking copy on the legacy stream
    // would wait for every other thread's work and fails outright while any thread captures a graph (hipErrorStreamCaptureImplicit)
    const size_t B = (size_t)batch;
    ORBX_HIP(hipMalloc(&ex->d_pyr, ex->frame_bytes * B));
    ORBX_HIP(hipMemsetAsync(ex->d_pyr, 0, ex->frame_bytes * B, ex->stream)); // the row padding outside the 19-px border is never written by k_pyr_resize
    ORBX_HIP(hipMalloc(&ex->d_blur, ex->blur_frame_bytes * B));
    ORBX_HIP(hipMemsetAsync(ex->d_blur, 0, ex->blur_frame_bytes * B, ex->stream));
    ORBX_HIP(hipMalloc(&ex->d_lv, sizeof(LevelInfo) * MAXL));
    {
        int dev = 0, ncu = 0;
        ORBX_HIP(hipGetDevice(&dev));
        ORBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        if (ncu > 0) ex->resident_waves = 32 * ncu;
    }
    ORBX_HIP(hipMalloc(&ex->d_cells, sizeof(CellInfo) * ex->cells.size()));
    ORBX_HIP(hipMalloc(&ex->d_tiles, sizeof(BlurTile) * ex->tiles.size()));
    if (ex->chain_tiles) {
        ORBX_HIP(hipMalloc(&ex->d_chain, sizeof(ChainTile) * ex->chain.size()));
        ORBX_HIP(hipMemcpyAsync(ex->d_chain, ex->chain.data(), sizeof(ChainTile) * ex->chain.size(), hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMalloc(&ex->d_chain_tabs, sizeof(uint4) * ex->chain_tabs.size()));
        ORBX_HIP(hipMemcpyAsync(ex->d_chain_tabs, ex->chain_tabs.data(), sizeof(uint4) * ex->chain_tabs.size(), hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMalloc(&ex->d_chain_span, sizeof(int2) * ex->chain_span.size()));
        ORBX_HIP(hipMemcpyAsync(ex->d_chain_span, ex->chain_span.data(), sizeof(int2) * ex->chain_span.size(), hipMemcpyHostToDevice, ex->stream));
        if (ex->chain_ldsA + ex->chain_ldsB + ex->chain_ldsT > 48 * 1024)
            ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pyr_chain), hipFuncAttributeMaxDynamicSharedMemorySize, ex->chain_ldsA + ex->chain_ldsB + ex->chain_ldsT));
    }

    ORBX_HIP(hipMalloc(&ex->d_xt, sizeof(int2) * (xt.size() + 1)));
    ORBX_HIP(hipMalloc(&ex->d_yt, sizeof(int4) * (yt.size() + 1)));
    ORBX_HIP(hipMalloc(&ex->d_cell_count, sizeof(int) * ex->cells_per_frame * B));
    ORBX_HIP(hipMalloc(&ex->d_level_count, sizeof(int) * MAXL * B));
    ORBX_HIP(hipMalloc(&ex->d_level_ncand, sizeof(int) * MAXL * B));
    ORBX_HIP(hipMalloc(&ex->d_counts, sizeof(int) * B));
    ORBX_HIP(hipMalloc(&ex->d_cands, sizeof(uint32_t) * ex->cands_per_frame * B));
    ORBX_HIP(hipMalloc(&ex->d_kpos, sizeof(uint32_t) * ex->keys_per_frame * B));
    ORBX_HIP(hipMalloc(&ex->d_knode, sizeof(unsigned short) * ex->keys_per_frame * B));
    ORBX_HIP(hipMalloc(&ex->d_kq, ex->keys_per_frame * B));
    ORBX_HIP(hipMalloc(&ex->d_sel, sizeof(uint32_t) * ((size_t)ex->sel_per_frame * B + DESC_KPB)));   // k_describe reads whole chunks of 16 slots
    ORBX_HIP(hipMalloc(&ex->d_kps, sizeof(orbx_keypoint) * ex->kcap * B + 16));   // (+ 16: k_result_out reads whole 16-byte pieces)
    ORBX_HIP(hipMalloc(&ex->d_desc, (size_t)32 * ex->kcap * B));
    ORBX_HIP(hipMemcpyAsync(ex->d_lv, ex->lv, sizeof(LevelInfo) * MAXL, hipMemcpyHostToDevice, ex->stream));
    ORBX_HIP(hipMemcpyAsync(ex->d_cells, ex->cells.data(), sizeof(CellInfo) * ex->cells.size(), hipMemcpyHostToDevice, ex->stream));
    ORBX_HIP(hipMemcpyAsync(ex->d_tiles, ex->tiles.data(), sizeof(BlurTile) * ex->tiles.size(), hipMemcpyHostToDevice, ex->stream));
    {
        // Toeplitz fragments of k_blur, per lane (r = lane & 31, h = lane >> 5), byte j of the 16-byte operand:
        // first product: input column 16h + j feeds output column r with tap (16h + j) - r - 1;
        // second product: the lane's j-th row is rho = (j & 3) + 8 (j >> 2) + 4h and feeds output row r with tap rho - r
        const int T[7] = {ex->taps[0], ex->taps[1], ex->taps[2], ex->taps[3], ex->taps[2], ex->taps[1], ex->taps[0]};
        uint8_t fr[64][2][16];
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int r = lane & 31, h = lane >> 5, d1 = 16 * h + j - r - 1, d2 = (j & 3) + 8 * (j >> 2) + 4 * h - r;
                fr[lane][0][j] = (uint8_t)(d1 >= 0 && d1 < 7 ? T[d1] : 0);
                fr[lane][1][j] = (uint8_t)(d2 >= 0 && d2 < 7 ? T[d2] : 0);
            }
        ORBX_HIP(hipMalloc(&ex->d_blur_frag, sizeof(fr)));
        ORBX_HIP(hipMemcpyAsync(ex->d_blur_frag, fr, sizeof(fr), hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipStreamSynchronize(ex->stream));   // fr lives in this scope
    }
    if (!xt.empty()) ORBX_HIP(hipMemcpyAsync(ex->d_xt, xt.data(), sizeof(int2) * xt.size(), hipMemcpyHostToDevice, ex->stream));
    if (!yt.empty()) ORBX_HIP(hipMemcpyAsync(ex->d_yt, yt.data(), sizeof(int4) * yt.size(), hipMemcpyHostToDevice, ex->stream));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    if (ex->oct_lds > 48 * 1024) {
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<OCT_T, OCT_KPT, OCT_KB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, ex->oct_lds));
        ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<OCT_TW, OCT_KPT * OCT_T / OCT_TW, OCT_KPT * OCT_T / OCT_TW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, ex->oct_lds));
    }
    ex->width = width; ex->height = height; ex->batch = batch;
    return ORBX_OK;
}

int orbx_plan(const orbx_params *prm, int width, int height, orbx_plan_info *info)
{
    if (!prm || !info || width <= 0 || height <= 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    orbx_extractor *ex = nullptr;
    int rc = orbx_create(prm, &ex);
    if (rc != ORBX_OK) return rc;
    std::vector<int2> xt;
    std::vector<int4> yt;
    rc = plan_frame(ex, width, height, xt, yt);
    if (rc == ORBX_OK) {
        memset(info, 0, sizeof(*info));
        info->nlevels = ex->nlevels; info->keypoint_capacity = ex->kcap; info->octree_nodes = ex->NC; info->cells_per_frame = ex->cells_per_frame;
        info->sel_per_frame = ex->sel_per_frame; info->fast_tile_stride = ex->TS; info->fast_lds = ex->fast_lds; info->octree_lds = ex->oct_lds;
        info->octree_kshift = ex->oct_kshift; info->frame_bytes = (int64_t)ex->frame_bytes; info->cands_per_frame = (int64_t)ex->cands_per_frame;
        for (int l = 0; l < ex->nlevels; l++) {
            const LevelInfo &lv = ex->lv[l];
            info->level_w[l] = lv.w; info->level_h[l] = lv.h; info->level_quota[l] = lv.N; info->level_nini[l] = lv.nIni;
            info->level_slots[l] = level_slots(lv); info->level_cells[l] = lv.ncells;
        }
    }
    orbx_destroy(ex);
    return rc;
}

int orbx_extract_batch(orbx_extractor *ex, const uint8_t *images, int is_device, int width, int height, int stride,
                       size_t frame_stride, int batch, void *stream_)
{
    if (!ex || !images || width <= 0 || height <= 0 || batch <= 0 || stride < width)
        ORBX_FAIL(ORBX_ERR_ARG, "bad extract arguments");
    ORBX_NEED_DEVICE();
    int rc = orbx_reserve(ex, width, height, batch);
    if (rc != ORBX_OK) return rc;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ex->stream;
    if (ex->reader_pending) {   // a *_dev consumer on another stream still reads the last batch's results (order_after_producer)
        if (ex->reader_stream != st) ORBX_HIP(hipStreamWaitEvent(st, ex->reader_ev, 0));
        ex->reader_pending = false;
    }
    const uint8_t *d_img = images;
    if (!is_device) {
        const size_t need = frame_stride * (size_t)(batch - 1) + (size_t)stride * height;
        if (need > ex->in_bytes) {
            if (ex->d_in) ORBX_HIP(hipFree(ex->d_in));
            ex->d_in = nullptr;
            ORBX_HIP(hipMalloc(&ex->d_in, need));
            ex->in_bytes = need;
        }
        // (what is copied ends with the last pixel of the last row: a cv::Mat with a row pitch -- a region of interest at the bottom of its
        // parent -- owns nothing behind it)
        ORBX_HIP(hipMemcpyAsync(ex->d_in, images, need - (size_t)(stride - width), hipMemcpyHostToDevice, st));
        d_img = ex->d_in;
    }
    const int nl = ex->nlevels;
    orbx::KernelProfiler &pf = ex->prof;
    {
        const LevelInfo &l0 = ex->lv[0];
        const int aligned = (((uintptr_t)d_img | (uintptr_t)stride | (uintptr_t)frame_stride) & 3) == 0;
        pf.start(0, st);
        // the linear form splits a piece index p into (row, piece) with the reciprocal inv = ceil(2^32 / d): exact while
        // p (inv d - 2^32) < 2^32 for every p < rows d -- holds with room for any size orbx_reserve accepts (sides <= 4,000 px:
        // 2 M x 500), checked here all the same so that a larger limit one day cannot shift pixels by a row unnoticed
        auto recip_exact = [](uint64_t count, uint64_t d) {
            const uint64_t inv = (0x100000000ull + d - 1) / d;
            return count == 0 || (count - 1) * (inv * d - 0x100000000ull) < 0x100000000ull;
        };
        const int rows0 = l0.h + 2 * EDGE, pp0 = l0.w / 8, nleft0 = PADX / 4 - ((PADX - EDGE) >> 2);
        const int nb0 = nleft0 + ((PADX + l0.w + EDGE - 1) >> 2) - ((PADX + l0.w) >> 2) + 1;
        if (aligned && l0.w % 8 == 0 && l0.w >= 64 && recip_exact((uint64_t)rows0 * pp0, (uint64_t)pp0) && recip_exact((uint64_t)rows0 * nb0, (uint64_t)nb0)) {
            const int rows = rows0, pp = pp0, nleft = nleft0;
            const int nb = nb0;
            const int nblk_int = (rows * pp + 255) / 256, nblk_b = (rows * nb + 63) / 64;
            hipLaunchKernelGGL(k_pyr_level0_lin, dim3(nblk_int + nblk_b, batch), dim3(64), 0, st, d_img, stride, frame_stride, ex->d_pyr,
                               ex->frame_bytes, l0, pp, (unsigned)((0x100000000ull + pp - 1) / pp), nblk_int, nb,
                               (unsigned)((0x100000000ull + nb - 1) / nb), nleft);
        } else {
            dim3 g((l0.stride / 4 + 63) / 64, (l0.h + 2 * EDGE + PYR_ROWS - 1) / PYR_ROWS, batch);
            hipLaunchKernelGGL(k_pyr_level0, g, dim3(64), 0, st, d_img, stride, frame_stride, ex->d_pyr, ex->frame_bytes, l0, aligned);
        }
        pf.stop(0, st);
    }
    // Small batches -- the live tracker's one frame per call, a camera rig's few -- take the levels 1.. in ONE launch (k_pyr_chain): seven dependent launches of
    // a few microseconds each are 52 of one frame's 134 us of kernels.  Large batches keep the per-level launches (the one-launch form was
    // slower in the pipelined 64-frame step, profiles/r04_notes.md).  ORBX_PYR_CHAIN_MAX_BATCH moves the limit (0: never; tests run both).
    const char *cmb = getenv("ORBX_PYR_CHAIN_MAX_BATCH");            // (read per call: the tests flip it)
    const int chain_max_batch = cmb ? atoi(cmb) : 6;   // (tools/small_batch_sweep.py: ahead up to 6 frames per call, level at 8, behind from 16)
    const bool use_chain = ex->chain_tiles > 0 && batch <= chain_max_batch;
    if (use_chain) {
        pf.start(1, st);
        hipLaunchKernelGGL(k_pyr_chain, dim3(ex->chain_tiles, batch), dim3(256), ex->chain_ldsA + ex->chain_ldsB + ex->chain_ldsT + 2 * MAXL * 16, st, ex->d_pyr, ex->frame_bytes,
                           (const LevelInfo *)ex->d_lv, nl, (const ChainTile *)ex->d_chain, (const uint4 *)ex->d_chain_tabs, (const int2 *)ex->d_chain_span,
                           ex->chain_ldsA, ex->chain_ldsB);
        pf.stop(1, st);
    }
    for (int l = 1; l < nl && !use_chain; l++) {
        const LevelInfo &lv = ex->lv[l];
        const int ngi = lv.w >> 2, xlo = (PADX - EDGE) >> 2, xhi = (PADX + lv.w + EDGE - 1) >> 2;
        // waves of 64 consecutive interior groups of a row group (8-byte-window path); a last, partly filled one when at least
        // 24 groups are left: the byte-by-byte path of the tail tiles costs 64 loads per lane (at levels 6-7 of a 640 x 480
        // frame EVERY group went through it: fewer than 64 interior groups per row)
        const int nff = ex->resize_nxi[l] ? (ngi % 64 >= 24 ? ngi / 64 + 1 : ngi / 64) : 0;
        const int nint = std::min(nff * 64, ngi);                                  // interior groups the full-wave path takes
        const int ntail = (xhi - xlo + 1) - nint;                                  // what is left of a row: leftover interior + border groups
        const int tw_shift = ntail <= 16 ? 4 : ntail <= 32 ? 5 : 6;                // tail tile: 16 x 4, 32 x 2 or 64 x 1 (groups x row groups)
        const int nrg = (lv.h + 2 * EDGE + PYR_ROWS - 1) / PYR_ROWS, rpt = 64 >> tw_shift;
        const int ntw = (ntail + (1 << tw_shift) - 1) >> tw_shift;                 // tail waves across a row (> 1 only when ntail > 64)
        // row groups per full wave: two when one would need a second generation of resident waves (the kernel holds 8 waves
        // per SIMD) -- a second generation costs a whole wave's life, a longer wave only its extra rows.  Measured per 64-frame
        // chain of seven launches: 1 -> 0.100, 2 -> 0.094, 3 -> 0.098, 4 -> 0.103 ms
        const int ntailw = ntw * ((nrg + rpt - 1) / rpt);
        int rpw = 1;
        while (rpw < 2 && (size_t)(nff * ((nrg + rpw - 1) / rpw) + ntailw) * batch > (size_t)ex->resident_waves) ++rpw;
        dim3 g(nff * ((nrg + rpw - 1) / rpw) + ntailw, batch);
        pf.start(1, st);
        hipLaunchKernelGGL(k_pyr_resize, g, dim3(64), 0, st, ex->d_pyr, ex->frame_bytes, ex->lv[l - 1], lv,
                           ex->d_xt + lv.xtab, ex->d_yt + lv.ytab, nff, nint, nrg, ntail, tw_shift, ntw, rpw, ex->resize_tailwin[l]);
        pf.stop(1, st);
    }
    {   // (profiled: the kernel's own start and end through hipExtLaunchKernelGGL -- see KernelProfiler::pair)
        hipEvent_t e0, e1;
        pf.pair(2, &e0, &e1);
#define ORBX_FAST_LAUNCH(TS_, SS_) hipExtLaunchKernelGGL((k_fast_cells<TS_, SS_>), dim3(ex->cells_per_frame, batch), dim3(64), ex->fast_lds, st, e0, e1, 0, \
            (const uint8_t *)ex->d_pyr, ex->frame_bytes, (const LevelInfo *)ex->d_lv, (const CellInfo *)ex->d_cells, ex->d_cell_count, ex->cells_per_frame, ex->d_cands, \
            ex->cands_per_frame, ex->prm.ini_th_fast, ex->prm.min_th_fast, ex->tile_bytes, ex->sc_bytes, ex->queue_bytes)
        if (ex->TS == 52) ORBX_FAST_LAUNCH(52, 40);
        else if (ex->TS == 64) ORBX_FAST_LAUNCH(64, 64);
        else ORBX_FAST_LAUNCH(80, 80);
#undef ORBX_FAST_LAUNCH
    }
    pf.start(3, st);
    // Small batches: OCT_TW = 512 threads per (level, frame) -- a level's workgroup is the one frame's critical path, and its passes over
    // keys and nodes are loops of dependent LDS reads that twice the threads make half as long: 43.6 -> 37.0 us for one frame (1,024
    // threads: 36.5 -- what is left is the chain of scans and barriers).  Large batches fill the chip with 256-thread workgroups
    // (ORBX_OCT_WIDE_MAX_BATCH moves the limit; tests run both).
    const char *owb = getenv("ORBX_OCT_WIDE_MAX_BATCH");
    if (batch <= (owb ? atoi(owb) : 6))
        hipLaunchKernelGGL((k_octree<OCT_TW, OCT_KPT * OCT_T / OCT_TW, OCT_KPT * OCT_T / OCT_TW>), dim3(nl, batch), dim3(OCT_TW), ex->oct_lds, st, ex->d_lv, ex->d_cells,
                           ex->d_cell_count, ex->cells_per_frame, ex->d_cands, ex->cands_per_frame, ex->d_kpos,
                           ex->d_knode, ex->d_kq, ex->keys_per_frame, ex->d_sel, ex->sel_per_frame, ex->d_level_count,
                           ex->d_level_ncand, nl, ex->NC, ex->maxcells, ex->oct_kcap, ex->oct_kshift);
    else
        hipLaunchKernelGGL((k_octree<OCT_T, OCT_KPT, OCT_KB>), dim3(nl, batch), dim3(OCT_T), ex->oct_lds, st, ex->d_lv, ex->d_cells,
                           ex->d_cell_count, ex->cells_per_frame, ex->d_cands, ex->cands_per_frame, ex->d_kpos,
                           ex->d_knode, ex->d_kq, ex->keys_per_frame, ex->d_sel, ex->sel_per_frame, ex->d_level_count,
                           ex->d_level_ncand, nl, ex->NC, ex->maxcells, ex->oct_kcap, ex->oct_kshift);
    pf.stop(3, st);
    pf.start(4, st);
    {
        // with H'' = sum (P - 128) T + bias = 256 hi + lo' + 128:  H = H'' - bias + 128 S, so
        // V + 2^15 = 256 sum(T hi) + sum(T lo') + S (128 - bias + 128 S) + 2^15     (S = tap sum, bias = 128 iff S > 256)
        const int S = 2 * (ex->taps[0] + ex->taps[1] + ex->taps[2]) + ex->taps[3], nt = (int)ex->tiles.size();
        const dim3 grid((unsigned)((nt + 3) / 4), batch);
        if (S > 256)
            hipLaunchKernelGGL(k_blur<true>, grid, dim3(256), 0, st, ex->d_pyr, ex->d_blur, ex->frame_bytes, ex->blur_frame_bytes, ex->d_lv, ex->d_tiles, nt,
                               ex->d_blur_frag, 128 * S * S + 32768);
        else
            hipLaunchKernelGGL(k_blur<false>, grid, dim3(256), 0, st, ex->d_pyr, ex->d_blur, ex->frame_bytes, ex->blur_frame_bytes, ex->d_lv, ex->d_tiles, nt,
                               ex->d_blur_frag, S * (128 + 128 * S) + 32768);
    }
    pf.stop(4, st);
    pf.start(5, st);
    {
        DescLevels D;
        memset(&D, 0, sizeof(D));
        int nchunks = 0;
        for (int l = 0; l < nl; l++) {
            const LevelInfo &lv = ex->lv[l];
            D.off[l] = lv.off; D.stride[l] = lv.stride; D.sel_base[l] = lv.sel_base; D.patch[l] = lv.patch; D.scale[l] = lv.scale;
            D.boff[l] = ex->boff[l]; D.bcol[l] = ex->bcol[l];
            D.chunk_base[l] = nchunks;
            nchunks += (level_slots(lv) + DESC_KPB - 1) / DESC_KPB;   // a level holds at most max(N + 3, 4 nIni) keypoints
        }
        for (int l = nl; l <= MAXL; l++) D.chunk_base[l] = nchunks;
        hipLaunchKernelGGL(k_describe, dim3(nchunks, batch), dim3(256), 0, st, ex->d_pyr, ex->d_blur, ex->frame_bytes, ex->blur_frame_bytes, D, nl,
                           ex->d_sel, ex->sel_per_frame, ex->d_level_count, ex->d_kps, ex->d_desc, ex->d_counts, ex->kcap, ex->prm.trig_variant,
                           batch == 1 ? ex->mirror : (uint32_t *)nullptr, ex->mirror_desc);
    }
    pf.stop(5, st);
    ORBX_HIP(hipGetLastError());
    ex->last_batch = batch;
    ex->last_stream = st;
    return ORBX_OK;
}

int orbx_profile_enable(orbx_extractor *ex, int on)
{
    if (!ex) ORBX_FAIL(ORBX_ERR_ARG, "null argument");
    static const char *names[6] = {"k_pyr_level0", "k_pyr_resize", "k_fast_cells", "k_octree", "k_blur", "k_describe"};
    for (int i = 0; i < 6; ++i) ex->prof.names[i] = names[i];
    ex->prof.reset();
    ex->prof.mask = on < 0 ? 0xffffu : (unsigned)on;
    return ORBX_OK;
}

#ifdef ORBX_PHASE_TIMING
int orbx_debug_phases(unsigned long long *out, int reset)   // out: 2 x 65536 x 16 records (FAST, describe)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_rec), sizeof(unsigned long long) * 4 * 65536 * 16) != hipSuccess) return -1;
    if (reset) {
        void *p = nullptr;
        hipStream_t st = nullptr;   // (a stream of its own: nothing in this library touches the legacy stream)
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase_rec)) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -1;
        const bool ok = hipMemsetAsync(p, 0, sizeof(unsigned long long) * 4 * 65536 * 16, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        (void)hipStreamDestroy(st);
        if (!ok) return -1;
    }
    return 0;
}
#endif

int orbx_profile_read(orbx_extractor *ex, int max_kinds, const char **names, double *total_ms, int64_t *launches, int *nkinds)
{
    if (!ex || !nkinds) ORBX_FAIL(ORBX_ERR_ARG, "null argument");
    ex->prof.flush();
    int n = 0;
    for (int i = 0; i < orbx::KernelProfiler::MAXK && n < max_kinds; ++i) {
        if (!ex->prof.names[i]) continue;
        if (names) names[n] = ex->prof.names[i];
        if (total_ms) total_ms[n] = ex->prof.ms[i];
        if (launches) launches[n] = ex->prof.launches[i];
        ++n;
    }
    *nkinds = n;
    return ORBX_OK;
}

int orbx_download(orbx_extractor *ex, int frame, orbx_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!ex || frame < 0 || frame >= ex->last_batch || !n) ORBX_FAIL(ORBX_ERR_ARG, "bad download arguments");
    // copies queued behind the batch on ITS stream and one wait for that stream: the other SLAM threads' streams
    // (LocalMapping, LoopClosing) are not stalled, as a device-wide synchronisation would
    hipStream_t st = ex->last_stream;
    int cnt = 0;
    ORBX_HIP(hipMemcpyAsync(&cnt, ex->d_counts + frame, sizeof(int), hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    *n = cnt;
    if (cnt > cap) ORBX_FAIL(ORBX_ERR_CAPACITY, "keypoint buffer too small");
    if (cnt > 0) {
        if (kps) ORBX_HIP(hipMemcpyAsync(kps, ex->d_kps + (size_t)frame * ex->kcap, sizeof(orbx_keypoint) * cnt, hipMemcpyDeviceToHost, st));
        if (desc) ORBX_HIP(hipMemcpyAsync(desc, ex->d_desc + (size_t)frame * ex->kcap * 32, (size_t)32 * cnt, hipMemcpyDeviceToHost, st));
        ORBX_HIP(hipStreamSynchronize(st));
    }
    return ORBX_OK;
}

int orbx_download_batch(orbx_extractor *ex, orbx_keypoint *kps, uint8_t *desc, int32_t *counts)
{
    if (!ex || ex->last_batch <= 0 || !ex->d_kps) ORBX_FAIL(ORBX_ERR_ARG, "no results");
    hipStream_t st = ex->last_stream;
    const size_t B = (size_t)ex->last_batch;
    if (counts) ORBX_HIP(hipMemcpyAsync(counts, ex->d_counts, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    if (kps) ORBX_HIP(hipMemcpyAsync(kps, ex->d_kps, sizeof(orbx_keypoint) * ex->kcap * B, hipMemcpyDeviceToHost, st));
    if (desc) ORBX_HIP(hipMemcpyAsync(desc, ex->d_desc, (size_t)32 * ex->kcap * B, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}

// One host image through the extractor, in two halves: everything up to the last launch (extract_enqueue), and the wait + the copy
// out of the pinned block (extract_finish).  orbx_extract is one after the other; orbx_extract_pair enqueues the left and the right
// image of a stereo frame on their two handles' streams before it waits for either.
static int extract_enqueue(orbx_extractor *ex, const uint8_t *image, int width, int height, int stride, bool as_graph = false)
{
    // ORBextractor::operator() on one host image is latency-bound: stage the image and the results through pinned
    // buffers so that the call is one H2D, the kernel chain, one D2H and a single stream synchronisation
    if (stride < width) ORBX_FAIL(ORBX_ERR_ARG, "row pitch smaller than the width");
    const size_t in_bytes = (size_t)stride * height, in_room = (in_bytes + 255) & ~(size_t)255;   // 16-byte pieces on both sides
    int rc = orbx_reserve(ex, width, height, 1);
    if (rc != ORBX_OK) return rc;
    const size_t kp_bytes = (sizeof(orbx_keypoint) * (size_t)ex->kcap + 15) & ~(size_t)15, de_bytes = (size_t)32 * ex->kcap;
    const size_t out_bytes = 16 + kp_bytes + de_bytes;
    if (in_room + out_bytes > ex->pin_bytes) {
        drop_graph(ex);
        if (ex->h_pin) (void)hipHostFree(ex->h_pin);
        ex->h_pin = nullptr; ex->pin_bytes = 0;
        ORBX_HIP(hipHostMalloc((void **)&ex->h_pin, in_room + out_bytes, hipHostMallocDefault));
        ex->pin_bytes = in_room + out_bytes;
    }
    hipStream_t st = ex->stream;
    if (ex->reader_pending) {   // a consumer on another stream (the stereo matcher, a *_dev matcher call) still reads the last results
        if (ex->reader_stream != st) ORBX_HIP(hipStreamWaitEvent(st, ex->reader_ev, 0));
        ex->reader_pending = false;
    }
    // An image that already lies in pinned (hipHostMalloc'ed or hipHostRegister'ed) memory -- a capture buffer the caller allocated that way --
    // is read where it is: no staging copy (10 us of a 111-us call for 640 x 480).  Not for the pair call's captured graph, whose kernel
    // arguments are fixed.
    const uint8_t *src = ex->h_pin;
    if (!as_graph) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, image) == hipSuccess && pa.type == hipMemoryTypeHost && pa.devicePointer) src = static_cast<const uint8_t *>(pa.devicePointer);
        else (void)hipGetLastError();
    }
    if (src == ex->h_pin) memcpy(ex->h_pin, image, in_bytes - (size_t)(stride - width));   // (ends with the last row's last pixel: a region of interest owns nothing behind it)
    ex->pin_result_off = in_room;
    // image in and results out by the compute queue itself (the kernels read / write the pinned block): no hand-over to the copy engine
    // in front of and behind the kernels of a frame
    auto enqueue = [&]() -> int {
        // (level 0 reads the pinned image itself -- it is mapped into the device's address space --: 16 us instead of a staging copy + 7, and
        // one launch less; 0.149 -> 0.146 ms per call)
        // (the results: k_describe stores every record into the pinned block as well -- no copy kernel behind it)
        ex->mirror = reinterpret_cast<uint32_t *>(ex->h_pin + in_room); ex->mirror_desc = (int)((16 + kp_bytes) / 4);
        const int rc2 = orbx_extract_batch(ex, src, 1, width, height, stride, in_bytes, 1, st);
        ex->mirror = nullptr;
        if (rc2 != ORBX_OK) return rc2;
        return ORBX_OK;
    };
    // As a graph (orbx_extract_pair): the launches of a frame (fourteen when this was written, seven now) cost the host time to enqueue, which is what the SECOND
    // image of a stereo frame waits for; one graph launch per image lets the two chains run side by side on the device.  Captured
    // on the second call of a frame size (the first has done every one-time set-up), only while nothing in the chain depends on
    // the call (no per-kernel profiling events; a pending reader of the last results is waited for in front of the chain); any
    // failure falls back to plain launches for good.
    // ORBX_NO_GRAPHS=1 switches the capture off: while a stream captures (once per frame size, ~0.2 ms), a copy on the legacy stream
    // from ANY other thread fails with hipErrorStreamCaptureImplicit (tests/stress_threads.py met it in the FEM solver's former graph)
    // -- a host that cannot make its first two pairs before its other threads call into the library should set it.
    static const bool graphs_allowed = getenv("ORBX_NO_GRAPHS") == nullptr;
    const bool can = as_graph && graphs_allowed && ex->prof.mask == 0 && !ex->g_failed;
    if (can && ex->g_exec && ex->g_w == width && ex->g_h == height && ex->g_stride == stride) {
        ORBX_HIP(hipGraphLaunch(ex->g_exec, st));
        ex->last_batch = 1; ex->last_stream = st;
        return ORBX_OK;
    }
    const bool second = ex->g_seen_w == width && ex->g_seen_h == height && ex->g_seen_stride == stride;
    ex->g_seen_w = width; ex->g_seen_h = height; ex->g_seen_stride = stride;
    if (can && second && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        rc = enqueue();
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(st, &g);
        hipGraphExec_t x = nullptr;
        if (rc == ORBX_OK && e == hipSuccess && g && hipGraphInstantiate(&x, g, nullptr, nullptr, 0) == hipSuccess) {
            drop_graph(ex);
            ex->g_graph = g; ex->g_exec = x; ex->g_w = width; ex->g_h = height; ex->g_stride = stride;
            ORBX_HIP(hipGraphLaunch(ex->g_exec, st));
            return ORBX_OK;
        }
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        ex->g_failed = true;        // nothing of the captured chain has run: enqueue it plainly
    }
    return enqueue();
}

static int extract_finish(orbx_extractor *ex, orbx_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const size_t kp_bytes = (sizeof(orbx_keypoint) * (size_t)ex->kcap + 15) & ~(size_t)15;
    const uint8_t *o = ex->h_pin + ex->pin_result_off;
    int cnt = 0;
    memcpy(&cnt, o, sizeof(int));
    *n = cnt;
    if (cnt > cap) ORBX_FAIL(ORBX_ERR_CAPACITY, "keypoint buffer too small");
    if (cnt > 0) {
        if (kps) memcpy(kps, o + 16, sizeof(orbx_keypoint) * cnt);
        if (desc) memcpy(desc, o + 16 + kp_bytes, (size_t)32 * cnt);
    }
    return ORBX_OK;
}

int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, int stride, orbx_keypoint *kps,
                 uint8_t *desc, int cap, int *n)
{
    if (!ex || !n) ORBX_FAIL(ORBX_ERR_ARG, "null argument");
    if (!image || width <= 0 || height <= 0) { *n = 0; return ORBX_OK; } // empty image: outputs untouched (:1054)
    const int rc = extract_enqueue(ex, image, width, height, stride);
    return rc != ORBX_OK ? rc : extract_finish(ex, kps, desc, cap, n);
}

int orbx_extract_pair(orbx_extractor *left, const uint8_t *image_left, orbx_extractor *right, const uint8_t *image_right, int width,
                      int height, int stride, orbx_keypoint *kps_left, uint8_t *desc_left, int cap_left, int *n_left,
                      orbx_keypoint *kps_right, uint8_t *desc_right, int cap_right, int *n_right)
{
    if (!left || !right || left == right || !n_left || !n_right || !image_left || !image_right || width <= 0 || height <= 0 || stride < width)
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    // both chains are on their queues before the host waits for either: the two extractions of a stereo frame overlap on the
    // device as they do on the reference's two threads (Frame.cc:78-81), without the threads
    int rc = extract_enqueue(left, image_left, width, height, stride, true);
    if (rc == ORBX_OK) rc = extract_enqueue(right, image_right, width, height, stride, true);
    if (rc != ORBX_OK) { (void)hipStreamSynchronize(left->stream); return rc; }
    rc = extract_finish(left, kps_left, desc_left, cap_left, n_left);
    const int rc2 = extract_finish(right, kps_right, desc_right, cap_right, n_right);
    return rc != ORBX_OK ? rc : rc2;
}

int orbx_result_dev(orbx_extractor *ex, const orbx_keypoint **kps, const uint8_t **desc, const int32_t **counts, int *capacity)
{
    if (!ex || !ex->d_kps) ORBX_FAIL(ORBX_ERR_ARG, "no results");
    if (kps) *kps = ex->d_kps;
    if (desc) *desc = ex->d_desc;
    if (counts) *counts = ex->d_counts;
    if (capacity) *capacity = ex->kcap;
    return ORBX_OK;
}

int orbx_copy_results_dev(orbx_extractor *ex, orbx_keypoint *kps_dst, uint8_t *desc_dst, int32_t *counts_dst, void *stream_)
{
    if (!ex || !ex->d_kps || ex->last_batch <= 0) ORBX_FAIL(ORBX_ERR_ARG, "no results");
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ex->stream;
    const size_t B = (size_t)ex->last_batch;
    if (kps_dst) ORBX_HIP(hipMemcpyAsync(kps_dst, ex->d_kps, sizeof(orbx_keypoint) * ex->kcap * B, hipMemcpyDefault, st));
    if (desc_dst) ORBX_HIP(hipMemcpyAsync(desc_dst, ex->d_desc, (size_t)32 * ex->kcap * B, hipMemcpyDefault, st));
    if (counts_dst) ORBX_HIP(hipMemcpyAsync(counts_dst, ex->d_counts, sizeof(int) * B, hipMemcpyDefault, st));
    return ORBX_OK;
}

static int copy_level(orbx_extractor *ex, const uint8_t *buf, int frame, int level, uint8_t *out, int out_stride, int padded)
{
    if (!ex || !out || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->nlevels)
        ORBX_FAIL(ORBX_ERR_ARG, "bad frame/level");
    const LevelInfo &lv = ex->lv[level];
    const int w = padded ? lv.w + 2 * EDGE : lv.w, h = padded ? lv.h + 2 * EDGE : lv.h;
    if (out_stride < w) ORBX_FAIL(ORBX_ERR_ARG, "out_stride too small");
    const uint8_t *src = buf + (size_t)frame * ex->frame_bytes + lv.off +
                         (padded ? (size_t)(PADX - EDGE) : (size_t)EDGE * lv.stride + PADX);
    ORBX_HIP(hipMemcpy2DAsync(out, out_stride, src, lv.stride, w, h, hipMemcpyDeviceToHost, ex->last_stream));
    ORBX_HIP(hipStreamSynchronize(ex->last_stream));
    return ORBX_OK;
}

int orbx_pyramid_level(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride)
{ return copy_level(ex, ex ? ex->d_pyr : nullptr, frame, level, out, out_stride, 0); }
int orbx_pyramid_level_padded(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride)
{ return copy_level(ex, ex ? ex->d_pyr : nullptr, frame, level, out, out_stride, 1); }
int orbx_debug_blurred_level(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride)
{
    // the blurred buffer is laid out in strips (orbx_internal.h): the level's strips come over whole and are put back into rows here
    if (!ex || !ex->d_blur || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->nlevels || !out)
        ORBX_FAIL(ORBX_ERR_ARG, "bad frame/level");
    const LevelInfo &lv = ex->lv[level];
    if (out_stride < lv.w) ORBX_FAIL(ORBX_ERR_ARG, "out_stride too small");
    const int nstrips = lv.stride / 16, bcol = ex->bcol[level];
    std::vector<uint8_t> strips((size_t)nstrips * bcol);
    ORBX_HIP(hipMemcpyAsync(strips.data(), ex->d_blur + (size_t)frame * ex->blur_frame_bytes + ex->boff[level], strips.size(),
                            hipMemcpyDeviceToHost, ex->last_stream));
    ORBX_HIP(hipStreamSynchronize(ex->last_stream));
    for (int y = 0; y < lv.h; ++y)
        for (int x = 0; x < lv.w; ++x) {
            const int X = PADX + x;
            out[(size_t)y * out_stride + x] = strips[(size_t)(X >> 4) * bcol + (size_t)(y + EDGE) * 16 + (X & 15)];
        }
    return ORBX_OK;
}

int orbx_debug_level_candidates(orbx_extractor *ex, int frame, int level, float *xyr, int cap, int *n)
{
    if (!ex || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->nlevels || !n)
        ORBX_FAIL(ORBX_ERR_ARG, "bad frame/level");
    hipStream_t st = ex->last_stream;
    int M = 0;
    ORBX_HIP(hipMemcpyAsync(&M, ex->d_level_ncand + frame * ex->nlevels + level, sizeof(int), hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    *n = M;
    if (M > cap) ORBX_FAIL(ORBX_ERR_CAPACITY, "candidate buffer too small");
    std::vector<uint32_t> pk(M > 0 ? M : 1);
    if (M > 0) {
        ORBX_HIP(hipMemcpyAsync(pk.data(), ex->d_kpos + (size_t)frame * ex->keys_per_frame + ex->lv[level].key_base,
                                sizeof(uint32_t) * M, hipMemcpyDeviceToHost, st));
        ORBX_HIP(hipStreamSynchronize(st));
    }
    for (int i = 0; i < M; i++) {
        xyr[3 * i] = (float)((pk[i] >> 8) & 0xfffu);
        xyr[3 * i + 1] = (float)(pk[i] >> 20);
        xyr[3 * i + 2] = (float)(pk[i] & 0xffu);
    }
    return ORBX_OK;
}

int orbx_debug_level_keypoints(orbx_extractor *ex, int frame, int level, orbx_keypoint *kps, int cap, int *n)
{
    if (!ex || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->nlevels || !n)
        ORBX_FAIL(ORBX_ERR_ARG, "bad frame/level");
    hipStream_t st = ex->last_stream;
    std::vector<int> lc(ex->nlevels);
    ORBX_HIP(hipMemcpyAsync(lc.data(), ex->d_level_count + frame * ex->nlevels, sizeof(int) * ex->nlevels, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    int first = 0;
    for (int l = 0; l < level; l++) first += lc[l];
    const int cnt = lc[level];
    *n = cnt;
    if (cnt > cap) ORBX_FAIL(ORBX_ERR_CAPACITY, "keypoint buffer too small");
    if (cnt == 0) return ORBX_OK;
    std::vector<uint32_t> pk(cnt);
    ORBX_HIP(hipMemcpyAsync(pk.data(), ex->d_sel + (size_t)frame * ex->sel_per_frame + ex->lv[level].sel_base,
                            sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(kps, ex->d_kps + (size_t)frame * ex->kcap + first, sizeof(orbx_keypoint) * cnt, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < cnt; i++) { // level coordinates, before pt *= scale
        kps[i].x = (float)((pk[i] >> 8) & 0xfffu) + MIN_BORDER;
        kps[i].y = (float)(pk[i] >> 20) + MIN_BORDER;
    }
    return ORBX_OK;
}

} // extern "C"
