/*
 * oracle/orb_oracle.c -- CPU restatement of the ORB extractor path of ORB_SLAM2_E.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under orb_slam2_e_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors for this path
 * and cannot be built here (needs OpenCV 3.4, absent).  This file restates
 *   - the reference's own logic, citing src/ORBextractor.cc line ranges, and
 *   - the five OpenCV 3.4 primitives it calls (cv::FAST, cv::resize INTER_LINEAR,
 *     cv::copyMakeBorder REFLECT_101, cv::GaussianBlur 7x7 sigma 2, cv::fastAtan2,
 *     cvRound) from their published algorithms (SURVEY.md Appendix B is the
 *     normative spec; OpenCV pinned by CMakeLists.txt:19 "find_package(OpenCV 3.4)").
 * It is pinned only by first-principles known-answer tests (tests/test_oracle_*.py).
 *
 * Compile: gcc -O2 -ffp-contract=off (no FMA contraction: SURVEY F9).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stddef.h>

#include "orb_pattern_data.h"

#define PATCH_SIZE 31        /* ORBextractor.cc:72 */
#define HALF_PATCH_SIZE 15   /* ORBextractor.cc:73 */
#define EDGE_THRESHOLD 19    /* ORBextractor.cc:74 */
#define MAX_LEVELS 16

typedef struct {
    float x, y, size, angle, response;
    int octave, class_id;
} oracle_keypoint; /* same 28-byte layout as cv::KeyPoint */

typedef struct {
    float x, y, response;
} oracle_cand;

typedef struct {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    float scaleFactor;
    float mvScaleFactor[MAX_LEVELS], mvInvScaleFactor[MAX_LEVELS];
    float mvLevelSigma2[MAX_LEVELS], mvInvLevelSigma2[MAX_LEVELS];
    int mnFeaturesPerLevel[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    int blur_taps[7];
    int trig_variant; /* 0: the C library's cosf / sinf (glibc >= 2.28 for the documented results); 1: (float)cos((double)x) */
    /* per-call state (mvImagePyramid analogue) */
    int lw[MAX_LEVELS], lh[MAX_LEVELS], lstride[MAX_LEVELS];
    uint8_t *padded[MAX_LEVELS]; /* (lw+38) x (lh+38), stride lstride */
    uint8_t *blurred[MAX_LEVELS]; /* lw x lh contiguous */
    oracle_cand *cands[MAX_LEVELS];
    int ncands[MAX_LEVELS];
    oracle_keypoint *lkps[MAX_LEVELS]; /* per level, level coordinates */
    int nlkps[MAX_LEVELS];
} oracle_orb;

/* ---------------------------------------------------------------- primitives */

/* cvRound: round-half-to-even (SSE cvtss2si / lrint). SURVEY App. B. */
int oracle_cvRound(float v) { return (int)lrintf(v); }
static int cvRoundD(double v) { return (int)lrint(v); }
static int cvFloorF(float v) { return (int)floorf(v); }

/* cv::fastAtan2(y,x) in degrees, OpenCV 3.4 scalar path. */
float oracle_fastAtan2(float y, float x)
{
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* BORDER_REFLECT_101 index map: -k -> k, n-1+k -> n-1-k. */
static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* cv::resize(..., INTER_LINEAR) for CV_8UC1, OpenCV 3.4 fixed-point path:
 * 11-bit coefficients, horizontal pass to int32, vertical pass
 * (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  SURVEY App. B. */
void oracle_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    int dx, dy;
    for (dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        {
            float c0 = 1.f - fx, c1 = fx;
            int a0 = oracle_cvRound(c0 * 2048.f), a1 = oracle_cvRound(c1 * 2048.f);
            ialpha[2 * dx] = (short)(a0 > 32767 ? 32767 : a0);
            ialpha[2 * dx + 1] = (short)(a1 > 32767 ? 32767 : a1);
        }
    }
    for (dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        int sy0, sy1;
        short b0, b1;
        const uint8_t *S0, *S1;
        fy -= sy;
        b0 = (short)oracle_cvRound((1.f - fy) * 2048.f);
        b1 = (short)oracle_cvRound(fy * 2048.f);
        sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        S0 = src + (size_t)sy0 * sstride;
        S1 = src + (size_t)sy1 * sstride;
        for (dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* weight is 0 there */
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        for (dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dstride + dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* Fill the EDGE_THRESHOLD border of a padded buffer by REFLECT_101 of the inner
 * w x h image (cv::copyMakeBorder, ORBextractor.cc:1130,1135). */
static void fill_border101(uint8_t *padded, int w, int h, int stride)
{
    int W = w + 2 * EDGE_THRESHOLD, H = h + 2 * EDGE_THRESHOLD, x, y;
    for (y = 0; y < H; y++) {
        int sy = reflect101(y - EDGE_THRESHOLD, h) + EDGE_THRESHOLD;
        for (x = 0; x < W; x++) {
            int sx = reflect101(x - EDGE_THRESHOLD, w) + EDGE_THRESHOLD;
            if (sy != y || sx != x)
                padded[(size_t)y * stride + x] = padded[(size_t)sy * stride + sx];
        }
    }
}

/* cv::GaussianBlur(src,dst,Size(7,7),2,2,BORDER_REFLECT_101) on CV_8UC1:
 * 8.8 fixed-point taps, exact 16.16 accumulation, (v + 2^15) >> 16, saturate.
 * taps default {18,34,48,56,48,34,18} (sum exactly 256; SURVEY App. B canonical);
 * {18,34,49,55,49,34,18} is the individually-rounded variant of 3.2..3.4.8. */
void oracle_gauss7(const uint8_t *src, int w, int h, int sstride,
                   uint8_t *dst, int dstride, const int *taps)
{
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * h);
    int *rx = (int *)malloc(sizeof(int) * (w + 6)), *ry = (int *)malloc(sizeof(int) * (h + 6));
    int x, y, k;
    for (x = -3; x < w + 3; x++) rx[x + 3] = reflect101(x, w);
    for (y = -3; y < h + 3; y++) ry[y + 3] = reflect101(y, h);
    for (y = 0; y < h; y++) {
        const uint8_t *row = src + (size_t)y * sstride;
        for (x = 0; x < w; x++) {
            uint32_t s = 0;
            for (k = 0; k < 7; k++) s += (uint32_t)taps[k] * row[rx[x + k]];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            uint32_t s = 0;
            for (k = 0; k < 7; k++) s += (uint32_t)taps[k] * tmp[(size_t)ry[y + k] * w + x];
            s = (s + (1u << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(tmp); free(rx); free(ry);
}

/* 16-pixel Bresenham circle of radius 3 (OpenCV makeOffsets, patternSize 16). */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* OpenCV cornerScore<16>: largest threshold for which p is still a FAST-9 corner. */
int oracle_fast_corner_score(const uint8_t *p, int stride, int threshold)
{
    enum { K = 8, N = K * 3 + 1 };
    int k, v = p[0];
    short d[N];
    int a0, b0;
    for (k = 0; k < N; k++)
        d[k] = (short)(v - p[fast_dy[k % 16] * stride + fast_dx[k % 16]]);
    a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        if (d[k + 4] < a) a = d[k + 4];
        if (d[k + 5] < a) a = d[k + 5];
        if (d[k + 6] < a) a = d[k + 6];
        if (d[k + 7] < a) a = d[k + 7];
        if (d[k + 8] < a) a = d[k + 8];
        { int m = a < d[k] ? a : d[k]; if (m > a0) a0 = m; }
        { int m = a < d[k + 9] ? a : d[k + 9]; if (m > a0) a0 = m; }
    }
    b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] > b) b = d[k + 3];
        if (d[k + 4] > b) b = d[k + 4];
        if (d[k + 5] > b) b = d[k + 5];
        if (b >= b0) continue;
        if (d[k + 6] > b) b = d[k + 6];
        if (d[k + 7] > b) b = d[k + 7];
        if (d[k + 8] > b) b = d[k + 8];
        { int m = b > d[k] ? b : d[k]; if (m < b0) b0 = m; }
        { int m = b > d[k + 9] ? b : d[k + 9]; if (m < b0) b0 = m; }
    }
    return -b0 - 1;
}

/* FAST-9/16 segment test: >= 9 contiguous circle pixels all darker than v-t or
 * all brighter than v+t (OpenCV FAST_t<16> inner loop, count > K over N = 25). */
int oracle_fast_is_corner(const uint8_t *p, int stride, int threshold)
{
    enum { K = 8, N = 25 };
    int v = p[0], k, count, vt;
    vt = v - threshold; count = 0;
    for (k = 0; k < N; k++) {
        int x = p[fast_dy[k % 16] * stride + fast_dx[k % 16]];
        if (x < vt) { if (++count > K) return 1; } else count = 0;
    }
    vt = v + threshold; count = 0;
    for (k = 0; k < N; k++) {
        int x = p[fast_dy[k % 16] * stride + fast_dx[k % 16]];
        if (x > vt) { if (++count > K) return 2; } else count = 0;
    }
    return 0;
}

/* cv::FAST(img, kps, threshold, nonmaxSuppression=true) on a cw x ch sub-image.
 * Output (x, y, score) in raster order; returns count (<= cap written).
 * Like OpenCV's FAST_t<16>, a pixel first passes the opposite-pair pre-test
 * (d = (tab[p0]|tab[p8]) & (tab[p2]|tab[p10]) & ...: a 9-arc contains one pixel of
 * every opposite pair) before the 25-step segment test runs. */
int oracle_fast_detect(const uint8_t *img, int cw, int ch, int stride, int threshold,
                       oracle_cand *out, int cap)
{
    int n = 0, x, y, k;
    uint8_t *score, *iscorner;
    if (cw < 7 || ch < 7) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    score = (uint8_t *)calloc((size_t)cw * ch, 1);
    iscorner = (uint8_t *)calloc((size_t)cw * ch, 1);
    for (y = 3; y < ch - 3; y++)
        for (x = 3; x < cw - 3; x++) {
            const uint8_t *p = img + (size_t)y * stride + x;
            const int v = p[0], lo = v - threshold, hi = v + threshold;
            int d = 3; /* bit 0: darker arc still possible, bit 1: brighter arc still possible */
            for (k = 0; k < 8 && d; k++) {
                const int a = p[fast_dy[k] * stride + fast_dx[k]], b = p[fast_dy[k + 8] * stride + fast_dx[k + 8]];
                d &= ((a < lo || b < lo) ? 1 : 0) | ((a > hi || b > hi) ? 2 : 0);
            }
            if (!d) continue;
            if (oracle_fast_is_corner(p, stride, threshold)) {
                iscorner[(size_t)y * cw + x] = 1;
                score[(size_t)y * cw + x] = (uint8_t)oracle_fast_corner_score(p, stride, threshold);
            }
        }
    for (y = 3; y < ch - 3; y++)
        for (x = 3; x < cw - 3; x++) {
            int s;
            const uint8_t *r0, *r1, *r2;
            if (!iscorner[(size_t)y * cw + x]) continue;
            s = score[(size_t)y * cw + x];
            r0 = score + (size_t)(y - 1) * cw + x;
            r1 = score + (size_t)y * cw + x;
            r2 = score + (size_t)(y + 1) * cw + x;
            if (s > r1[1] && s > r1[-1] && s > r0[-1] && s > r0[0] && s > r0[1] &&
                s > r2[-1] && s > r2[0] && s > r2[1]) {
                if (n < cap) { out[n].x = (float)x; out[n].y = (float)y; out[n].response = (float)s; }
                n++;
            }
        }
    free(score); free(iscorner);
    return n;
}

/* ------------------------------------------------- DistributeOctTree restated */

typedef struct onode {
    int ULx, ULy, URx, BRy; /* UL=(ULx,ULy) UR=(URx,ULy) BL=(ULx,BRy) BR=(URx,BRy) */
    int *keys; int nkeys;
    int bNoMore;
    int seq;                /* creation sequence number: R14 tie-break */
    struct onode *prev, *next;
} onode;

typedef struct { onode *head, *tail; int size; int seq; } olist;

static onode *node_new(olist *l, int cap)
{
    onode *n = (onode *)calloc(1, sizeof(onode));
    n->keys = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    n->seq = l->seq++;
    return n;
}
static void list_push_front(olist *l, onode *n)
{
    n->prev = NULL; n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n; l->size++;
}
static void list_push_back(olist *l, onode *n)
{
    n->next = NULL; n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n; l->size++;
}
static onode *list_erase(olist *l, onode *n)
{
    onode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    free(n->keys); free(n);
    return nx;
}

/* ExtractorNode::DivideNode, ORBextractor.cc:481-537 */
static void divide_node(olist *l, const onode *p, const oracle_cand *c, onode *ch[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    int i;
    for (i = 0; i < 4; i++) ch[i] = node_new(l, p->nkeys);
    ch[0]->ULx = p->ULx;         ch[0]->URx = p->ULx + halfX; ch[0]->ULy = p->ULy;         ch[0]->BRy = p->ULy + halfY;
    ch[1]->ULx = p->ULx + halfX; ch[1]->URx = p->URx;         ch[1]->ULy = p->ULy;         ch[1]->BRy = p->ULy + halfY;
    ch[2]->ULx = p->ULx;         ch[2]->URx = p->ULx + halfX; ch[2]->ULy = p->ULy + halfY; ch[2]->BRy = p->BRy;
    ch[3]->ULx = p->ULx + halfX; ch[3]->URx = p->URx;         ch[3]->ULy = p->ULy + halfY; ch[3]->BRy = p->BRy;
    for (i = 0; i < p->nkeys; i++) {
        const oracle_cand *kp = &c[p->keys[i]];
        onode *t;
        if (kp->x < (float)ch[0]->URx) t = (kp->y < (float)ch[0]->BRy) ? ch[0] : ch[2];
        else t = (kp->y < (float)ch[0]->BRy) ? ch[1] : ch[3];
        t->keys[t->nkeys++] = p->keys[i];
    }
    for (i = 0; i < 4; i++) if (ch[i]->nkeys == 1) ch[i]->bNoMore = 1;
}

typedef struct { int size; int seq; onode *node; } sizeptr;
static int sizeptr_cmp(const void *a, const void *b)
{
    const sizeptr *x = (const sizeptr *)a, *y = (const sizeptr *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* R14: later-created sorts higher */
}

/* Split `p`, push non-empty children to the front, record expandable ones. */
static void expand(olist *l, onode *p, const oracle_cand *c, sizeptr *vec, int *nvec, int *nToExpand)
{
    onode *ch[4]; int i;
    divide_node(l, p, c, ch);
    for (i = 0; i < 4; i++) {
        if (ch[i]->nkeys > 0) {
            list_push_front(l, ch[i]);
            if (ch[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = ch[i]->nkeys; vec[*nvec].seq = ch[i]->seq; vec[*nvec].node = ch[i];
                (*nvec)++;
            }
        } else { free(ch[i]->keys); free(ch[i]); }
    }
}

/* ORBextractor::DistributeOctTree, ORBextractor.cc:539-763.  Candidates are in
 * coordinates relative to (minX,minY).  Writes the index of the retained
 * candidate per leaf in final list order; returns the count, or -1 if nIni<1. */
int oracle_octree_distribute(const oracle_cand *c, int n, int minX, int maxX, int minY, int maxY,
                             int N, int *out_idx, int cap)
{
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    float hX;
    olist L; onode **ini; int i, nout = 0, bFinish = 0;
    sizeptr *vec, *prev; int nvec = 0;
    onode *lit;
    if (nIni < 1) return -1;
    hX = (float)(maxX - minX) / nIni;
    memset(&L, 0, sizeof(L));
    ini = (onode **)malloc(sizeof(onode *) * nIni);
    for (i = 0; i < nIni; i++) {
        onode *ni = node_new(&L, n);
        ni->ULx = (int)(hX * (float)i);
        ni->URx = (int)(hX * (float)(i + 1));
        ni->ULy = 0;
        ni->BRy = maxY - minY;
        list_push_back(&L, ni);
        ini[i] = ni;
    }
    for (i = 0; i < n; i++) {
        onode *t = ini[(size_t)(c[i].x / hX)];
        t->keys[t->nkeys++] = i;
    }
    free(ini);
    for (lit = L.head; lit;) {
        if (lit->nkeys == 1) { lit->bNoMore = 1; lit = lit->next; }
        else if (lit->nkeys == 0) lit = list_erase(&L, lit);
        else lit = lit->next;
    }
    vec = (sizeptr *)malloc(sizeof(sizeptr) * (4 * (size_t)(n + nIni) + 16));
    prev = (sizeptr *)malloc(sizeof(sizeptr) * (4 * (size_t)(n + nIni) + 16));
    while (!bFinish) {
        int prevSize = L.size, nToExpand = 0;
        nvec = 0;
        for (lit = L.head; lit;) {
            if (lit->bNoMore) { lit = lit->next; continue; }
            expand(&L, lit, c, vec, &nvec, &nToExpand);
            lit = list_erase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) bFinish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!bFinish) {
                int j, nprev = nvec;
                prevSize = L.size;
                memcpy(prev, vec, sizeof(sizeptr) * nprev);
                nvec = 0;
                qsort(prev, nprev, sizeof(sizeptr), sizeptr_cmp);
                for (j = nprev - 1; j >= 0; j--) {
                    expand(&L, prev[j].node, c, vec, &nvec, NULL);
                    list_erase(&L, prev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    for (lit = L.head; lit; lit = lit->next) { /* :741-760 best response, first wins */
        int best = lit->keys[0], k;
        float maxResponse = c[best].response;
        for (k = 1; k < lit->nkeys; k++)
            if (c[lit->keys[k]].response > maxResponse) { best = lit->keys[k]; maxResponse = c[best].response; }
        if (nout < cap) out_idx[nout] = best;
        nout++;
    }
    while (L.head) list_erase(&L, L.head);
    free(vec); free(prev);
    return nout;
}

/* ----------------------------------------------------------- extractor object */

/* ORBextractor::ORBextractor, ORBextractor.cc:410-470 */
oracle_orb *oracle_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    oracle_orb *o;
    int i, v, v0, vmax, vmin, sumFeatures = 0, level;
    float factor, nDesired;
    static const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    o = (oracle_orb *)calloc(1, sizeof(oracle_orb));
    o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels;
    o->iniThFAST = iniThFAST; o->minThFAST = minThFAST;
    memcpy(o->blur_taps, taps, sizeof(taps));
    o->mvScaleFactor[0] = 1.0f; o->mvLevelSigma2[0] = 1.0f;
    for (i = 1; i < nlevels; i++) {
        o->mvScaleFactor[i] = o->mvScaleFactor[i - 1] * scaleFactor;
        o->mvLevelSigma2[i] = o->mvScaleFactor[i] * o->mvScaleFactor[i];
    }
    for (i = 0; i < nlevels; i++) {
        o->mvInvScaleFactor[i] = 1.0f / o->mvScaleFactor[i];
        o->mvInvLevelSigma2[i] = 1.0f / o->mvLevelSigma2[i];
    }
    factor = 1.0f / scaleFactor;
    nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    for (level = 0; level < nlevels - 1; level++) {
        o->mnFeaturesPerLevel[level] = oracle_cvRound(nDesired);
        sumFeatures += o->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    o->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sumFeatures > 0 ? nfeatures - sumFeatures : 0;
    /* umax, :454-469 */
    vmax = cvFloorF(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    vmin = (int)ceilf(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v)
        o->umax[v] = cvRoundD(sqrt((double)(HALF_PATCH_SIZE * HALF_PATCH_SIZE) - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    return o;
}

static void free_state(oracle_orb *o)
{
    int l;
    for (l = 0; l < MAX_LEVELS; l++) {
        free(o->padded[l]); free(o->blurred[l]); free(o->cands[l]); free(o->lkps[l]);
        o->padded[l] = o->blurred[l] = NULL; o->cands[l] = NULL; o->lkps[l] = NULL;
        o->ncands[l] = o->nlkps[l] = 0;
    }
}

void oracle_orb_destroy(oracle_orb *o) { if (o) { free_state(o); free(o); } }

void oracle_orb_set_blur_taps(oracle_orb *o, const int *taps7) { memcpy(o->blur_taps, taps7, 7 * sizeof(int)); }
void oracle_orb_set_trig_variant(oracle_orb *o, int v) { o->trig_variant = v; }

/* ORBextractor::ComputePyramid, ORBextractor.cc:1115-1140 */
static void compute_pyramid(oracle_orb *o, const uint8_t *img, int w, int h, int stride)
{
    int level, y;
    for (level = 0; level < o->nlevels; ++level) {
        float scale = o->mvInvScaleFactor[level];
        int sw = oracle_cvRound((float)w * scale), sh = oracle_cvRound((float)h * scale);
        int pst = sw + 2 * EDGE_THRESHOLD;
        uint8_t *inner;
        o->lw[level] = sw; o->lh[level] = sh; o->lstride[level] = pst;
        o->padded[level] = (uint8_t *)malloc((size_t)pst * (sh + 2 * EDGE_THRESHOLD));
        inner = o->padded[level] + (size_t)EDGE_THRESHOLD * pst + EDGE_THRESHOLD;
        if (level != 0) {
            const uint8_t *pin = o->padded[level - 1] + (size_t)EDGE_THRESHOLD * o->lstride[level - 1] + EDGE_THRESHOLD;
            oracle_resize_linear(pin, o->lw[level - 1], o->lh[level - 1], o->lstride[level - 1], inner, sw, sh, pst);
        } else {
            for (y = 0; y < h; y++) memcpy(inner + (size_t)y * pst, img + (size_t)y * stride, w);
        }
        fill_border101(o->padded[level], sw, sh, pst);
    }
}

/* IC_Angle, ORBextractor.cc:77-104 */
static float ic_angle(const uint8_t *image, int step, float ptx, float pty, const int *u_max)
{
    int m_01 = 0, m_10 = 0, u, v;
    const uint8_t *center = image + (ptrdiff_t)oracle_cvRound(pty) * step + oracle_cvRound(ptx);
    for (u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = u_max[v];
        for (u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return oracle_fastAtan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, ORBextractor.cc:107-147.  `cos(angle)` / `sin(angle)` with a float argument under `using namespace std`
 * are the float overloads: trig_variant 0 calls the C library's cosf / sinf as the reference does (their values are a property of
 * the library: glibc >= 2.28 here, whose FMA and generic builds agree on [0, 2 pi] -- tools/trig/trig_variant_count.c);
 * variant 1 = the correctly rounded float of the double-precision value. */
void oracle_orb_descriptor(const uint8_t *img, int step, float ptx, float pty, float angle_deg, uint8_t *desc, int trig_variant)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factorPI;
    float a = trig_variant == 0 ? cosf(angle) : (float)cos((double)angle), b = trig_variant == 0 ? sinf(angle) : (float)sin((double)angle);
    const uint8_t *center = img + (ptrdiff_t)oracle_cvRound(pty) * step + oracle_cvRound(ptx);
    const signed char *pat = oracle_orb_pattern;
    int i, k;
    for (i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (k = 0; k < 8; k++) {
            float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1];
            float x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
            int t0 = center[oracle_cvRound(x0 * b + y0 * a) * step + oracle_cvRound(x0 * a - y0 * b)];
            int t1 = center[oracle_cvRound(x1 * b + y1 * a) * step + oracle_cvRound(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBextractor::ComputeKeyPointsOctTree, ORBextractor.cc:765-861 (one level). */
static int compute_level_keypoints(oracle_orb *o, int level)
{
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = o->lw[level] - EDGE_THRESHOLD + 3;
    const int maxBorderY = o->lh[level] - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    int wCell, hCell, i, j, ncand = 0, capc, nk, *idx, scaledPatchSize;
    const int pst = o->lstride[level];
    const uint8_t *inner = o->padded[level] + (size_t)EDGE_THRESHOLD * pst + EDGE_THRESHOLD;
    oracle_cand *cand, *cell;
    if (nCols < 1 || nRows < 1) return -1;
    wCell = (int)ceilf(width / nCols); hCell = (int)ceilf(height / nRows);
    capc = o->lw[level] * o->lh[level] / 4 + 16;
    cand = (oracle_cand *)malloc(sizeof(oracle_cand) * capc);
    cell = (oracle_cand *)malloc(sizeof(oracle_cand) * capc);
    for (i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            int y0, y1, x0, x1, nc, k;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            y0 = (int)iniY; y1 = (int)maxY; x0 = (int)iniX; x1 = (int)maxX;
            nc = oracle_fast_detect(inner + (size_t)y0 * pst + x0, x1 - x0, y1 - y0, pst, o->iniThFAST, cell, capc);
            if (nc == 0)
                nc = oracle_fast_detect(inner + (size_t)y0 * pst + x0, x1 - x0, y1 - y0, pst, o->minThFAST, cell, capc);
            for (k = 0; k < nc; k++) {
                cell[k].x += j * wCell;
                cell[k].y += i * hCell;
                cand[ncand++] = cell[k];
            }
        }
    }
    free(cell);
    o->cands[level] = cand; o->ncands[level] = ncand;
    idx = (int *)malloc(sizeof(int) * (ncand + 4));
    nk = oracle_octree_distribute(cand, ncand, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                  o->mnFeaturesPerLevel[level], idx, ncand + 4);
    if (nk < 0) { free(idx); return -1; }
    scaledPatchSize = (int)(PATCH_SIZE * o->mvScaleFactor[level]);
    o->lkps[level] = (oracle_keypoint *)malloc(sizeof(oracle_keypoint) * (nk + 1));
    o->nlkps[level] = nk;
    for (i = 0; i < nk; i++) {
        oracle_keypoint *kp = &o->lkps[level][i];
        kp->x = cand[idx[i]].x + minBorderX;
        kp->y = cand[idx[i]].y + minBorderY;
        kp->response = cand[idx[i]].response;
        kp->octave = level;
        kp->size = (float)scaledPatchSize;
        kp->class_id = -1;
        kp->angle = ic_angle(inner, pst, kp->x, kp->y, o->umax); /* computeOrientation :859-860 */
    }
    free(idx);
    return nk;
}

/* ORBextractor::operator(), ORBextractor.cc:1051-1113.
 * Returns 0 ok; -1 bad args / image too small for the cell grid (reference: UB);
 * -2 output capacity too small (n still reports the needed count). */
int oracle_orb_extract(oracle_orb *o, const uint8_t *img, int w, int h, int stride,
                       oracle_keypoint *kps, uint8_t *desc, int cap, int *n_out)
{
    int level, total = 0, offset = 0, i;
    if (!img || w <= 0 || h <= 0) { if (n_out) *n_out = 0; return 0; } /* :1054 empty image: untouched */
    free_state(o);
    /* a level narrower or lower than 56 px has no 30-px cell (nCols = (w - 26) / 30 = 0: the reference divides by it, :781-787;
     * a level of 0 or 1 px cannot even be bordered): decided before the pyramid is built */
    for (level = 0; level < o->nlevels; level++)
        if (oracle_cvRound((float)w * o->mvInvScaleFactor[level]) < 56 || oracle_cvRound((float)h * o->mvInvScaleFactor[level]) < 56) {
            if (n_out) *n_out = 0;
            return -1;
        }
    compute_pyramid(o, img, w, h, stride);
    for (level = 0; level < o->nlevels; level++)
        if (compute_level_keypoints(o, level) < 0) return -1;
    for (level = 0; level < o->nlevels; level++) total += o->nlkps[level];
    if (n_out) *n_out = total;
    if (total > cap) return -2;
    for (level = 0; level < o->nlevels; level++) {
        const int pst = o->lstride[level], lw = o->lw[level], lh = o->lh[level];
        const uint8_t *inner = o->padded[level] + (size_t)EDGE_THRESHOLD * pst + EDGE_THRESHOLD;
        int nk = o->nlkps[level];
        /* the reference blurs only levels that hold keypoints; the blurred image is
         * kept for all levels here so staged tests can compare it */
        o->blurred[level] = (uint8_t *)malloc((size_t)lw * lh);
        oracle_gauss7(inner, lw, lh, pst, o->blurred[level], lw, o->blur_taps);
        for (i = 0; i < nk; i++) {
            oracle_keypoint kp = o->lkps[level][i];
            oracle_orb_descriptor(o->blurred[level], lw, kp.x, kp.y, kp.angle, desc + (size_t)(offset + i) * 32, o->trig_variant);
            if (level != 0) { float scale = o->mvScaleFactor[level]; kp.x *= scale; kp.y *= scale; }
            kps[offset + i] = kp;
        }
        offset += nk;
    }
    return 0;
}

/* ------------------------------------------------------------ staged accessors */
int oracle_orb_nlevels(const oracle_orb *o) { return o->nlevels; }
int oracle_orb_features_per_level(const oracle_orb *o, int level) { return o->mnFeaturesPerLevel[level]; }
float oracle_orb_scale_factor(const oracle_orb *o, int level) { return o->mvScaleFactor[level]; }
const int *oracle_orb_umax(const oracle_orb *o) { return o->umax; }
int oracle_orb_level_dims(const oracle_orb *o, int level, int *w, int *h, int *stride)
{ *w = o->lw[level]; *h = o->lh[level]; *stride = o->lstride[level]; return 0; }
const uint8_t *oracle_orb_level_padded(const oracle_orb *o, int level) { return o->padded[level]; }
const uint8_t *oracle_orb_level_blurred(const oracle_orb *o, int level) { return o->blurred[level]; }
int oracle_orb_level_ncands(const oracle_orb *o, int level) { return o->ncands[level]; }
const oracle_cand *oracle_orb_level_cands(const oracle_orb *o, int level) { return o->cands[level]; }
int oracle_orb_level_nkps(const oracle_orb *o, int level) { return o->nlkps[level]; }
const oracle_keypoint *oracle_orb_level_kps(const oracle_orb *o, int level) { return o->lkps[level]; }
