/*
 * oracle/stereo_oracle.c -- CPU restatement of Frame::ComputeStereoMatches.
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.c header).  PARITY UNPINNED.
 * Follows src/Frame.cc:527-701 literally: row table of right keypoints (band
 * +-2*scale, :537-554), per left keypoint the best Hamming match among same-row
 * candidates within +-1 octave and the disparity range (:565-610), 11x11 / 11-shift
 * L1 correlation on the pyramid level of the left keypoint (:612-653), parabola
 * sub-pixel fit (:659-666), depth (:669-683), median cut 1.5*1.4*median (:687-700).
 * The L1 norms are sums of integer-valued floats, hence exact.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float x, y, size, angle, response;
    int octave, class_id;
} okp;

int oracle_descriptor_distance(const uint8_t *a, const uint8_t *b);

typedef struct { int dist, idx; } distidx;
static int distidx_cmp(const void *a, const void *b)
{
    const distidx *x = (const distidx *)a, *y = (const distidx *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* pyrL/pyrR: per level pointer to pixel (0,0) of mvImagePyramid[level]; strides, widths per level. */
int oracle_stereo_matches(const okp *kL, const uint8_t *dL, int N, const okp *kR, const uint8_t *dR, int Nr,
                          const uint8_t *const *pyrL, const uint8_t *const *pyrR, const int *stride, const int *cols,
                          int nRows, const float *scaleFactors, const float *invScaleFactors, float mb, float mbf,
                          float *mvuRight, float *mvDepth)
{
    const int TH_HIGH = 95, TH_LOW = 45;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    int **rows = (int **)calloc(nRows, sizeof(int *));
    int *rown = (int *)calloc(nRows, sizeof(int)), *rowc = (int *)calloc(nRows, sizeof(int));
    distidx *vDistIdx = (distidx *)malloc(sizeof(distidx) * (N > 0 ? N : 1));
    int nd = 0, iL, iR, yi, i;
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    for (iL = 0; iL < N; iL++) { mvuRight[iL] = -1.0f; mvDepth[iL] = -1.0f; }
    for (iR = 0; iR < Nr; iR++) {
        const float kpY = kR[iR].y, r = 2.0f * scaleFactors[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (yi = minr; yi <= maxr; yi++) {
            if (yi < 0 || yi >= nRows) continue; /* reference: out-of-range write (UB) */
            if (rown[yi] == rowc[yi]) { rowc[yi] = rowc[yi] ? 2 * rowc[yi] : 16; rows[yi] = (int *)realloc(rows[yi], sizeof(int) * rowc[yi]); }
            rows[yi][rown[yi]++] = iR;
        }
    }
    for (iL = 0; iL < N; iL++) {
        const int levelL = kL[iL].octave;
        const float vL = kL[iL].y, uL = kL[iL].x;
        const int row = (int)vL;
        const float minU = uL - maxD, maxU = uL - minD;
        int bestDist = TH_HIGH, bestIdxR = 0, iC;
        if (row < 0 || row >= nRows || rown[row] == 0) continue;
        if (maxU < 0) continue;
        for (iC = 0; iC < rown[row]; iC++) {
            const int c = rows[row][iC];
            float uR;
            if (kR[c].octave < levelL - 1 || kR[c].octave > levelL + 1) continue;
            uR = kR[c].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = oracle_descriptor_distance(dL + 32 * (size_t)iL, dR + 32 * (size_t)c);
                if (dist < bestDist) { bestDist = dist; bestIdxR = c; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float scaleFactor = invScaleFactors[levelL];
            const float scaleduL = roundf(kL[iL].x * scaleFactor);
            const float scaledvL = roundf(kL[iL].y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const uint8_t *PL = pyrL[levelL], *PR = pyrR[levelL];
            const int st = stride[levelL];
            const int cL = PL[(int)scaledvL * st + (int)scaleduL];
            int sadBest = INT_MAX, bestincR = 0, incR, dy, dx;
            float vDists[11];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            float dist1, dist2, dist3, deltaR, bestuR, disparity;
            if (iniu < 0 || endu >= cols[levelL]) continue;
            for (incR = -L; incR <= +L; incR++) {
                const int cR = PR[(int)scaledvL * st + (int)scaleduR0 + incR];
                float dist = 0;
                for (dy = -w; dy <= w; dy++)
                    for (dx = -w; dx <= w; dx++) {
                        const float a = (float)PL[((int)scaledvL + dy) * st + (int)scaleduL + dx] - (float)cL;
                        const float b = (float)PR[((int)scaledvL + dy) * st + (int)scaleduR0 + incR + dx] - (float)cR;
                        dist += fabsf(a - b);
                    }
                if (dist < sadBest) { sadBest = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            dist1 = vDists[L + bestincR - 1]; dist2 = vDists[L + bestincR]; dist3 = vDists[L + bestincR + 1];
            deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            bestuR = scaleFactors[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx[nd].dist = sadBest; vDistIdx[nd].idx = iL; nd++;
            }
        }
    }
    if (nd > 0) { /* reference indexes vDistIdx[size/2] unconditionally (UB when empty) */
        float median, thDist;
        qsort(vDistIdx, nd, sizeof(distidx), distidx_cmp);
        median = (float)vDistIdx[nd / 2].dist;
        thDist = 1.5f * 1.4f * median;
        for (i = nd - 1; i >= 0; i--) {
            if (vDistIdx[i].dist < thDist) break;
            mvuRight[vDistIdx[i].idx] = -1;
            mvDepth[vDistIdx[i].idx] = -1;
        }
    }
    for (i = 0; i < nRows; i++) free(rows[i]);
    free(rows); free(rown); free(rowc); free(vDistIdx);
    return nd;
}
