/*
 * oracle/match_oracle.c -- CPU restatement of the ORBmatcher data plane.
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.c header).  PARITY UNPINNED: the
 * reference holds no tests/golden vectors; pinned by known-answer tests only.
 *
 * Follows src/ORBmatcher.cc: DescriptorDistance :1848-1864 (SWAR popcount, literal),
 * the best / second-best selection loop common to every Search* function
 * (e.g. :645-672, :96-118: strict '<' so the first-seen candidate wins ties,
 * second updated with 'else if'), the TH_LOW / nnratio acceptance test
 * (:674-676: bestDist<=TH_LOW && bestDist<(float)bestDist2*mfNNratio),
 * ComputeThreeMaxima :1802-1843, and Frame::GetFeaturesInArea (src/Frame.cc:342-395)
 * with AssignFeaturesToGrid/PosInGrid (:245-260,:397-407).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HISTO_LENGTH 30 /* ORBmatcher.cc:40 */

/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1848-1864 */
int oracle_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0, i;
    for (i = 0; i < 8; i++) {
        uint32_t pa, pb, v;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void oracle_hamming_matrix(const uint8_t *A, int nA, const uint8_t *B, int nB, uint16_t *out)
{
    int i, j;
    for (i = 0; i < nA; i++)
        for (j = 0; j < nB; j++)
            out[(size_t)i * nB + j] = (uint16_t)oracle_descriptor_distance(A + 32 * (size_t)i, B + 32 * (size_t)j);
}

/* Per query row: best, second-best and arg-best over ALL of B in index order
 * (SURVEY 8d config-2 task).  Empty B: best=second=INT_MAX, idx=-1. */
void oracle_match_bruteforce(const uint8_t *A, int nA, const uint8_t *B, int nB,
                             int *best, int *second, int *idx)
{
    int i, j;
    for (i = 0; i < nA; i++) {
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
        for (j = 0; j < nB; j++) {
            int dist = oracle_descriptor_distance(A + 32 * (size_t)i, B + 32 * (size_t)j);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best[i] = bestDist; second[i] = bestDist2; idx[i] = bestIdx;
    }
}

/* Same selection over a gated candidate list per query (CSR: cand_off[nA+1],
 * cand_idx[] in GetFeaturesInArea / BoW-member order). */
void oracle_match_candidates(const uint8_t *A, int nA, const uint8_t *B,
                             const int *cand_off, const int *cand_idx,
                             int *best, int *second, int *idx)
{
    int i, k;
    for (i = 0; i < nA; i++) {
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
        for (k = cand_off[i]; k < cand_off[i + 1]; k++) {
            int j = cand_idx[k];
            int dist = oracle_descriptor_distance(A + 32 * (size_t)i, B + 32 * (size_t)j);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best[i] = bestDist; second[i] = bestDist2; idx[i] = bestIdx;
    }
}

/* Acceptance test: bestDist<=th && bestDist<(float)bestDist2*nnratio.
 * match12[i] = idx or -1; returns the number of accepted rows. */
int oracle_match_filter(int nA, const int *best, const int *second, const int *idx,
                        int th, float nnratio, int *match12)
{
    int i, n = 0;
    for (i = 0; i < nA; i++) {
        match12[i] = -1;
        if (idx[i] >= 0 && best[i] <= th && (float)best[i] < (float)second[i] * nnratio) {
            match12[i] = idx[i];
            n++;
        }
    }
    return n;
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1802-1843 (histogram sizes in). */
void oracle_three_maxima(const int *histo_size, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0, i;
    *ind1 = *ind2 = *ind3 = -1;
    for (i = 0; i < L; i++) {
        const int s = histo_size[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ---- Frame grid: AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea ------- */
#define FRAME_GRID_ROWS 48 /* include/Frame.h:37 */
#define FRAME_GRID_COLS 64 /* include/Frame.h:38 */

typedef struct {
    float minX, minY, maxX, maxY, invW, invH;
    int *cell_off;   /* [COLS*ROWS+1], cells indexed col*ROWS+row */
    int *cell_items; /* keypoint indices, insertion order inside a cell */
    int n;
} oracle_grid;

/* Frame::PosInGrid, Frame.cc:397-407 */
static int pos_in_grid(const oracle_grid *g, float x, float y, int *px, int *py)
{
    *px = (int)roundf((x - g->minX) * g->invW);
    *py = (int)roundf((y - g->minY) * g->invH);
    if (*px < 0 || *px >= FRAME_GRID_COLS || *py < 0 || *py >= FRAME_GRID_ROWS) return 0;
    return 1;
}

/* Frame::AssignFeaturesToGrid, Frame.cc:245-260; bounds as Frame ctor sets them
 * (mfGridElementWidthInv = COLS/(maxX-minX), Frame.cc:221-222). */
oracle_grid *oracle_grid_build(const float *xy, int n, float minX, float minY, float maxX, float maxY)
{
    oracle_grid *g = (oracle_grid *)calloc(1, sizeof(oracle_grid));
    const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    int i, *cnt, *cell;
    g->minX = minX; g->minY = minY; g->maxX = maxX; g->maxY = maxY; g->n = n;
    g->invW = (float)FRAME_GRID_COLS / (maxX - minX);
    g->invH = (float)FRAME_GRID_ROWS / (maxY - minY);
    g->cell_off = (int *)calloc(NC + 1, sizeof(int));
    g->cell_items = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    cell = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    cnt = (int *)calloc(NC, sizeof(int));
    for (i = 0; i < n; i++) {
        int px, py;
        cell[i] = pos_in_grid(g, xy[2 * i], xy[2 * i + 1], &px, &py) ? px * FRAME_GRID_ROWS + py : -1;
        if (cell[i] >= 0) g->cell_off[cell[i] + 1]++;
    }
    for (i = 0; i < NC; i++) g->cell_off[i + 1] += g->cell_off[i];
    for (i = 0; i < n; i++)
        if (cell[i] >= 0) g->cell_items[g->cell_off[cell[i]] + cnt[cell[i]]++] = i;
    free(cnt); free(cell);
    return g;
}
/* mGrid as flat tables: cell_off[COLS*ROWS + 1] and the keypoint indices cell by cell (col * ROWS + row), insertion order inside a cell */
void oracle_grid_tables(const oracle_grid *g, int *cell_off, int *items)
{
    const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    memcpy(cell_off, g->cell_off, sizeof(int) * (NC + 1));
    if (g->cell_off[NC]) memcpy(items, g->cell_items, sizeof(int) * g->cell_off[NC]);
}
void oracle_grid_free(oracle_grid *g) { if (g) { free(g->cell_off); free(g->cell_items); free(g); } }

/* Frame::GetFeaturesInArea, Frame.cc:342-395.  Result order: column-major
 * cells, insertion order inside a cell.  Returns the count (<= cap written). */
int oracle_grid_features_in_area(const oracle_grid *g, const float *xy, const int *octave,
                                 float x, float y, float r, int minLevel, int maxLevel,
                                 int *out, int cap)
{
    int n = 0, ix, iy, k;
    const int nMinCellX = (int)fmaxf(0.f, floorf((x - g->minX - r) * g->invW));
    int nMaxCellX, nMinCellY, nMaxCellY, bCheckLevels;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    nMaxCellX = (int)fminf((float)FRAME_GRID_COLS - 1, ceilf((x - g->minX + r) * g->invW));
    if (nMaxCellX < 0) return 0;
    nMinCellY = (int)fmaxf(0.f, floorf((y - g->minY - r) * g->invH));
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    nMaxCellY = (int)fminf((float)FRAME_GRID_ROWS - 1, ceilf((y - g->minY + r) * g->invH));
    if (nMaxCellY < 0) return 0;
    bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * FRAME_GRID_ROWS + iy;
            for (k = g->cell_off[c]; k < g->cell_off[c + 1]; k++) {
                const int j = g->cell_items[k];
                float distx, disty;
                if (bCheckLevels) {
                    if (octave[j] < minLevel) continue;
                    if (maxLevel >= 0 && octave[j] > maxLevel) continue;
                }
                distx = xy[2 * j] - x; disty = xy[2 * j + 1] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = j; n++; }
            }
        }
    return n;
}

/* ---- SearchForTriangulation inner loop, ORBmatcher.cc:892-990 + CheckDistEpipolarLine
 * :341-358.  The BoW-node co-iteration (host, DBoW2) is given as per-query candidate
 * lists in member order.  Note vbMatched2 is never set in the reference, so queries
 * are independent; `dist>bestDist` (non-strict) lets a later equal candidate win. */
typedef struct { float x, y, size, angle, response; int octave, class_id; } tri_kp;

static int check_dist_epipolar_line(const tri_kp *kp1, const tri_kp *kp2, const float *F12, const float *levelSigma2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    float dsqr;
    if (den == 0) return 0;
    dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[kp2->octave];
}

void oracle_match_triangulation(const tri_kp *kps1, const uint8_t *d1, int n1, const tri_kp *kps2, const uint8_t *d2,
                                const int *cand_off, const int *cand_idx, const uint8_t *hasmp1, const uint8_t *hasmp2,
                                const uint8_t *stereo1, const uint8_t *stereo2, int bOnlyStereo, const float *F12,
                                float ex, float ey, const float *scaleFactors2, const float *levelSigma2,
                                int *match12, int *bestdist)
{
    const int TH_LOW = 45;
    int i, k;
    for (i = 0; i < n1; i++) {
        int bestDist = TH_LOW, bestIdx2 = -1;
        match12[i] = -1; bestdist[i] = TH_LOW;
        if (hasmp1[i]) continue;
        if (bOnlyStereo && !stereo1[i]) continue;
        for (k = cand_off[i]; k < cand_off[i + 1]; k++) {
            const int idx2 = cand_idx[k];
            int dist;
            if (hasmp2[idx2]) continue;
            if (bOnlyStereo && !stereo2[idx2]) continue;
            dist = oracle_descriptor_distance(d1 + 32 * (size_t)i, d2 + 32 * (size_t)idx2);
            if (dist > TH_LOW || dist > bestDist) continue;
            if (!stereo1[i] && !stereo2[idx2]) {
                const float distex = ex - kps2[idx2].x, distey = ey - kps2[idx2].y;
                if (distex * distex + distey * distey < 100 * scaleFactors2[kps2[idx2].octave]) continue;
            }
            if (check_dist_epipolar_line(&kps1[i], &kps2[idx2], F12, levelSigma2)) { bestIdx2 = idx2; bestDist = dist; }
        }
        match12[i] = bestIdx2; bestdist[i] = bestDist;
    }
}

/* ---- ORBmatcher::SearchForTriangulation as a whole, ORBmatcher.cc:858-1024: FeatureVector co-iteration (:889-985;
 * std::map iteration with lower_bound = a merge of the two ascending node lists), the inner loop above per member
 * of a shared node, the rotation histogram (:965-976), ComputeThreeMaxima and the rejection of the other bins
 * (:992-1011).  FeatureVectors as (nodes ascending, off, items in member order).  match12[n1] = index in KF2 or -1
 * (= vMatchedPairs); returns nmatches. */
int oracle_search_for_triangulation(const tri_kp *kps1, const uint8_t *d1, int n1, const tri_kp *kps2, const uint8_t *d2,
                                    const int *nodes1, const int *off1, const int *items1, int nn1, const int *nodes2,
                                    const int *off2, const int *items2, int nn2, const uint8_t *hasmp1,
                                    const uint8_t *hasmp2, const uint8_t *stereo1, const uint8_t *stereo2, int bOnlyStereo,
                                    const float *F12, float ex, float ey, const float *scaleFactors2,
                                    const float *levelSigma2, int check_orientation, int *match12)
{
    const int TH_LOW = 45;
    const float factor = 1.0f / HISTO_LENGTH;
    int *bin_of = (int *)malloc(sizeof(int) * (n1 + 1));
    int hist[HISTO_LENGTH] = {0};
    int a = 0, b = 0, i, nmatches = 0;
    for (i = 0; i < n1; i++) { match12[i] = -1; bin_of[i] = -1; }
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            int i1, i2;
            for (i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const int idx1 = items1[i1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                if (hasmp1[idx1]) continue;
                if (bOnlyStereo && !stereo1[idx1]) continue;
                for (i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const int idx2 = items2[i2];
                    int dist;
                    if (hasmp2[idx2]) continue;       /* vbMatched2 is never set in the reference */
                    if (bOnlyStereo && !stereo2[idx2]) continue;
                    dist = oracle_descriptor_distance(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!stereo1[idx1] && !stereo2[idx2]) {
                        const float distex = ex - kps2[idx2].x, distey = ey - kps2[idx2].y;
                        if (distex * distex + distey * distey < 100 * scaleFactors2[kps2[idx2].octave]) continue;
                    }
                    if (check_dist_epipolar_line(&kps1[idx1], &kps2[idx2], F12, levelSigma2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kps1[idx1].angle - kps2[bestIdx2].angle;
                        int bin;
                        if (rot < 0.0) rot += 360.0f;
                        bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        bin_of[idx1] = bin;
                        hist[bin]++;
                    }
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) a++; /* lower_bound(f2it->first) on an ascending list */
        else b++;
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { match12[i] = -1; nmatches--; }
    }
    free(bin_of);
    return nmatches;
}

/* ---- windowed search: GetFeaturesInArea + the best/second-with-levels loop of
 * SearchByProjection(Frame&, vector<MapPoint*>&, th), ORBmatcher.cc:69-118.
 * q = {u, v, r, xr, minLevel, maxLevel}; skip[idx] stands for "already holds an
 * observed MapPoint" (:87-89); uright may be NULL (mono), else the stereo check of
 * :91-96 applies.  init_dist = 256 (:79-81) or INT_MAX (SearchForInitialization). */
typedef struct { float u, v, r, xr; int min_level, max_level; } oracle_wquery;

void oracle_search_window(const oracle_grid *g, const float *xy, const int *octave, const uint8_t *desc,
                          const uint8_t *skip, const float *uright, const oracle_wquery *q, const uint8_t *qdesc, int nq,
                          int init_dist, int *best, int *best_level, int *second, int *second_level, int *idx)
{
    int *cand = (int *)malloc(sizeof(int) * (g->n + 1));
    int i, k;
    for (i = 0; i < nq; i++) {
        const int nc = oracle_grid_features_in_area(g, xy, octave, q[i].u, q[i].v, q[i].r, q[i].min_level, q[i].max_level, cand, g->n + 1);
        int bestDist = init_dist, bestLevel = -1, bestDist2 = init_dist, bestLevel2 = -1, bestIdx = -1;
        for (k = 0; k < nc; k++) {
            const int j = cand[k];
            int dist;
            if (skip && skip[j]) continue;
            if (uright && uright[j] > 0) {
                const float er = fabsf(q[i].xr - uright[j]);
                if (er > q[i].r) continue;
            }
            dist = oracle_descriptor_distance(qdesc + 32 * (size_t)i, desc + 32 * (size_t)j);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = octave[j]; bestIdx = j; }
            else if (dist < bestDist2) { bestLevel2 = octave[j]; bestDist2 = dist; }
        }
        best[i] = bestDist; best_level[i] = bestLevel; second[i] = bestDist2; second_level[i] = bestLevel2; idx[i] = bestIdx;
    }
    free(cand);
}

/* ---- Whole search loops, with the reference's in-loop bookkeeping (the coupling between
 * queries that oracle_search_window leaves out) and the rotation-consistency check.
 *
 * Rotation histogram as every Search* builds it (e.g. ORBmatcher.cc:1642-1650):
 * rot = angle1 - angle2 (+360 if negative), bin = round(rot * 1/HISTO_LENGTH), 30 -> 0.
 * (With factor = 1/30 and 30 bins only bins 0..12 are ever hit: SURVEY M12.) */
static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    int bin;
    if (rot < 0.0) rot += 360.0f;
    bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* The SearchByProjection family over already projected queries:
 *   (Frame, vector<MapPoint*>, th)            ORBmatcher.cc:46-132    th_accept = TH_HIGH, ratio_same_level = 1, no rotation check
 *   (CurrentFrame, LastFrame, th, mono)       :1529-1671              TH_HIGH, best only, rotation check
 *   (CurrentFrame, KeyFrame, found, th, dist) :1673-1800              ORBdist, best only, rotation check
 *   (KeyFrame, Scw, points, matched, th)      :491-604                TH_LOW,  best only, no rotation check
 * Query i = one map point that passed the caller's projection / frustum tests (in the
 * caller's loop order): window (u, v, r), level range, xr = u - bf/z for the stereo check,
 * its descriptor, the angle of its source keypoint and takes[i] = whether the pointer it
 * leaves in mvpMapPoints[best] makes later queries skip that keypoint (Observations()>0 for
 * the first two forms, always for the last two).  occupied[j] = keypoint j is skipped from
 * the start.  Outputs: match_kp[j] = query whose point ends in slot j, -1 = slot untouched,
 * -2 = set to NULL by the rotation check; match_q[i] = keypoint chosen by query i when its
 * match was accepted (before the rotation check), else -1.  Returns nmatches as counted
 * by the reference (:124-127, :1634-1637 ++, :1664 --). */
int oracle_search_projection_seq(const oracle_grid *g, const float *xy, const int *octave, const float *angle,
                                 const uint8_t *desc, const uint8_t *occupied, const float *uright,
                                 const oracle_wquery *q, const uint8_t *qdesc, const float *qangle, const uint8_t *qtakes,
                                 int nq, int th_accept, float nnratio, int ratio_same_level, int check_orientation,
                                 int *match_kp, int *match_q)
{
    const int n = g->n;
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    uint8_t *blocked = (uint8_t *)malloc(n + 1);
    int *qbin = (int *)malloc(sizeof(int) * (nq + 1));
    int hist[HISTO_LENGTH] = {0};
    int i, k, nmatches = 0;
    for (k = 0; k < n; k++) { blocked[k] = occupied ? occupied[k] : 0; match_kp[k] = -1; }
    for (i = 0; i < nq; i++) {
        const int nc = oracle_grid_features_in_area(g, xy, octave, q[i].u, q[i].v, q[i].r, q[i].min_level, q[i].max_level, cand, n + 1);
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        match_q[i] = -1; qbin[i] = -1;
        for (k = 0; k < nc; k++) {
            const int j = cand[k];
            int dist;
            if (blocked[j]) continue;
            if (uright && uright[j] > 0) {
                const float er = fabsf(q[i].xr - uright[j]);
                if (er > q[i].r) continue;
            }
            dist = oracle_descriptor_distance(qdesc + 32 * (size_t)i, desc + 32 * (size_t)j);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = octave[j]; bestIdx = j; }
            else if (dist < bestDist2) { bestLevel2 = octave[j]; bestDist2 = dist; }
        }
        if (bestDist <= th_accept) {
            if (ratio_same_level && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue; /* :121-122 */
            match_kp[bestIdx] = i; /* F.mvpMapPoints[bestIdx] = pMP */
            match_q[i] = bestIdx;
            if (qtakes[i]) blocked[bestIdx] = 1;
            nmatches++;
            if (check_orientation) { qbin[i] = rot_bin(qangle[i], angle[bestIdx]); hist[qbin[i]]++; }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i = 0; i < nq; i++) /* rotHist holds keypoint indices: every entry of a rejected bin clears its slot */
            if (qbin[i] >= 0 && qbin[i] != ind1 && qbin[i] != ind2 && qbin[i] != ind3) { match_kp[match_q[i]] = -2; nmatches--; }
    }
    free(cand); free(blocked); free(qbin);
    return nmatches;
}

/* ORBmatcher::SearchForInitialization, ORBmatcher.cc:606-721.  prev[i1] = vbPrevMatched[i1]
 * (in/out, updated at :715-718), kps1/kps2 = mvKeysUn of F1/F2 (x, y, octave, angle),
 * g2 = grid of F2.  vnMatches12[n1] out.  Returns nmatches. */
int oracle_search_for_initialization(const float *xy1, const int *octave1, const float *angle1, const uint8_t *desc1, int n1,
                                     const oracle_grid *g2, const float *xy2, const int *octave2, const float *angle2,
                                     const uint8_t *desc2, float *prev, int windowSize, float nnratio, int check_orientation,
                                     int *vnMatches12)
{
    const int n2 = g2->n, TH_LOW = 45; /* ORBmatcher.cc:38 */
    int *cand = (int *)malloc(sizeof(int) * (n2 + 1));
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (n2 + 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (n2 + 1));
    int *qbin = (int *)malloc(sizeof(int) * (n1 + 1));
    int hist[HISTO_LENGTH] = {0};
    int i1, k, nmatches = 0;
    (void)xy1;
    for (k = 0; k < n2; k++) { vMatchedDistance[k] = INT_MAX; vnMatches21[k] = -1; }
    for (i1 = 0; i1 < n1; i1++) { vnMatches12[i1] = -1; qbin[i1] = -1; }
    for (i1 = 0; i1 < n1; i1++) {
        int nc, bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        const int level1 = octave1[i1];
        if (level1 > 0) continue;
        nc = oracle_grid_features_in_area(g2, xy2, octave2, prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1, cand, n2 + 1);
        for (k = 0; k < nc; k++) {
            const int i2 = cand[k];
            const int dist = oracle_descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) { qbin[i1] = rot_bin(angle1[i1], angle2[bestIdx2]); hist[qbin[i1]]++; }
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i1 = 0; i1 < n1; i1++)
            if (qbin[i1] >= 0 && qbin[i1] != ind1 && qbin[i1] != ind2 && qbin[i1] != ind3 && vnMatches12[i1] >= 0) {
                vnMatches12[i1] = -1; nmatches--;
            }
    }
    for (i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = xy2[2 * vnMatches12[i1]]; prev[2 * i1 + 1] = xy2[2 * vnMatches12[i1] + 1]; }
    free(cand); free(vMatchedDistance); free(vnMatches21); free(qbin);
    return nmatches;
}

/* ORBmatcher::SearchByBoW, both forms: (KeyFrame*, Frame&, matches) ORBmatcher.cc:360-489
 * (kf_kf = 0) and (KeyFrame*, KeyFrame*, matches12) :723-856 (kf_kf = 1).  A
 * DBoW2::FeatureVector is given as its std::map in key order: nodes[nn] ascending,
 * members of node k = items[off[k] .. off[k+1]) in insertion (= feature) order.
 * valid1[i] = "feature i of the first keyframe owns a good MapPoint" (:395-399, :763-767);
 * valid2 (kf_kf only) the same for the second (:782-786).  match12[n1] = feature of
 * the second set or -1; match21[n2] = feature of the first or -1.  kf_kf = 0 accepts
 * bestDist1 <= TH_LOW (:429), kf_kf = 1 bestDist1 < TH_LOW (:799).  Returns nmatches. */
int oracle_search_by_bow(int kf_kf, const int *nodes1, const int *off1, const int *items1, int nn1, const uint8_t *valid1,
                         const uint8_t *desc1, const float *angle1, int n1, const int *nodes2, const int *off2, const int *items2,
                         int nn2, const uint8_t *valid2, const uint8_t *desc2, const float *angle2, int n2, float nnratio,
                         int check_orientation, int *match12, int *match21)
{
    const int TH_LOW = 45; /* ORBmatcher.cc:38 */
    int hist[HISTO_LENGTH] = {0};
    int *bin1 = (int *)malloc(sizeof(int) * (n1 + 1));
    int a = 0, b = 0, i, nmatches = 0;
    for (i = 0; i < n1; i++) { match12[i] = -1; bin1[i] = -1; }
    for (i = 0; i < n2; i++) match21[i] = -1;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            int k1, k2;
            for (k1 = off1[a]; k1 < off1[a + 1]; k1++) {
                const int idx1 = items1[k1];
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                if (!valid1[idx1]) continue;
                for (k2 = off2[b]; k2 < off2[b + 1]; k2++) {
                    const int idx2 = items2[k2];
                    int dist;
                    if (match21[idx2] >= 0) continue;          /* vpMapPointMatches[realIdxF] / vbMatched2[idx2] */
                    if (kf_kf && !valid2[idx2]) continue;
                    dist = oracle_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2; match21[bestIdx2] = idx1;
                        if (check_orientation) { bin1[idx1] = rot_bin(angle1[idx1], angle2[bestIdx2]); hist[bin1[idx1]]++; }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;          /* lower_bound */
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i = 0; i < n1; i++)
            if (bin1[i] >= 0 && bin1[i] != ind1 && bin1[i] != ind2 && bin1[i] != ind3) {
                match21[match12[i]] = -1; match12[i] = -1; nmatches--;
            }
    }
    free(bin1);
    return nmatches;
}

/* Candidate loop of ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th), ORBmatcher.cc:1092-1146 (the same loop
 * at :1245-1276 in the Sim3 form, there without the reprojection gate): keypoints of the window whose level is in
 * [l-1, l]; reprojection error gate e2 * invLevelSigma2[level] > 7.8 (stereo keypoint, mvuRight >= 0: ex, ey, er)
 * or > 5.99 (mono: ex, ey); best distance from 256, strict '<'.  q = {u, v, radius, ur, l-1, l}.  The map update
 * that follows (:1149-1170) reads only bestDist / bestIdx, so the points are independent. */
void oracle_search_fuse(const oracle_grid *g, const float *xy, const int *octave, const uint8_t *desc, const float *uright,
                        const float *invLevelSigma2, int use_gate, const oracle_wquery *q, const uint8_t *qdesc, int nq,
                        int *best, int *idx)
{
    int *cand = (int *)malloc(sizeof(int) * (g->n + 1));
    int i, k;
    for (i = 0; i < nq; i++) {
        /* KeyFrame::GetFeaturesInArea(u, v, radius) has no level filter; the loop applies it (:1106-1107) */
        const int nc = oracle_grid_features_in_area(g, xy, octave, q[i].u, q[i].v, q[i].r, -1, -1, cand, g->n + 1);
        int bestDist = 256, bestIdx = -1;
        for (k = 0; k < nc; k++) {
            const int j = cand[k];
            const int kpLevel = octave[j];
            int dist;
            if (kpLevel < q[i].min_level || kpLevel > q[i].max_level) continue;
            if (use_gate) {
                const float kpx = xy[2 * j], kpy = xy[2 * j + 1];
                const float ex = q[i].u - kpx, ey = q[i].v - kpy;
                if (uright && uright[j] >= 0) {
                    const float er = q[i].xr - uright[j];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
                }
            }
            dist = oracle_descriptor_distance(qdesc + 32 * (size_t)i, desc + 32 * (size_t)j);
            if (dist < bestDist) { bestDist = dist; bestIdx = j; }
        }
        best[i] = bestDist; idx[i] = bestIdx;
    }
    free(cand);
}

/* ---- the fork's whole-map SearchByProjection(Frame&, Map*, Rcw, tcw, ...),
 * ORBmatcher.cc:134-222, with isInFrustum :262-330, ComputeDistance :224-260 and
 * RadiusByViewingCos :332-338, literally (mixed float / double arithmetic kept).
 * F.mvpMapPoints is only read, so map points are independent; a keypoint chosen by
 * several map points keeps the last one (vMatchedMPs[bestIdx] = pMP). */
typedef struct { float fx, fy, cx, cy; int bminx, bmaxx, bminy, bmaxy; float gminx, gminy, gmaxx, gmaxy; } oracle_cam;

static int oracle_in_frustum(const float *P, const float *Pn, float minDistance, float maxDistance,
                             const oracle_cam *cam, double mRcw[3][3], double mtcw[3], float viewingCosLimit,
                             const float *scaleFactors, int nLevels, float *pu, float *pv, int *plevel, float *pcos)
{
    float ptX = P[0], ptY = P[1], ptZ = P[2];
    double Pt[3], Rt[3][3], Rtt[3], PO[3], normSum, norm;
    float PcX, PcY, PcZ, invz, u, v, dist, viewCos, ratio;
    int i, nPredictedLevel;
    Pt[0] = ptX; Pt[1] = ptY; Pt[2] = ptZ;
    PcX = mRcw[0][0] * ptX + mRcw[0][1] * ptY + mRcw[0][2] * ptZ + mtcw[0];
    PcY = mRcw[1][0] * ptX + mRcw[1][1] * ptY + mRcw[1][2] * ptZ + mtcw[1];
    PcZ = mRcw[2][0] * ptX + mRcw[2][1] * ptY + mRcw[2][2] * ptZ + mtcw[2];
    if (PcZ < 0.0) return 0;
    invz = 1.0 / PcZ;
    u = cam->fx * PcX * invz + cam->cx;
    v = cam->fy * PcY * invz + cam->cy;
    if (u < cam->bminx || u > cam->bmaxx) return 0;
    if (v < cam->bminy || v > cam->bmaxy) return 0;
    /* ComputeDistance */
    Rt[0][0] = (-1) * mRcw[0][0]; Rt[1][0] = (-1) * mRcw[0][1]; Rt[2][0] = (-1) * mRcw[0][2];
    Rt[0][1] = (-1) * mRcw[1][0]; Rt[1][1] = (-1) * mRcw[1][1]; Rt[2][1] = (-1) * mRcw[1][2];
    Rt[0][2] = (-1) * mRcw[2][0]; Rt[1][2] = (-1) * mRcw[2][1]; Rt[2][2] = (-1) * mRcw[2][2];
    for (i = 0; i < 3; i++) Rtt[i] = Rt[i][0] * mtcw[0] + Rt[i][1] * mtcw[1] + Rt[i][2] * mtcw[2];
    for (i = 0; i < 3; i++) PO[i] = Pt[i] - Rtt[i];
    normSum = PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2];
    norm = sqrt(normSum);
    dist = norm;
    if (dist < (0.9 * minDistance) || dist > (maxDistance / 0.9)) return 0;
    viewCos = PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2];
    viewCos = viewCos / dist;
    if (viewCos < viewingCosLimit) return 0;
    ratio = dist / minDistance;
    for (nPredictedLevel = 0; nPredictedLevel < nLevels && scaleFactors[nPredictedLevel] < ratio; nPredictedLevel++) {} /* lower_bound */
    if (nPredictedLevel >= nLevels) nPredictedLevel = nLevels - 1;
    *pu = u; *pv = v; *plevel = nPredictedLevel; *pcos = viewCos;
    return 1;
}

int oracle_search_by_projection_map(const float *kxy, const int *koct, const uint8_t *kdesc, int n, const uint8_t *has_mp,
                                    const float *mp_pos, const float *mp_normal, const float *mp_mind, const float *mp_maxd,
                                    const uint8_t *mp_desc, int m, const double *Rcw, const double *tcw, const oracle_cam *cam,
                                    const float *scaleFactors, int nLevels, float th, float nnratio, int th_reloc,
                                    int *matched_mp, float *proj /* [m][4] u,v,viewCos,level (level<0: not in frustum) */)
{
    oracle_grid *g = oracle_grid_build(kxy, n, cam->gminx, cam->gminy, cam->gmaxx, cam->gmaxy);
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    double R[3][3], t[3];
    int i, k, nmatches = 0;
    const int bFactor = th != 1.0;
    for (i = 0; i < 9; i++) R[i / 3][i % 3] = Rcw[i];
    for (i = 0; i < 3; i++) t[i] = tcw[i];
    for (i = 0; i < n; i++) matched_mp[i] = -1;
    for (i = 0; i < m; i++) {
        float u, v, viewCos, r;
        int level, nc, bestDist = INT_MAX, bestLevel = -1, bestDist2 = INT_MAX, bestLevel2 = -1, bestIdx = -1;
        if (proj) { proj[4 * i] = proj[4 * i + 1] = proj[4 * i + 2] = 0; proj[4 * i + 3] = -1; }
        if (!oracle_in_frustum(mp_pos + 3 * i, mp_normal + 3 * i, mp_mind[i], mp_maxd[i], cam, R, t, 0.5f, scaleFactors, nLevels,
                               &u, &v, &level, &viewCos))
            continue;
        if (proj) { proj[4 * i] = u; proj[4 * i + 1] = v; proj[4 * i + 2] = viewCos; proj[4 * i + 3] = (float)level; }
        r = viewCos > 0.998 ? 3.0 : 4.5;
        if (bFactor) r *= th;
        nc = oracle_grid_features_in_area(g, kxy, koct, u, v, r * scaleFactors[level], level - 1, level, cand, n + 1);
        for (k = 0; k < nc; k++) {
            const int idx = cand[k];
            int dist;
            if (has_mp[idx]) continue;
            dist = oracle_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = koct[idx]; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = koct[idx]; bestDist2 = dist; }
        }
        if (bestDist <= th_reloc) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            matched_mp[bestIdx] = i;
            nmatches++;
        }
    }
    free(cand);
    oracle_grid_free(g);
    return nmatches;
}

/* ---- DBoW2 vocabulary-tree descent: TemplatedVocabulary::transform(feature, id,
 * weight, nid, levelsup), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1262,
 * with FORB::distance (FORB.cpp:81-101 = the same SWAR popcount).  Tree as flat
 * arrays: children of node i = child_ids[child_off[i] .. child_off[i+1]); leaves
 * have no children, a word id and a weight.  First child wins ties (strict d<best_d). */
void oracle_bow_transform(const int *child_off, const int *child_ids, const uint8_t *node_desc, const int *node_word,
                          const double *node_weight, int L, int levelsup, const uint8_t *feat, int n,
                          int *word_id, int *node_id, double *weight)
{
    const int nid_level = L - levelsup;
    int f;
    for (f = 0; f < n; f++) {
        int final_id = 0, current_level = 0, nid = 0, k;
        do {
            const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
            double best_d;
            ++current_level;
            final_id = child_ids[c0];
            best_d = oracle_descriptor_distance(feat + 32 * (size_t)f, node_desc + 32 * (size_t)final_id);
            for (k = c0 + 1; k < c1; k++) {
                const int id = child_ids[k];
                const double d = oracle_descriptor_distance(feat + 32 * (size_t)f, node_desc + 32 * (size_t)id);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (child_off[final_id + 1] > child_off[final_id]);
        word_id[f] = node_word[final_id]; node_id[f] = nid; weight[f] = node_weight[final_id];
    }
}

/* ---- MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:305-370, for a batch
 * of map points: observed descriptors of point i = rows off[i]..off[i+1); the
 * descriptor with the least median distance to the others wins (first on ties),
 * median = sorted row [int(0.5*(N-1))].  best[i] = row index within the point, -1 if N==0. */
static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
void oracle_distinctive_descriptors(const uint8_t *desc, const int *off, int m, int *best)
{
    int i, a, b;
    for (i = 0; i < m; i++) {
        const int N = off[i + 1] - off[i];
        const uint8_t *d = desc + 32 * (size_t)off[i];
        int BestMedian = INT_MAX, BestIdx = 0;
        int *row;
        if (N <= 0) { best[i] = -1; continue; }
        row = (int *)malloc(sizeof(int) * N);
        for (a = 0; a < N; a++) {
            int median;
            for (b = 0; b < N; b++) row[b] = a == b ? 0 : oracle_descriptor_distance(d + 32 * (size_t)a, d + 32 * (size_t)b);
            qsort(row, N, sizeof(int), int_cmp);
            median = row[(int)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = a; }
        }
        free(row);
        best[i] = BestIdx;
    }
}

/* ---- Frame::isInFrustum (src/Frame.cc:284-340) with MapPoint::PredictScale (src/MapPoint.cc:464-480) and the projection
 * blocks of the two ORBmatcher::Fuse forms (src/ORBmatcher.cc:1053-1094, :1212-1250), one point at a time as the reference
 * does.  The cv::Mat arithmetic (OpenCV 3.4, CV_32F; not under /root/reference -- parity unpinned) is restated as:
 * Rcw*P + tcw = gemm's 3x3 special case (row sum in float, left to right; result = float(double(sum) + double(t)));
 * cv::norm = sqrt of the double sum of squares; Mat::dot = double sum of double products.  std::log / std::ceil on
 * floats are the float overloads (using namespace std). */
typedef struct { float u, v, ur, view_cos, dist; int level, visible; } oracle_projected;

static int oracle_predict_scale(float mfMaxDistance, float currentDist, float logScaleFactor, int nScaleLevels)
{
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int)ceilf(logf(ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nScaleLevels) nScale = nScaleLevels - 1;
    return nScale;
}

static float oracle_radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 3.0f : 4.5f; } /* ORBmatcher.cc:332-338 */

void oracle_project_points(int mode, const float *pos, const float *nrm, const float *mfMinDistance, const float *mfMaxDistance,
                           int m, const float *Rcw, const float *tcw, const float *Ow, const float *cam4, const float *bounds4,
                           float mbf, float viewingCosLimit, float logScaleFactor, const float *scaleFactors, int nLevels,
                           float th, oracle_projected *out, oracle_wquery *q)
{
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float mnMinX = bounds4[0], mnMinY = bounds4[1], mnMaxX = bounds4[2], mnMaxY = bounds4[3];
    int i, k;
    for (i = 0; i < m; i++) {
        const float *P = pos + 3 * i, *Pn = nrm + 3 * i;
        float Pc[3], PO[3], invz, u, v, ur, dist, viewCos = 0.f, r;
        double dot;
        int level;
        oracle_projected o = {0.f, 0.f, 0.f, 0.f, 0.f, -1, 0};
        oracle_wquery w = {0.f, 0.f, -1.f, 0.f, 0, -1};
        out[i] = o;
        if (q) q[i] = w;
        for (k = 0; k < 3; k++) {
            const float s = Rcw[3 * k] * P[0] + Rcw[3 * k + 1] * P[1] + Rcw[3 * k + 2] * P[2];
            Pc[k] = (float)((double)s + (double)tcw[k]);
        }
        if (Pc[2] < 0.0f) continue;
        if (mode == 0) {
            invz = 1.0f / Pc[2];
            u = fx * Pc[0] * invz + cx;
            v = fy * Pc[1] * invz + cy;
            if (u < mnMinX || u > mnMaxX) continue;
            if (v < mnMinY || v > mnMaxY) continue;
        } else {
            float x, y;
            invz = mode == 1 ? 1 / Pc[2] : (float)(1.0 / Pc[2]);
            x = Pc[0] * invz; y = Pc[1] * invz;
            u = fx * x + cx; v = fy * y + cy;
            if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;
        }
        ur = u - mbf * invz;
        {
            const float maxDistance = 1.2f * mfMaxDistance[i], minDistance = 0.8f * mfMinDistance[i];
            double ss = 0;
            for (k = 0; k < 3; k++) PO[k] = P[k] - Ow[k];
            for (k = 0; k < 3; k++) ss += (double)PO[k] * (double)PO[k];
            dist = (float)sqrt(ss);
            dot = 0;
            for (k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn[k];
            if (mode == 0) {
                if (dist < 0.9 * minDistance || dist > maxDistance / 0.9) continue;
                viewCos = (float)(dot / dist);
                if (viewCos < viewingCosLimit) continue;
            } else {
                if (dist < minDistance || dist > maxDistance) continue;
                if (dot < 0.5 * dist) continue;
            }
        }
        level = oracle_predict_scale(mfMaxDistance[i], dist, logScaleFactor, nLevels);
        o.u = u; o.v = v; o.ur = ur; o.view_cos = viewCos; o.dist = dist; o.level = level; o.visible = 1;
        out[i] = o;
        if (mode == 0) {
            r = oracle_radius_by_viewing_cos(viewCos);
            if (th != 1.0) r *= th;
        } else {
            r = th;
        }
        if (q) { w.u = u; w.v = v; w.r = r * scaleFactors[level]; w.xr = ur; w.min_level = level - 1; w.max_level = level; q[i] = w; }
    }
}

/* ---- ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) as a whole, src/ORBmatcher.cc:1026-1176, on a toy map:
 * map points are indices 0..nmp-1 with Observations() = mp_obs[], isBad() = mp_bad[], their slot in this key frame
 * mp_in_kf[] (-1: IsInKeyFrame false); the key frame's mvpMapPoints = kf_mp[] (-1 = NULL).  vpMapPoints = list[] (-1 = NULL
 * entries, duplicates allowed).  Replace(a by b) (MapPoint.cc:209-258, what this loop can observe of it): a becomes bad,
 * b takes a's slot in this key frame unless b is already in it (then the slot is erased), b's observation count grows by
 * one per key frame taken over.  best / idx come from the candidate loop (oracle_search_fuse on the projected windows).
 * Returns nFused; ops[] receives the sequence of map operations {kind, mp, slot_or_other}: 0 = AddObservation + AddMapPoint,
 * 1 = pMP->Replace(pMPinKF) (pMP dies), 2 = pMPinKF->Replace(pMP). */
int oracle_fuse_replay(const int *list, int nlist, const int *visible, const int *best, const int *idx, int th_low,
                       int *mp_obs, uint8_t *mp_bad, int *mp_in_kf, int *kf_mp, int *ops, int *nops)
{
    int i, nFused = 0, no = 0;
    for (i = 0; i < nlist; i++) {
        const int pMP = list[i];
        if (pMP < 0) continue;
        if (mp_bad[pMP] || mp_in_kf[pMP] >= 0) continue;
        if (!visible[i] || idx[i] < 0) continue;          /* projection tests / vIndices.empty() / no candidate */
        if (best[i] <= th_low) {
            const int bestIdx = idx[i], pMPinKF = kf_mp[bestIdx];
            if (pMPinKF >= 0) {
                if (!mp_bad[pMPinKF]) {
                    if (mp_obs[pMPinKF] > mp_obs[pMP]) { /* pMP->Replace(pMPinKF): pMP is in no slot of this key frame */
                        mp_bad[pMP] = 1;
                        ops[3 * no] = 1; ops[3 * no + 1] = pMP; ops[3 * no + 2] = pMPinKF; no++;
                    } else {                              /* pMPinKF->Replace(pMP): pMP takes the slot */
                        mp_bad[pMPinKF] = 1;
                        mp_in_kf[pMPinKF] = -1;
                        kf_mp[bestIdx] = pMP; mp_in_kf[pMP] = bestIdx; mp_obs[pMP] += 1;
                        ops[3 * no] = 2; ops[3 * no + 1] = pMP; ops[3 * no + 2] = pMPinKF; no++;
                    }
                }
            } else {
                kf_mp[bestIdx] = pMP; mp_in_kf[pMP] = bestIdx; mp_obs[pMP] += 1;
                ops[3 * no] = 0; ops[3 * no + 1] = pMP; ops[3 * no + 2] = bestIdx; no++;
            }
            nFused++;
        }
    }
    *nops = no;
    return nFused;
}

/* ---- The Sim3 form, ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, vpReplacePoint), src/ORBmatcher.cc:
 * 1178-1301, tail on the same toy map.  What differs from the form above: the "already found" skip reads spAlreadyFound, a
 * SNAPSHOT of KeyFrame::GetMapPoints() (src/KeyFrame.cc:274-287: the key frame's non-NULL, non-bad map points) taken before
 * the loop (:1194), so a point the loop itself adds is not skipped when the list names it again; a taken slot is never
 * replaced, only recorded (vpReplacePoint[iMP] = pMPinKF when that point is not bad, :1283-1287); a free slot gets the
 * point (AddObservation + AddMapPoint, :1288-1292).  nFused counts both (:1293).  The reference dereferences every list
 * entry (no NULL test, :1203): a negative entry is a caller error here and is skipped.  replace[] is the caller's
 * vpReplacePoint (entries the loop does not write keep their value).  ops: {0, pMP, slot} for an add, {3, iMP, pMPinKF}
 * for a recorded replacement. */
int oracle_fuse_replay_sim3(const int *list, int nlist, const int *visible, const int *best, const int *idx, int th_low, int nmp,
                            int nkp, int *mp_obs, const uint8_t *mp_bad, int *mp_in_kf, int *kf_mp, int *replace, int *ops,
                            int *nops)
{
    uint8_t *already = (uint8_t *)calloc((size_t)(nmp > 0 ? nmp : 1), 1);
    int i, k, nFused = 0, no = 0;
    for (k = 0; k < nkp; k++)
        if (kf_mp[k] >= 0 && !mp_bad[kf_mp[k]]) already[kf_mp[k]] = 1;
    for (i = 0; i < nlist; i++) {
        const int pMP = list[i];
        if (pMP < 0) continue;
        if (mp_bad[pMP] || already[pMP]) continue;
        if (!visible[i] || idx[i] < 0) continue;          /* projection tests / vIndices.empty() / no candidate in level range */
        if (best[i] <= th_low) {
            const int bestIdx = idx[i], pMPinKF = kf_mp[bestIdx];
            if (pMPinKF >= 0) {
                if (!mp_bad[pMPinKF]) {
                    replace[i] = pMPinKF;
                    ops[3 * no] = 3; ops[3 * no + 1] = i; ops[3 * no + 2] = pMPinKF; no++;
                }
            } else {
                kf_mp[bestIdx] = pMP; mp_in_kf[pMP] = bestIdx; mp_obs[pMP] += 1;
                ops[3 * no] = 0; ops[3 * no + 1] = pMP; ops[3 * no + 2] = bestIdx; no++;
            }
            nFused++;
        }
    }
    *nops = no;
    free(already);
    return nFused;
}
