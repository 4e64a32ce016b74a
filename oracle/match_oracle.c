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

/* ================================================================================================================
 * The four SearchByProjection / SearchBySim3 forms AS WHOLE FUNCTIONS, projection prefix included -- literal loop
 * restatements of src/ORBmatcher.cc:1529-1671 (CurrentFrame, LastFrame), :1673-1800 (CurrentFrame, KeyFrame), :491-604
 * (KeyFrame, Scw) and :1303-1527 (SearchBySim3).  The pointer graph is passed flat: entry i of a frame's / key frame's /
 * list's map-point vector becomes valid[i] (what the reference's pointer tests leave: non-NULL, not an outlier / not bad /
 * not in the "already found" set), its world position, descriptor (MapPoint::GetDescriptor), mfMinDistance / mfMaxDistance,
 * normal, and takes[i] = Observations() > 0.  cv::Mat arithmetic (OpenCV 3.4, CV_32F; parity unpinned like the other OpenCV
 * primitives), as in oracle_project_points:
 *   A * b (+ c)          gemm's 3 x 3 special case: t = a0 b0 + a1 b1 + a2 b2 in float, left to right,
 *                        d = float(double(t) * alpha + double(c) * beta)   (c = 0, beta = 0 without a third operand)
 *   -A.t() * b           the transposed matrix materialised, then the product with alpha = -1
 *   s * A, A / s         convertTo with a FLOAT scale: a * float(s) + 0.0f, a * float(1.0 / s) + 0.0f
 *   A - B                float subtraction;  cv::norm = float(sqrt(double sum of squares));  Mat::dot = double sum of
 *                        double products.
 * Every function can return the GetFeaturesInArea query it forms per entry (qout, r < 0 = entry skipped before the search)
 * so that a device prefix is compared at float-bit level. */
static void cvm_gemm3(const float *A, const float *b, double alpha, const float *c, double beta, float *d)
{
    int k;
    float out[3];
    for (k = 0; k < 3; k++) {
        const float t = A[3 * k] * b[0] + A[3 * k + 1] * b[1] + A[3 * k + 2] * b[2];
        out[k] = (float)(t * alpha + (c ? c[k] : 0.0f) * beta);
    }
    d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
}
static void cvm_transpose3(const float *A, float *At)
{
    int r, c;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) At[3 * r + c] = A[3 * c + r];
}
static void cvm_scale(const float *A, int n, double s, float *out) /* convertTo(.., alpha = s): cvtScale with float(alpha), float(0) */
{
    const float a = (float)s, b = (float)0.0;
    int i;
    for (i = 0; i < n; i++) out[i] = A[i] * a + b;
}
static void cvm_pose_parts(const float *T16, float *R, float *t) /* mTcw.rowRange(0,3).colRange(0,3), .col(3) */
{
    int r, c;
    for (r = 0; r < 3; r++) { for (c = 0; c < 3; c++) R[3 * r + c] = T16[4 * r + c]; t[r] = T16[4 * r + 3]; }
}
static float cvm_norm3(const float *a) /* cv::norm(NORM_L2) of a 3 x 1 CV_32F */
{
    double s = 0;
    int k;
    for (k = 0; k < 3; k++) s += (double)a[k] * a[k];
    return (float)sqrt(s);
}
/* -R.t() * t: Ow / twc (ORBmatcher.cc:1542, :1679, :504) */
static void cvm_neg_Rt_t(const float *R, const float *t, float *out)
{
    float Rt[9];
    cvm_transpose3(R, Rt);
    cvm_gemm3(Rt, t, -1.0, NULL, 0.0, out);
}
void oracle_camera_centre(const float *T16, float *Ow)
{
    float R[9], t[3];
    cvm_pose_parts(T16, R, t);
    cvm_neg_Rt_t(R, t, Ow);
}
/* Scw -> Rcw, tcw, Ow as ORBmatcher.cc:500-504 (and :1186-1192) decompose it */
void oracle_decompose_sim3(const float *S16, float *Rcw, float *tcw, float *Ow)
{
    float sR[9], st[3], scw;
    double dot = 0;
    int k;
    cvm_pose_parts(S16, sR, st);
    for (k = 0; k < 3; k++) dot += (double)sR[k] * sR[k];          /* sRcw.row(0).dot(sRcw.row(0)) */
    scw = (float)sqrt(dot);
    cvm_scale(sR, 9, 1. / scw, Rcw);                               /* operator / (Mat, double): alpha = 1./s */
    cvm_scale(st, 3, 1. / scw, tcw);
    cvm_neg_Rt_t(Rcw, tcw, Ow);
}

/* bForward / bBackward of ORBmatcher.cc:1540-1550 */
void oracle_motion_direction(const float *Tcw16, const float *Tlw16, float mb, int bMono, int *bForward, int *bBackward)
{
    float Rcw[9], tcw[3], Rlw[9], tlw[3], twc[3], tlc[3];
    cvm_pose_parts(Tcw16, Rcw, tcw);
    cvm_pose_parts(Tlw16, Rlw, tlw);
    cvm_neg_Rt_t(Rcw, tcw, twc);
    cvm_gemm3(Rlw, twc, 1.0, tlw, 1.0, tlc);
    *bForward = tlc[2] > mb && !bMono;
    *bBackward = -tlc[2] > mb && !bMono;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono),
 * src/ORBmatcher.cc:1529-1671.  Current frame: mvKeysUn (kxy, koct, kangle), mDescriptors, mvuRight (NULL = all -1),
 * occupied[j] = "mvpMapPoints[j] holds a point with Observations() > 0" before the call (:1603-1605), grid bounds4 =
 * mnMinX, mnMinY, mnMaxX, mnMaxY, cam4 = fx, fy, cx, cy.  Last frame, per keypoint i: valid[i] = mvpMapPoints[i] &&
 * !mvbOutlier[i], pos / mp_desc / takes of that point, last_octave[i] = mvKeys[i].octave, last_angle[i] = mvKeysUn[i].angle.
 * match_kp[j] = i whose point ends in slot j (-1 untouched, -2 set to NULL by the rotation check), match_q[i] = bestIdx2
 * of an accepted i (before the rotation check) or -1.  Returns nmatches. */
int oracle_search_by_projection_last(const float *kxy, const int *koct, const float *kangle, const uint8_t *kdesc, int n,
                                     const float *uright, const uint8_t *occupied, const float *bounds4, const float *cam4,
                                     float mb, float mbf, const float *Tcw16, const float *scaleFactors, const float *Tlw16, int nl,
                                     const uint8_t *valid, const float *pos, const uint8_t *mp_desc, const uint8_t *takes,
                                     const int *last_octave, const float *last_angle, float th, int bMono, int th_high,
                                     int check_orientation, int *match_kp, int *match_q, oracle_wquery *qout)
{
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float mnMinX = bounds4[0], mnMinY = bounds4[1], mnMaxX = bounds4[2], mnMaxY = bounds4[3];
    oracle_grid *g = oracle_grid_build(kxy, n, mnMinX, mnMinY, mnMaxX, mnMaxY);
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    uint8_t *blocked = (uint8_t *)malloc(n + 1);
    int *qbin = (int *)malloc(sizeof(int) * (nl + 1));
    int hist[HISTO_LENGTH] = {0};
    float Rcw[9], tcw[3];
    int bForward, bBackward, i, k, nmatches = 0;
    cvm_pose_parts(Tcw16, Rcw, tcw);
    oracle_motion_direction(Tcw16, Tlw16, mb, bMono, &bForward, &bBackward);
    for (k = 0; k < n; k++) { blocked[k] = occupied ? occupied[k] : 0; match_kp[k] = -1; }
    for (i = 0; i < nl; i++) {
        float x3Dc[3], xc, yc, invzc, u, v, radius;
        int nLastOctave, nc, bestDist = 256, bestIdx2 = -1;
        const oracle_wquery none = {0.f, 0.f, -1.f, 0.f, 0, -1};
        match_q[i] = -1; qbin[i] = -1;
        if (qout) qout[i] = none;
        if (!valid[i]) continue;
        cvm_gemm3(Rcw, pos + 3 * i, 1.0, tcw, 1.0, x3Dc);
        xc = x3Dc[0]; yc = x3Dc[1];
        invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        u = fx * xc * invzc + cx;
        v = fy * yc * invzc + cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        nLastOctave = last_octave[i];
        radius = th * scaleFactors[nLastOctave];
        {
            const int minLevel = bForward ? nLastOctave : (bBackward ? 0 : nLastOctave - 1);
            const int maxLevel = bForward ? -1 : (bBackward ? nLastOctave : nLastOctave + 1);
            nc = oracle_grid_features_in_area(g, kxy, koct, u, v, radius, minLevel, maxLevel, cand, n + 1);
            if (qout) { const oracle_wquery w = {u, v, radius, u - mbf * invzc, minLevel, maxLevel}; qout[i] = w; }
        }
        if (nc == 0) continue;
        for (k = 0; k < nc; k++) {
            const int i2 = cand[k];
            int dist;
            if (blocked[i2]) continue;
            if (uright && uright[i2] > 0) {
                const float ur = u - mbf * invzc;
                const float er = fabsf(ur - uright[i2]);
                if (er > radius) continue;
            }
            dist = oracle_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= th_high) {
            match_kp[bestIdx2] = i;                    /* CurrentFrame.mvpMapPoints[bestIdx2] = pMP */
            match_q[i] = bestIdx2;
            if (takes[i]) blocked[bestIdx2] = 1;
            nmatches++;
            if (check_orientation) { qbin[i] = rot_bin(last_angle[i], kangle[bestIdx2]); hist[qbin[i]]++; }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i = 0; i < nl; i++)
            if (qbin[i] >= 0 && qbin[i] != ind1 && qbin[i] != ind2 && qbin[i] != ind3) { match_kp[match_q[i]] = -2; nmatches--; }
    }
    free(cand); free(blocked); free(qbin);
    oracle_grid_free(g);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
 * const int ORBdist), src/ORBmatcher.cc:1673-1800.  No depth test (:1706-1713 project whatever the sign of z), no stereo test;
 * occupied[j] = CurrentFrame.mvpMapPoints[j] != NULL before the call (:1741-1742), every assignment blocks its slot.
 * Key frame entry i: valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP), pos, mfMinDistance / mfMaxDistance, descriptor,
 * kf_angle[i] = pKF->mvKeysUn[i].angle. */
int oracle_search_by_projection_kf(const float *kxy, const int *koct, const float *kangle, const uint8_t *kdesc, int n,
                                   const uint8_t *occupied, const float *bounds4, const float *cam4, const float *Tcw16,
                                   const float *scaleFactors, int nLevels, float logScaleFactor, int nk, const uint8_t *valid,
                                   const float *pos, const float *mfMinDistance, const float *mfMaxDistance, const uint8_t *mp_desc,
                                   const float *kf_angle, float th, int ORBdist, int check_orientation, int *match_kp, int *match_q,
                                   oracle_wquery *qout)
{
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float mnMinX = bounds4[0], mnMinY = bounds4[1], mnMaxX = bounds4[2], mnMaxY = bounds4[3];
    oracle_grid *g = oracle_grid_build(kxy, n, mnMinX, mnMinY, mnMaxX, mnMaxY);
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    uint8_t *blocked = (uint8_t *)malloc(n + 1);
    int *qbin = (int *)malloc(sizeof(int) * (nk + 1));
    int hist[HISTO_LENGTH] = {0};
    float Rcw[9], tcw[3], Ow[3];
    int i, k, nmatches = 0;
    cvm_pose_parts(Tcw16, Rcw, tcw);
    cvm_neg_Rt_t(Rcw, tcw, Ow);
    for (k = 0; k < n; k++) { blocked[k] = occupied ? occupied[k] : 0; match_kp[k] = -1; }
    for (i = 0; i < nk; i++) {
        float x3Dc[3], PO[3], xc, yc, invzc, u, v, dist3D, maxDistance, minDistance, radius;
        int nPredictedLevel, nc, bestDist = 256, bestIdx2 = -1;
        const oracle_wquery none = {0.f, 0.f, -1.f, 0.f, 0, -1};
        match_q[i] = -1; qbin[i] = -1;
        if (qout) qout[i] = none;
        if (!valid[i]) continue;
        cvm_gemm3(Rcw, pos + 3 * i, 1.0, tcw, 1.0, x3Dc);
        xc = x3Dc[0]; yc = x3Dc[1];
        invzc = 1.0 / x3Dc[2];
        u = fx * xc * invzc + cx;
        v = fy * yc * invzc + cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        for (k = 0; k < 3; k++) PO[k] = pos[3 * i + k] - Ow[k];
        dist3D = cvm_norm3(PO);
        maxDistance = 1.2f * mfMaxDistance[i];
        minDistance = 0.8f * mfMinDistance[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        nPredictedLevel = oracle_predict_scale(mfMaxDistance[i], dist3D, logScaleFactor, nLevels);
        radius = th * scaleFactors[nPredictedLevel];
        nc = oracle_grid_features_in_area(g, kxy, koct, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, cand, n + 1);
        if (qout) { const oracle_wquery w = {u, v, radius, 0.f, nPredictedLevel - 1, nPredictedLevel + 1}; qout[i] = w; }
        if (nc == 0) continue;
        for (k = 0; k < nc; k++) {
            const int i2 = cand[k];
            int dist;
            if (blocked[i2]) continue;
            dist = oracle_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            match_kp[bestIdx2] = i;
            match_q[i] = bestIdx2;
            blocked[bestIdx2] = 1;
            nmatches++;
            if (check_orientation) { qbin[i] = rot_bin(kf_angle[i], kangle[bestIdx2]); hist[qbin[i]]++; }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (i = 0; i < nk; i++)
            if (qbin[i] >= 0 && qbin[i] != ind1 && qbin[i] != ind2 && qbin[i] != ind3) { match_kp[match_q[i]] = -2; nmatches--; }
    }
    free(cand); free(blocked); free(qbin);
    oracle_grid_free(g);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched,
 * int th), src/ORBmatcher.cc:491-604.  Key frame: mvKeysUn, mDescriptors, grid (KeyFrame::GetFeaturesInArea, KeyFrame.cc:613-652:
 * no level argument, the loop filters :580-583), occupied[j] = vpMatched[j] != NULL before the call.  List entry i: valid[i] =
 * !isBad() && !spAlreadyFound.count(pMP).  match_kp[j] = list entry written to vpMatched[j] (-1 untouched). */
int oracle_search_by_projection_sim3(const float *kxy, const int *koct, const uint8_t *kdesc, int n, const uint8_t *occupied,
                                     const float *bounds4, const float *cam4, const float *Scw16, const float *scaleFactors,
                                     int nLevels, float logScaleFactor, int np, const uint8_t *valid, const float *pos,
                                     const float *nrm, const float *mfMinDistance, const float *mfMaxDistance, const uint8_t *mp_desc,
                                     int th, int th_low, int *match_kp, int *match_q, oracle_wquery *qout, const float *kf_bounds4)
{
    /* kf_bounds4 (may be NULL): the KeyFrame's own int-valued mnMinX .. mnMaxY (include/KeyFrame.h:201-204) used by its
     * GetFeaturesInArea / IsInImage, while mGrid and the cell pitch are the Frame's (bounds4; src/KeyFrame.cc:36,44) */
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float *kb = kf_bounds4 ? kf_bounds4 : bounds4;
    const float mnMinX = kb[0], mnMinY = kb[1], mnMaxX = kb[2], mnMaxY = kb[3];
    oracle_grid *g = oracle_grid_build(kxy, n, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    uint8_t *blocked = (uint8_t *)malloc(n + 1);
    float Rcw[9], tcw[3], Ow[3];
    int iMP, k, nmatches = 0;
    oracle_decompose_sim3(Scw16, Rcw, tcw, Ow);
    g->minX = mnMinX; g->minY = mnMinY;            /* the queries' arithmetic; cell membership stays as built */
    for (k = 0; k < n; k++) { blocked[k] = occupied ? occupied[k] : 0; match_kp[k] = -1; }
    for (iMP = 0; iMP < np; iMP++) {
        const float *p3Dw = pos + 3 * iMP, *Pn = nrm + 3 * iMP;
        float p3Dc[3], PO[3], invz, x, y, u, v, maxDistance, minDistance, dist, radius;
        double dot = 0;
        int nPredictedLevel, nc, bestDist = 256, bestIdx = -1;
        const oracle_wquery none = {0.f, 0.f, -1.f, 0.f, 0, -1};
        match_q[iMP] = -1;
        if (qout) qout[iMP] = none;
        if (!valid[iMP]) continue;
        cvm_gemm3(Rcw, p3Dw, 1.0, tcw, 1.0, p3Dc);
        if (p3Dc[2] < 0.0) continue;
        invz = 1 / p3Dc[2];
        x = p3Dc[0] * invz;
        y = p3Dc[1] * invz;
        u = fx * x + cx;
        v = fy * y + cy;
        if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;   /* KeyFrame::IsInImage */
        maxDistance = 1.2f * mfMaxDistance[iMP];
        minDistance = 0.8f * mfMinDistance[iMP];
        for (k = 0; k < 3; k++) PO[k] = p3Dw[k] - Ow[k];
        dist = cvm_norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        for (k = 0; k < 3; k++) dot += (double)PO[k] * Pn[k];
        if (dot < 0.5 * dist) continue;
        nPredictedLevel = oracle_predict_scale(mfMaxDistance[iMP], dist, logScaleFactor, nLevels);
        radius = th * scaleFactors[nPredictedLevel];
        nc = oracle_grid_features_in_area(g, kxy, koct, u, v, radius, -1, -1, cand, n + 1);
        if (qout) { const oracle_wquery w = {u, v, radius, 0.f, nPredictedLevel - 1, nPredictedLevel}; qout[iMP] = w; }
        if (nc == 0) continue;
        for (k = 0; k < nc; k++) {
            const int idx = cand[k];
            int d;
            if (blocked[idx]) continue;
            if (koct[idx] < nPredictedLevel - 1 || koct[idx] > nPredictedLevel) continue;
            d = oracle_descriptor_distance(mp_desc + 32 * (size_t)iMP, kdesc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= th_low) {
            match_kp[bestIdx] = iMP;      /* vpMatched[bestIdx] = pMP */
            match_q[iMP] = bestIdx;
            blocked[bestIdx] = 1;
            nmatches++;
        }
    }
    free(cand); free(blocked);
    oracle_grid_free(g);
    return nmatches;
}

/* One direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1348-1428 with (Ra, ta) = (R1w, t1w), (sRb, tb) = (sR21, t21),
 * searched in key frame 2; :1430-1507 the other way round).  vnMatch[i] = bestIdx or -1. */
static void sim3_direction(const float *Ra, const float *ta, const float *sRb, const float *tb, int na, const uint8_t *valid,
                           const float *pos, const float *mfMinDistance, const float *mfMaxDistance, const uint8_t *mp_desc,
                           const oracle_grid *g, const float *kxy, const int *koct, const uint8_t *kdesc, int n,
                           const float *bounds4, const float *cam4, const float *scaleFactors, int nLevels, float logScaleFactor,
                           float th, int th_high, int *vnMatch, oracle_wquery *qout)
{
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float mnMinX = bounds4[0], mnMinY = bounds4[1], mnMaxX = bounds4[2], mnMaxY = bounds4[3];
    int *cand = (int *)malloc(sizeof(int) * (n + 1));
    int i, k;
    for (i = 0; i < na; i++) {
        float pa[3], pb[3], invz, x, y, u, v, maxDistance, minDistance, dist3D, radius;
        int nPredictedLevel, nc, bestDist = INT_MAX, bestIdx = -1;
        const oracle_wquery none = {0.f, 0.f, -1.f, 0.f, 0, -1};
        vnMatch[i] = -1;
        if (qout) qout[i] = none;
        if (!valid[i]) continue;
        cvm_gemm3(Ra, pos + 3 * i, 1.0, ta, 1.0, pa);
        cvm_gemm3(sRb, pa, 1.0, tb, 1.0, pb);
        if (pb[2] < 0.0) continue;
        invz = 1.0 / pb[2];
        x = pb[0] * invz;
        y = pb[1] * invz;
        u = fx * x + cx;
        v = fy * y + cy;
        if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;
        maxDistance = 1.2f * mfMaxDistance[i];
        minDistance = 0.8f * mfMinDistance[i];
        dist3D = cvm_norm3(pb);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        nPredictedLevel = oracle_predict_scale(mfMaxDistance[i], dist3D, logScaleFactor, nLevels);
        radius = th * scaleFactors[nPredictedLevel];
        nc = oracle_grid_features_in_area(g, kxy, koct, u, v, radius, -1, -1, cand, n + 1);
        if (qout) { const oracle_wquery w = {u, v, radius, 0.f, nPredictedLevel - 1, nPredictedLevel}; qout[i] = w; }
        if (nc == 0) continue;
        for (k = 0; k < nc; k++) {
            const int idx = cand[k];
            int d;
            if (koct[idx] < nPredictedLevel - 1 || koct[idx] > nPredictedLevel) continue;
            d = oracle_descriptor_distance(mp_desc + 32 * (size_t)i, kdesc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= th_high) vnMatch[i] = bestIdx;
    }
    free(cand);
}

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1303-1527.  Both key frames share
 * the calibration and the image bounds (static members).  valid1[i] = vpMapPoints1[i] && !vbAlreadyMatched1[i] && !isBad()
 * (:1352-1358), valid2 likewise (:1434-1440; vbAlreadyMatched2 comes from GetIndexInKeyFrame, :1338-1341).  T1w / T2w:
 * GetRotation / GetTranslation of the two key frames as 4 x 4 poses.  match12[i1] = idx2 where both directions agree (the
 * entries the reference overwrites in vpMatches12, :1513-1521), else -1.  Returns nFound. */
int oracle_search_by_sim3(const float *kxy1, const int *koct1, const uint8_t *kdesc1, int n1, const float *kxy2, const int *koct2,
                          const uint8_t *kdesc2, int n2, const float *bounds4, const float *cam4, const float *scaleFactors,
                          int nLevels, float logScaleFactor, const float *T1w16, const float *T2w16, float s12, const float *R12,
                          const float *t12, const uint8_t *valid1, const float *pos1, const float *mind1, const float *maxd1,
                          const uint8_t *desc_mp1, const uint8_t *valid2, const float *pos2, const float *mind2, const float *maxd2,
                          const uint8_t *desc_mp2, float th, int th_high, int *vnMatch1, int *vnMatch2, int *match12,
                          oracle_wquery *q12, oracle_wquery *q21, const float *kf_bounds4)
{
    float R1w[9], t1w[3], R2w[9], t2w[3], sR12[9], R12t[9], sR21[9], t21[3];
    oracle_grid *g1 = oracle_grid_build(kxy1, n1, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    oracle_grid *g2 = oracle_grid_build(kxy2, n2, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    int i1, nFound = 0;
    if (kf_bounds4) { g1->minX = g2->minX = kf_bounds4[0]; g1->minY = g2->minY = kf_bounds4[1]; bounds4 = kf_bounds4; }   /* as in the form above */
    cvm_pose_parts(T1w16, R1w, t1w);
    cvm_pose_parts(T2w16, R2w, t2w);
    cvm_scale(R12, 9, s12, sR12);                 /* sR12 = s12 * R12 */
    cvm_transpose3(R12, R12t);
    cvm_scale(R12t, 9, 1.0 / s12, sR21);          /* sR21 = (1.0 / s12) * R12.t() */
    cvm_gemm3(sR21, t12, -1.0, NULL, 0.0, t21);   /* t21 = -sR21 * t12 */
    sim3_direction(R1w, t1w, sR21, t21, n1, valid1, pos1, mind1, maxd1, desc_mp1, g2, kxy2, koct2, kdesc2, n2, bounds4, cam4,
                   scaleFactors, nLevels, logScaleFactor, th, th_high, vnMatch1, q12);
    sim3_direction(R2w, t2w, sR12, t12, n2, valid2, pos2, mind2, maxd2, desc_mp2, g1, kxy1, koct1, kdesc1, n1, bounds4, cam4,
                   scaleFactors, nLevels, logScaleFactor, th, th_high, vnMatch2, q21);
    for (i1 = 0; i1 < n1; i1++) {
        const int idx2 = vnMatch1[i1];
        match12[i1] = -1;
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { match12[i1] = idx2; nFound++; }
        }
    }
    oracle_grid_free(g1); oracle_grid_free(g2);
    return nFound;
}

/* Transforms of SearchBySim3 (:1320-1323) for a caller that hands them to a device prefix: sR12, sR21, t21. */
void oracle_sim3_transforms(float s12, const float *R12, const float *t12, float *sR12, float *sR21, float *t21)
{
    float R12t[9];
    cvm_scale(R12, 9, s12, sR12);
    cvm_transpose3(R12, R12t);
    cvm_scale(R12t, 9, 1.0 / s12, sR21);
    cvm_gemm3(sR21, t12, -1.0, NULL, 0.0, t21);
}
