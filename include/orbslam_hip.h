/*
 * orbslam_hip.h -- C ABI of the MI355X-native front-end hot path of ORB_SLAM2_E.
 *
 * One shared library (liborbslam_hip.so), plain pointers and sizes, int status
 * codes, never throws.  Each entry point names the reference interface it
 * replaces (paths relative to the ORB_SLAM2_E tree).  INTEGRATION.md shows the
 * reference-side C++ binding (ORBextractor / ORBmatcher / FEA2 shells).
 *
 * Conventions
 *   - `*_dev` pointers are device (HBM) addresses, everything else is host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own stream).
 *   - Status: 0 = ORBX_OK, negative = error (see enum).  No CPU fallback exists:
 *     without a usable HIP device every compute call returns ORBX_ERR_NO_DEVICE.
 */
#ifndef ORBSLAM_HIP_H
#define ORBSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORBX_OK = 0,
    ORBX_ERR_ARG = -1,        /* null pointer, bad size, image too small for the 30-px cell grid */
    ORBX_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure at init */
    ORBX_ERR_HIP = -3,        /* a HIP call failed; orbx_last_error() has the text */
    ORBX_ERR_CAPACITY = -4,   /* caller buffer or configured capacity too small */
    ORBX_ERR_UNSUPPORTED = -5 /* parameter outside the supported range */
};

const char *orbx_last_error(void);
/* ABI version of this header (major*100+minor). */
int orbx_abi_version(void);

/* ------------------------------------------------------------------ extractor
 * Replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:46-112,
 * src/ORBextractor.cc:410-470 ctor, :1051-1113 operator()).                   */

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct orbx_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbx_keypoint;

typedef struct orbx_params {
    int32_t nfeatures;     /* ORBextractor ctor arg 1 */
    float scale_factor;    /* arg 2 */
    int32_t nlevels;       /* arg 3 (1..16) */
    int32_t ini_th_fast;   /* arg 4 */
    int32_t min_th_fast;   /* arg 5 */
    int32_t blur_variant;  /* 0: 7-tap 8.8 kernel {18,34,48,56,48,34,18} (sum 256, canonical);
                              1: {18,34,49,55,49,34,18} (taps rounded individually) */
    int32_t trig_variant;  /* cos / sin of the keypoint angle in computeOrbDescriptor (ORBextractor.cc:111-113: a float argument under
                              `using namespace std`, i.e. cosf / sinf):
                              0: cosf / sinf as glibc >= 2.28 computes them (any current Linux; bit-exact restatement, checked over
                                 every float in [0, 2 pi] by tools/trig/trig_variant_count.c);
                              1: the correctly rounded float of the double-precision cos / sin (differs from 0 by one ulp at 0.36 % /
                                 0.83 % of the angles: DESIGN 4.2) */
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) */
int orbx_create(const orbx_params *params, orbx_extractor **out);
int orbx_destroy(orbx_extractor *ex);

/* Getters: ORBextractor.h:61-84 (GetLevels, GetScaleFactors, GetInverseScaleFactors,
 * GetScaleSigmaSquares, GetInverseScaleSigmaSquares) + mnFeaturesPerLevel (:97). */
int orbx_get_levels(const orbx_extractor *ex);
int orbx_get_scale_factors(const orbx_extractor *ex, float *out_nlevels);
int orbx_get_inv_scale_factors(const orbx_extractor *ex, float *out_nlevels);
int orbx_get_level_sigma2(const orbx_extractor *ex, float *out_nlevels);
int orbx_get_inv_level_sigma2(const orbx_extractor *ex, float *out_nlevels);
int orbx_get_features_per_level(const orbx_extractor *ex, int32_t *out_nlevels);
/* Upper bound on keypoints per frame: nfeatures + 3*nlevels. */
int orbx_keypoint_capacity(const orbx_extractor *ex);

/* Size the device workspace for `batch` frames of width x height (idempotent;
 * called implicitly by the extract calls). */
int orbx_reserve(orbx_extractor *ex, int width, int height, int batch);

/* What orbx_reserve decides about a frame size, computed on the host alone (no device needed): level sizes
 * (ComputePyramid, ORBextractor.cc:1119-1121), quotas, DistributeOctTree's initial nodes round(W / H) per level (:543), the
 * keypoint slots of a level = max(quota + 3, 4 nIni) + 1, the frame's keypoint capacity, the FAST cell count per level
 * (:781-806) and the workspace / LDS budgets.  Same error codes as orbx_reserve for sizes it refuses. */
typedef struct {
    int32_t nlevels, keypoint_capacity, octree_nodes, cells_per_frame, sel_per_frame, fast_tile_stride, fast_lds, octree_lds, octree_kshift, reserved;
    int64_t frame_bytes, cands_per_frame;
    int32_t level_w[16], level_h[16], level_quota[16], level_nini[16], level_slots[16], level_cells[16];
} orbx_plan_info;
int orbx_plan(const orbx_params *params, int width, int height, orbx_plan_info *info);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) for ONE host image
 * (CV_8UC1, `stride` bytes per row).  Writes up to `cap` keypoints / 32-byte
 * descriptor rows; *n = count.  An empty image (NULL / 0 size) returns ORBX_OK
 * with outputs untouched and *n = 0 (ORBextractor.cc:1054-1055). */
int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, int stride,
                 orbx_keypoint *kps, uint8_t *desc, int cap, int *n);

/* The two images of a stereo frame (Frame.cc:78-81: ExtractORB on two threads, then join) in one call from one host thread:
 * both kernel chains are enqueued on their handles' streams before the host waits for either, so they overlap on the
 * device as the reference's threads overlap on the CPU.  Results as from two orbx_extract calls (left != right; same
 * error behaviour).  Frame::ComputeStereoMatches = orbx_stereo_match on the two handles afterwards. */
int orbx_extract_pair(orbx_extractor *left, const uint8_t *image_left, orbx_extractor *right, const uint8_t *image_right, int width,
                      int height, int stride, orbx_keypoint *kps_left, uint8_t *desc_left, int cap_left, int *n_left,
                      orbx_keypoint *kps_right, uint8_t *desc_right, int cap_right, int *n_right);

/* Batch form: `batch` frames, frame f at images + f*frame_stride.  images_dev may
 * be a device pointer (is_device=1: inputs already resident in HBM) or host.
 * Results stay on the device until orbx_download / orbx_result_dev. Asynchronous
 * on `stream`. */
int orbx_extract_batch(orbx_extractor *ex, const uint8_t *images, int is_device,
                       int width, int height, int stride, size_t frame_stride, int batch, void *stream);
/* Copy frame f's result of the last batch to host (synchronises the stream). */
int orbx_download(orbx_extractor *ex, int frame, orbx_keypoint *kps, uint8_t *desc, int cap, int *n);
/* Whole last batch to host in three copies: kps[batch][capacity],
 * desc[batch][capacity][32], counts[batch]; any pointer may be NULL. */
int orbx_download_batch(orbx_extractor *ex, orbx_keypoint *kps, uint8_t *desc, int32_t *counts);
/* Device-side result arrays of the last batch: kps[batch][capacity],
 * desc[batch][capacity][32], counts[batch] (int32). */
int orbx_result_dev(orbx_extractor *ex, const orbx_keypoint **kps_dev, const uint8_t **desc_dev,
                    const int32_t **counts_dev, int *capacity);

/* Asynchronous export of the last batch's result arrays into caller-owned buffers
 * (same shapes as orbx_result_dev) -- device memory, or pinned host memory for a
 * download that overlaps the next batch; any pointer may be NULL.  Used to stage the
 * fixed-size records of the multi-GPU gather. */
int orbx_copy_results_dev(orbx_extractor *ex, orbx_keypoint *kps_dst_dev, uint8_t *desc_dst_dev,
                          int32_t *counts_dst_dev, void *stream);

/* mvImagePyramid[level] of frame f of the last call (ORBextractor.h:86; read by
 * Frame::ComputeStereoMatches, src/Frame.cc:534,624,636,641): copies the level
 * image (without border) into out[height][out_stride]. */
int orbx_level_size(const orbx_extractor *ex, int level, int *width, int *height);
int orbx_pyramid_level(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride);
/* Same with the 19-px REFLECT_101 border: out[(h+38)][out_stride], width w+38. */
int orbx_pyramid_level_padded(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride);

/* Frame::ComputeStereoMatches (src/Frame.cc:527-701) for every frame of the last
 * batch extracted on `left` and `right` (two distinct handles with the same
 * parameters, Frame.cc:78-81): row-band Hamming match (+-1 octave, disparity in
 * [0, mbf/mb]), 11x11 L1 sub-pixel refinement on the keypoint's pyramid level,
 * median cut.  Results stay on the device in the left handle; download returns
 * mvuRight[n] / mvDepth[n] (-1 where unmatched) for one frame. */
int orbx_stereo_match(orbx_extractor *left, orbx_extractor *right, float mb, float mbf, void *stream);
int orbx_stereo_download(orbx_extractor *left, int frame, float *uRight, float *depth, int cap, int *n);
/* All frames of the last orbx_stereo_match at once: uRight / depth [batch][capacity] (rows past a frame's count are
 * unspecified), counts[batch] = the left keypoint counts; any of the three may be NULL. */
int orbx_stereo_download_batch(orbx_extractor *left, float *uRight, float *depth, int32_t *counts);

/* Staged outputs for parity tests (no reference counterpart; they expose the
 * intermediate values the reference keeps in locals):
 *   blurred level image (GaussianBlur output, ORBextractor.cc:1093-1094),
 *   FAST candidates per level = vToDistributeKeys (:826-834) as (x,y,response) triples,
 *   per-level keypoints after DistributeOctTree + orientation (:842-860). */
int orbx_debug_blurred_level(orbx_extractor *ex, int frame, int level, uint8_t *out, int out_stride);
int orbx_debug_level_candidates(orbx_extractor *ex, int frame, int level, float *xyr, int cap, int *n);
int orbx_debug_level_keypoints(orbx_extractor *ex, int frame, int level, orbx_keypoint *kps, int cap, int *n);

/* Per-kernel timing with HIP events on the launch stream (no reference
 * counterpart; feeds bench.py's roofline object).  `on`: 0 = off, < 0 = every
 * kernel kind, > 0 = bit mask of kinds (bit i = i-th name returned by read): each
 * recorded event widens the dispatch gap by a few microseconds, so a timed run
 * enables only the kernel it reports.  Read returns, per kernel kind, its name,
 * accumulated milliseconds and launch count since enable. */
int orbx_profile_enable(orbx_extractor *ex, int on);
int orbx_profile_read(orbx_extractor *ex, int max_kinds, const char **names, double *total_ms,
                      int64_t *launches, int *nkinds);

/* -------------------------------------------------------------------- matcher
 * Data plane of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:37-111).            */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1848-1864) for all pairs:
 * out[nA][nB] uint16.  Host pointers; runs on the device. */
int orbm_hamming_matrix(const uint8_t *A, int nA, const uint8_t *B, int nB, uint16_t *out);

/* Best / second-best / arg-best per query row over all of B, first index wins
 * ties -- the selection loop shared by every ORBmatcher::Search* (e.g.
 * ORBmatcher.cc:645-672).  best/second = INT32_MAX and idx = -1 when nB == 0. */
int orbm_match_bruteforce(const uint8_t *A, int nA, const uint8_t *B, int nB,
                          int32_t *best, int32_t *second, int32_t *idx);
/* Same over gated candidate lists (CSR: cand_off[nA+1], cand_idx[] in
 * GetFeaturesInArea / BoW-member order). */
int orbm_match_candidates(const uint8_t *A, int nA, const uint8_t *B, int nB,
                          const int32_t *cand_off, const int32_t *cand_idx,
                          int32_t *best, int32_t *second, int32_t *idx);
/* Frame::GetFeaturesInArea(u, v, r, minLevel, maxLevel) (src/Frame.cc:342-395, on
 * the 64x48 grid of Frame::AssignFeaturesToGrid :245-260) fused with the best /
 * second-best-with-levels loop that the SearchByProjection family runs over its
 * result (ORBmatcher.cc:69-118; same shape at :166-205, :1600-1640, :1730-1770).
 * kps = mvKeysUn, (min_x..max_y) = mnMinX..mnMaxY; skip[j] != 0 excludes keypoint j
 * (already holds an observed MapPoint, :87-89); uright (may be NULL) enables the
 * stereo check |xr - uRight[j]| <= r of :91-96.  init_dist: 256 or INT32_MAX.
 * Outputs per query: best / second distance, their octaves, arg-best (-1 if none).
 * Queries are independent here; the loops that assign matches as they go (e.g. :126
 * F.mvpMapPoints[bestIdx]=pMP) are orbm_search_projection and the whole-function entries below. */
typedef struct orbm_window_query {
    float u, v, r, xr;
    int32_t min_level, max_level;
} orbm_window_query;
int orbm_search_window(const orbm_window_query *queries, const uint8_t *qdesc, int nq, const orbx_keypoint *kps,
                       const uint8_t *desc, int n, const uint8_t *skip, const float *uright, float min_x, float min_y,
                       float max_x, float max_y, int init_dist, int32_t *best, int32_t *best_level, int32_t *second,
                       int32_t *second_level, int32_t *idx);

/* Candidate loop of ORBmatcher::Fuse (src/ORBmatcher.cc:1092-1146; Sim3 form :1245-1276): per map point that passed
 * the caller's projection / distance / viewing-angle tests, the window (u, v, r), levels [min_level, max_level] =
 * [l-1, l], xr = ur; a keypoint is scored only if its reprojection error passes e2 * inv_level_sigma2[level] <= 7.8
 * (stereo keypoint, uright[j] >= 0: ex, ey, er) or <= 5.99 (mono: ex, ey) -- inv_level_sigma2 = NULL switches the
 * gate off (the Sim3 form has none).  best[i] = smallest distance (256 if none), idx[i] = its keypoint (-1).  The map
 * update that follows in the reference (:1149-1170: Replace / AddObservation, nFused) reads only these two values
 * and stays on the host, in the reference's order. */
int orbm_search_fuse(const orbm_window_query *queries, const uint8_t *qdesc, int nq, const orbx_keypoint *kps, const uint8_t *desc,
                     int n, const float *uright, const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float max_x,
                     float max_y, int32_t *best, int32_t *idx);

/* The SearchByProjection family as whole loops: the selection of orbm_search_window plus
 * what the reference does between queries and after them,
 *   SearchByProjection(Frame&, vector<MapPoint*>&, th)          src/ORBmatcher.cc:46-132
 *       th_accept = TH_HIGH, ratio_same_level = 1 (:121), check_orientation = 0
 *   SearchByProjection(CurrentFrame, LastFrame, th, bMono)      :1529-1671   TH_HIGH, 0, mbCheckOrientation
 *   SearchByProjection(CurrentFrame, KeyFrame*, found, th, d)   :1673-1800   ORBdist, 0, mbCheckOrientation
 *   SearchByProjection(KeyFrame*, Scw, points, matched, th)     :491-604     TH_LOW,  0, 0
 * Query i = one map point that passed the caller's projection / frustum tests, in the
 * caller's loop order: its window and level range (as orbm_search_window), xr = u - bf/z,
 * its descriptor (MapPoint::GetDescriptor), qangle[i] = angle of its source keypoint (only
 * read when check_orientation), qtakes[i] != 0 when the pointer it leaves in
 * mvpMapPoints[best] makes later queries skip that keypoint (Observations() > 0 in the
 * first two forms; NULL = always, the last two).  occupied[j] != 0: keypoint j is skipped
 * from the start.  Each query sees the assignments of the queries before it, exactly as
 * the sequential loop (:87-89, :1603-1605, :1741-1742, :574-575).  Then the 30-bin rotation
 * histogram, ComputeThreeMaxima (:1802-1843) and the rejection of the other bins.
 * match_kp[j] = query whose point ends in slot j, -1 = slot untouched, -2 = slot set to
 * NULL by the rotation check; match_q[i] = keypoint chosen by query i when it was accepted
 * (before the rotation check) or -1; *nmatches as the reference counts it.
 * Limits: at most 8192 keypoints inside the grid, 65536 queries. */
int orbm_search_projection(const orbm_window_query *queries, const uint8_t *qdesc, const float *qangle, const uint8_t *qtakes,
                           int nq, const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *occupied,
                           const float *uright, float min_x, float min_y, float max_x, float max_y, int th_accept, float nnratio,
                           int ratio_same_level, int check_orientation, int32_t *match_kp, int32_t *match_q, int *nmatches);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (src/ORBmatcher.cc:606-721), whole: level-0 keypoints of F1, window around
 * vbPrevMatched[i1] on level 0 of F2 (grid bounds = F2's mnMinX..mnMaxY), candidates whose
 * vMatchedDistance is not above the new distance are skipped (:645-646), best <= TH_LOW and
 * best < second * nnratio, a better match steals the keypoint (:678-682), rotation
 * histogram over every accepted match (stolen ones included, as the reference pushes
 * them), ComputeThreeMaxima, rejection, vbPrevMatched update (:715-718).  kps1 / kps2 =
 * mvKeysUn.  matches12[n1] = vnMatches12; prev_matched[n1][2] in/out. */
int orbm_search_for_initialization(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbx_keypoint *kps2,
                                   const uint8_t *desc2, int n2, float *prev_matched, float min_x, float min_y, float max_x,
                                   float max_y, int window_size, float nnratio, int check_orientation, int32_t *matches12,
                                   int *nmatches);

/* ORBmatcher::SearchByBoW, both forms, as a whole loop:
 *   SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)   src/ORBmatcher.cc:360-489  valid2 = NULL, th = TH_LOW, strict_th = 0 (:429)
 *   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)      :723-856                   valid2 given, th = TH_LOW, strict_th = 1 (:799)
 * A DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) is passed flat, in map order:
 * nodes[nn] ascending, members of node k = items[off[k] .. off[k+1]) in insertion order.
 * valid1[i] != 0: feature i of the first keyframe owns a good MapPoint (:395-399, :763-767);
 * valid2 likewise for the second keyframe (:782-786).  The lower_bound co-iteration of
 * :384-456 runs on the host; then, in the reference's visiting order, every feature of the
 * first set selects best / second (from 256) over the members of its node in the second
 * set that no earlier feature has matched (:415-416, :784), accepts on best <= th (< th if
 * strict_th) and (float)best < nnratio * (float)second; rotation histogram +
 * ComputeThreeMaxima + rejection.  match12[n1] = feature of the second set or -1;
 * match21[n2] (may be NULL) the inverse; *nmatches as the reference returns it.
 * At most 8192 features in the second set. */
int orbm_search_by_bow(const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1, const uint8_t *valid1,
                       const uint8_t *desc1, const float *angle1, int n1, const int32_t *nodes2, const int32_t *off2,
                       const int32_t *items2, int nn2, const uint8_t *valid2, const uint8_t *desc2, const float *angle2, int n2,
                       int th, int strict_th, float nnratio, int check_orientation, int32_t *match12, int32_t *match21,
                       int *nmatches);

/* The fork's whole-map relocalisation search, ORBmatcher::SearchByProjection(Frame&,
 * Map*, double Rcw[3][3], double tcw[3], ...) (src/ORBmatcher.cc:134-222): for
 * EVERY map point isInFrustum (:262-330, with ComputeDistance :224-260; mixed
 * float / double arithmetic reproduced operation by operation), level prediction
 * by lower_bound on the scale factors, window r = RadiusByViewingCos * th *
 * scale[level], GetFeaturesInArea(u, v, r, level-1, level), best / second with
 * levels over keypoints that hold no MapPoint, bestDist <= th_reloc and the
 * same-level ratio test.  matched_mp[j] = index of the (last) map point assigned
 * to keypoint j or -1 (vMatchedMPs); *nmatches as the reference counts them.
 * cam: intrinsics, GetImageBounds() ints, grid bounds mnMinX..mnMaxY.  proj
 * (optional, [m][4]): u, v, viewCos, level per map point (level -1 = culled). */
typedef struct orbm_camera {
    float fx, fy, cx, cy;
    int32_t min_x, max_x, min_y, max_y;
    float grid_min_x, grid_min_y, grid_max_x, grid_max_y;
} orbm_camera;
int orbm_search_by_projection_map(const orbx_keypoint *kps, const uint8_t *desc, int n, const uint8_t *has_mappoint,
                                  const float *mp_pos, const float *mp_normal, const float *mp_min_dist,
                                  const float *mp_max_dist, const uint8_t *mp_desc, int m, const double *Rcw,
                                  const double *tcw, const orbm_camera *cam, const float *scale_factors, int nlevels,
                                  float th, float nnratio, int th_reloc, int32_t *matched_mp, int *nmatches, float *proj);

/* Batch projection of map points (M15): replaces the per-point host loop in front of every SearchByProjection /
 * Fuse form.  mode ORBM_PROJECT_FRUSTUM = Frame::isInFrustum (src/Frame.cc:284-340: positive depth, image bounds
 * mnMinX..mnMaxY inclusive, distance in [0.9 min, max / 0.9], viewing cosine >= viewing_cos_limit) with
 * MapPoint::PredictScale (src/MapPoint.cc:464-480) -> mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos, and the
 * window of SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:62-70: RadiusByViewingCos, * th
 * unless th == 1, * mvScaleFactors[level], levels [l-1, l], xr = mTrackProjXR);
 * ORBM_PROJECT_FUSE = the projection block of ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:1053-1094: KeyFrame::IsInImage, distance in [min, max], PO.Pn >= 0.5 dist, radius th *
 * mvScaleFactors[level]); ORBM_PROJECT_FUSE_SIM3 = the Sim3 form (:1212-1250; the caller passes Rcw = sRcw / s,
 * tcw = t / s, Ow = -Rcw' tcw as :1188-1192 computes them).
 * mp_min_distance / mp_max_distance are MapPoint::mfMinDistance / mfMaxDistance (the 0.8 / 1.2 factors of
 * Get{Min,Max}DistanceInvariance are applied here); Rcw row-major 3x3, tcw, Ow = camera centre, all float as the
 * cv::Mat members are; cam->grid_* = mnMinX .. mnMaxY; log_scale_factor = mfLogScaleFactor.
 * out[i].visible = 0 for a point the reference rejects (then level = -1 and queries[i].r < 0 = "no candidates", so
 * the arrays feed orbm_search_projection / orbm_search_fuse as they are).  queries may be NULL. */
enum { ORBM_PROJECT_FRUSTUM = 0, ORBM_PROJECT_FUSE = 1, ORBM_PROJECT_FUSE_SIM3 = 2 };
typedef struct orbm_projected_point {
    float u, v, ur, view_cos, dist; /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos, |P - Ow| */
    int32_t level;                  /* mnTrackScaleLevel / nPredictedLevel, -1 if rejected */
    int32_t visible;                /* mbTrackInView */
} orbm_projected_point;
int orbm_project_points(int mode, const float *mp_pos, const float *mp_normal, const float *mp_min_distance,
                        const float *mp_max_distance, int m, const float *Rcw, const float *tcw, const float *Ow,
                        const orbm_camera *cam, float mbf, float viewing_cos_limit, float log_scale_factor,
                        const float *scale_factors, int nlevels, float th, orbm_projected_point *out, orbm_window_query *queries);

/* Inner loop of ORBmatcher::SearchForTriangulation (ORBmatcher.cc:892-990) with
 * CheckDistEpipolarLine (:341-358): per keypoint of KF1 (skipped if it owns a
 * MapPoint, or is mono while only_stereo), scan its BoW-node candidates of KF2 in
 * member order (CSR lists built by the host-side FeatureVector co-iteration):
 * dist <= TH_LOW && dist <= bestDist (non-strict), epipole distance^2 >=
 * 100*scaleFactor[octave2] unless one side is stereo, epipolar line d^2 <
 * 3.84*sigma2[octave2].  F12 row-major 3x3 float (LocalMapping::ComputeF12),
 * (ex, ey) the epipole in image 2.  match12[i] = index in KF2 or -1; best_dist[i].
 * The rotation histogram and the pair list stay on the host (ORBmatcher.cc:992-1023). */
int orbm_match_triangulation(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbx_keypoint *kps2,
                             const uint8_t *desc2, int n2, const int32_t *cand_off, const int32_t *cand_idx,
                             const uint8_t *has_mappoint1, const uint8_t *has_mappoint2, const uint8_t *stereo1,
                             const uint8_t *stereo2, int only_stereo, const float *F12, float ex, float ey,
                             const float *scale_factors2, const float *level_sigma2, int nlevels, int32_t *match12,
                             int32_t *best_dist);

/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:858-1024) as a whole: the FeatureVector co-iteration (:881-891,
 * :1004-1012; a FeatureVector = nodes ascending, off[nn + 1], items, as for orbm_search_by_bow), the gated loop above, the
 * rotation histogram + ComputeThreeMaxima + rejection when check_orientation (:992-1012).  match12[n1] = index in KF2 or
 * -1; vMatchedPairs = its non-negative entries in index order (:1014-1021); *nmatches = their number. */
int orbm_search_for_triangulation(const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const int32_t *nodes1, const int32_t *off1,
                                  const int32_t *items1, int nn1, const uint8_t *has_mappoint1, const uint8_t *stereo1,
                                  const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const int32_t *nodes2, const int32_t *off2,
                                  const int32_t *items2, int nn2, const uint8_t *has_mappoint2, const uint8_t *stereo2, int only_stereo,
                                  const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2, int nlevels,
                                  int check_orientation, int32_t *match12, int *nmatches);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:305-370) for a batch of m
 * map points: the observed descriptors of point i are rows off[i]..off[i+1) of desc
 * (bad keyframes already filtered by the caller, :325-331); best[i] = index (within
 * the point's rows) of the descriptor with the least median Hamming distance to the
 * others, first on ties, median = sorted row [int(0.5*(N-1))]; -1 if the point has
 * none.  At most 65,535 observations per point (above 128 a slower row-by-row path). */
int orbm_distinctive_descriptors(const uint8_t *desc, const int32_t *off, int m, int32_t *best);

/* DBoW2 vocabulary-tree descent = the Hamming-heavy half of Frame::ComputeBoW
 * (src/Frame.cc:410-417 -> TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/
 * TemplatedVocabulary.h:1127-1160 batch, :1218-1262 descent; FORB::distance,
 * FORB.cpp:81-101).  The tree is given flat: children of node i =
 * child_ids[child_off[i] .. child_off[i+1]) (node 0 = root), node_desc[nnodes][32],
 * node_word[i] = word id of a leaf (-1 inside), node_weight[i] = its idf weight;
 * L = tree depth (m_L).  transform returns, per feature, the word id, the node at
 * level L - levelsup (the FeatureVector key) and the word weight; BowVector /
 * FeatureVector (std::map insertions, addWeight in feature order, L1 normalise)
 * stay on the host.  The batch form reads descriptor sets resident in HBM
 * (orbx_result_dev layout) and leaves its outputs there. */
typedef struct orbm_vocabulary orbm_vocabulary;
int orbm_vocab_create(const int32_t *child_off, const int32_t *child_ids, const uint8_t *node_desc, const int32_t *node_word,
                      const double *node_weight, int nnodes, int L, orbm_vocabulary **out);
int orbm_vocab_destroy(orbm_vocabulary *v);
int orbm_bow_transform(orbm_vocabulary *v, const uint8_t *features, int n, int levelsup, int32_t *word_id, int32_t *node_id,
                       double *weight);
int orbm_bow_transform_batch_dev(orbm_vocabulary *v, const uint8_t *desc_dev, const int32_t *counts_dev, int cap, int nsets,
                                 int levelsup, int32_t *word_id_dev, int32_t *node_id_dev, void *stream);

/* Acceptance test of ORBmatcher.cc:674-676: best<=th && best<(float)second*nnratio.
 * match12[i] = idx or -1; *nmatches = accepted rows.  Host arrays. */
int orbm_match_filter(int nA, const int32_t *best, const int32_t *second, const int32_t *idx,
                      int th, float nnratio, int32_t *match12, int *nmatches);

/* Batched device form used by the frames/s pipeline: descriptor sets
 * desc_dev[nsets][cap][32] with counts_dev[nsets]; pair p matches set qa[p]
 * (queries) against set qb[p].  Outputs (device): best/second/idx/match12
 * [npairs][cap] int32, nmatch[npairs].  Asynchronous on `stream`.  When the sets come straight from
 * orbx_extract_batch_dev, give both calls the SAME non-NULL stream (or synchronise between them): with NULL the
 * extractor works on its handle's stream and this call on the default stream, and nothing orders the next
 * extraction on that handle behind a match that still reads its descriptors. */
int orbm_match_batch_dev(const uint8_t *desc_dev, const int32_t *counts_dev, int cap,
                         const int32_t *pair_a_dev, const int32_t *pair_b_dev, int npairs,
                         int th, float nnratio,
                         int32_t *best_dev, int32_t *second_dev, int32_t *idx_dev,
                         int32_t *match12_dev, int32_t *nmatch_dev, void *stream);

/* Frame::AssignFeaturesToGrid (src/Frame.cc:245-260, PosInGrid :397-407) as the whole-loop searches lay a frame out: the
 * keypoints that fall inside the 64 x 48 grid (and are not skipped), ordered by (cell = col * 48 + row, keypoint index) --
 * the order in which GetFeaturesInArea visits a window.  Host only, no device call (introspection / test aid: the CPU
 * tests and the host sanitizer build check the counting sort through it).  perm[n] (first *nsorted entries valid),
 * cell_off[64 * 48 + 1]; both may be NULL. */
int orbm_sorted_frame(const orbx_keypoint *kps, int n, const uint8_t *skip, const float *uright, float min_x, float min_y,
                      float max_x, float max_y, int32_t *perm, int32_t *cell_off, int32_t *nsorted);

/* ------------------------------------------------------------------ resident frames
 * A Frame / KeyFrame as the searches need it, kept in HBM between calls: Frame::AssignFeaturesToGrid (src/Frame.cc:245-260)
 * runs once, on the device, when the handle is made -- as the reference builds mGrid once in the Frame constructor -- and
 * every search of that frame (Tracking::TrackWithMotionModel calls SearchByProjection twice, then SearchLocalPoints;
 * src/Tracking.cc) reads the same sorted keypoints, descriptors and cell table.  kps = mvKeysUn, desc = mDescriptors,
 * uright = mvuRight (NULL: monocular, all -1), (min_x .. max_y) = mnMinX .. mnMaxY.  At most 8,192 keypoints.
 * orbm_frame_from_extractor takes keypoints and descriptors of frame `frame` of the extractor's last call where they are, in
 * HBM (no trip over PCIe): xy_undistorted (n x 2 floats, or NULL when the camera has no distortion: mvKeysUn = mvKeys,
 * Frame.cc:270-275) replaces the keypoint coordinates, uright comes from the host or, with uright_from_stereo, from the
 * last orbx_stereo_match on this (left) handle.  A handle is immutable and may be searched from several threads at once;
 * what changes between calls (which keypoints already hold a map point) is an argument of each search. */
typedef struct orbm_frame orbm_frame;
int orbm_frame_create(const orbx_keypoint *kps, const uint8_t *desc, int n, const float *uright, float min_x, float min_y,
                      float max_x, float max_y, orbm_frame **out);
int orbm_frame_from_extractor(orbx_extractor *ex, int frame, const float *xy_undistorted, const float *uright, int uright_from_stereo,
                              float min_x, float min_y, float max_x, float max_y, orbm_frame **out);
int orbm_frame_destroy(orbm_frame *frame);
/* A second handle on the same device data whose SEARCHES use other bounds: a KeyFrame keeps the Frame's grid (mGrid and
 * mfGridElementWidthInv / HeightInv are copied, src/KeyFrame.cc:36) but stores mnMinX .. mnMaxY as int (include/KeyFrame.h:201-204), and
 * KeyFrame::GetFeaturesInArea / IsInImage (src/KeyFrame.cc:613-657) compute with those -- on a camera with distortion the Frame's
 * bounds are fractional and the two differ.  The KeyFrame made from a Frame takes orbm_frame_alias(frame, (int)mnMinX, ...); no copy,
 * no device work; either handle may be destroyed first. */
int orbm_frame_alias(const orbm_frame *frame, float min_x, float min_y, float max_x, float max_y, orbm_frame **out);
/* n = keypoints, nsorted = those inside the grid */
int orbm_frame_size(const orbm_frame *frame, int *n, int *nsorted);
/* introspection (tests): the device-built order as orbm_sorted_frame returns it: perm[nsorted], cell_off[64 * 48 + 1] */
int orbm_frame_layout(const orbm_frame *frame, int32_t *perm, int32_t *cell_off);

/* The host-array searches above, on a resident frame (same results; `skip` / `occupied` by keypoint index as there). */
int orbm_frame_search_window(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, int nq, const uint8_t *skip,
                             int init_dist, int32_t *best, int32_t *best_level, int32_t *second, int32_t *second_level, int32_t *idx);
int orbm_frame_search_fuse(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, int nq,
                           const float *inv_level_sigma2, int nlevels, int32_t *best, int32_t *idx);
int orbm_frame_search_projection(const orbm_frame *frame, const orbm_window_query *queries, const uint8_t *qdesc, const float *qangle,
                                 const uint8_t *qtakes, int nq, const uint8_t *occupied, int th_accept, float nnratio, int ratio_same_level,
                                 int check_orientation, int32_t *match_kp, int32_t *match_q, int *nmatches);
int orbm_frame_search_for_initialization(const orbm_frame *frame2, const orbx_keypoint *kps2, const orbx_keypoint *kps1, const uint8_t *desc1,
                                         int n1, float *prev_matched, int window_size, float nnratio, int check_orientation,
                                         int32_t *matches12, int *nmatches);
int orbm_frame_search_by_projection_map(const orbm_frame *frame, const uint8_t *has_mappoint, const float *mp_pos, const float *mp_normal,
                                        const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc, int m, const double *Rcw,
                                        const double *tcw, const orbm_camera *cam, const float *scale_factors, int nlevels, float th,
                                        float nnratio, int th_reloc, int32_t *matched_mp, int *nmatches, float *proj);

/* orbm_search_by_bow on two resident frames: descriptors and angles stay in HBM -- the first frame's queries are gathered on the
 * device, the second frame's features are addressed by index whether they lie inside its grid or not -- and the call uploads the
 * feature-vector lists only.  The frames' keypoint counts stand for n1 / n2 (valid1[n1], valid2[n2] or NULL, match12[n1],
 * match21[n2] or NULL); arguments and results otherwise as for orbm_search_by_bow. */
int orbm_frame_search_by_bow(const orbm_frame *frame1, const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1,
                             const uint8_t *valid1, const orbm_frame *frame2, const int32_t *nodes2, const int32_t *off2,
                             const int32_t *items2, int nn2, const uint8_t *valid2, int th, int strict_th, float nnratio,
                             int check_orientation, int32_t *match12, int32_t *match21, int *nmatches);

/* orbm_search_for_triangulation on two resident keyframes: a keypoint is "stereo" where the frame's right coordinate is >= 0
 * (mvuRight, as given to orbm_frame_create / taken from ComputeStereoMatches); the call uploads the feature-vector lists and the two
 * "owns a MapPoint" masks, and the rotation histogram runs on angle differences formed on the device.  Positions are the frames'
 * (mvKeysUn) coordinates.  has_mappoint1[n1], has_mappoint2[n2], match12[n1]; otherwise as orbm_search_for_triangulation. */
int orbm_frame_search_for_triangulation(const orbm_frame *kf1, const int32_t *nodes1, const int32_t *off1, const int32_t *items1, int nn1,
                                        const uint8_t *has_mappoint1, const orbm_frame *kf2, const int32_t *nodes2, const int32_t *off2,
                                        const int32_t *items2, int nn2, const uint8_t *has_mappoint2, int only_stereo, const float *F12,
                                        float ex, float ey, const float *scale_factors2, const float *level_sigma2, int nlevels,
                                        int check_orientation, int32_t *match12, int *nmatches);

/* ------------------------------------------- the SearchByProjection forms and SearchBySim3 as WHOLE functions
 * Projection prefix, candidate search, in-loop assignment, acceptance and rotation check in one call, nothing in between
 * returns to the host.  The pointer graph is passed flat: entry i of the vector the reference walks (LastFrame.mvpMapPoints,
 * pKF->GetMapPointMatches(), vpPoints) becomes
 *   valid[i]         what the reference's pointer tests leave: non-NULL and not an outlier (:1555-1558) / not bad and not in
 *                    sAlreadyFound (:1697-1700) / not bad and not in spAlreadyFound (:516-517) / non-NULL, not already
 *                    matched, not bad (:1352-1358)
 *   pos[i][3]        MapPoint::GetWorldPos          desc[i][32]   MapPoint::GetDescriptor
 *   normal[i][3]     GetNormal (Sim3 form)          min_distance / max_distance[i]   mfMinDistance / mfMaxDistance (the 0.8 /
 *                                                   1.2 of Get{Min,Max}DistanceInvariance are applied by the library)
 *   takes[i]         Observations() > 0 (Cur/Last form: :1603-1605; NULL = every assignment blocks its slot)
 *   octave[i]        LastFrame.mvKeys[i].octave (Cur/Last form)
 *   angle[i]         mvKeysUn[i].angle of the source keypoint (rotation check)
 * Arrays a form does not read may be NULL.  orbm_view = calibration + scale pyramid of the searched frame (fx .. cy, mb, mbf,
 * mfLogScaleFactor, mvScaleFactors); the image bounds are the frame handle's.  Poses are 4 x 4 row-major floats as cv::Mat
 * mTcw holds them; cv::Mat arithmetic on them (Rcw, tcw, twc, tlc, Ow, the Sim3 decomposition) is done by the library in the
 * reference's order.  occupied[j] (by keypoint index, NULL = none): the slot is taken before the call -- mvpMapPoints[j] with
 * Observations() > 0 (:1603-1605), mvpMapPoints[j] != NULL (:1741-1742), vpMatched[j] != NULL (:574-575).
 * Outputs as orbm_search_projection: match_kp[n] (entry index, -1 untouched, -2 cleared by the rotation check), match_q[entries],
 * *nmatches; queries_out (optional, test aid) = the GetFeaturesInArea query formed per entry (r < 0: skipped before the search). */
typedef struct orbm_points {
    int32_t n;
    const uint8_t *valid;
    const float *pos, *normal, *min_distance, *max_distance;
    const uint8_t *desc, *takes;
    const int32_t *octave;
    const float *angle;
} orbm_points;
typedef struct orbm_view {
    float fx, fy, cx, cy, mb, mbf, log_scale_factor;
    int32_t nlevels;
    const float *scale_factors;
} orbm_view;
/* Tracking::SearchLocalPoints' data plane (src/Tracking.cc): Frame::isInFrustum(pMP, viewing_cos_limit = 0.5) for every listed
 * point (src/Frame.cc:284-340, with MapPoint::PredictScale) chained into SearchByProjection(Frame &F, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:46-132: RadiusByViewingCos, levels [l-1, l], stereo test, best / second with levels, <= th_high = TH_HIGH,
 * ratio test on equal levels).  valid[i] = the point is not bad and was not already matched in this frame (mnLastFrameSeen);
 * takes[i] = Observations() > 0; occupied[j] = mvpMapPoints[j] holds an observed point (:87-89).  projected_out (optional) =
 * what isInFrustum leaves in the MapPoint (mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos) for the
 * caller's IncreaseVisible bookkeeping. */
int orbm_search_by_projection_points(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const orbm_points *points,
                                     const uint8_t *occupied, float th, float viewing_cos_limit, int th_high, float nnratio, int32_t *match_kp,
                                     int32_t *match_q, int *nmatches, orbm_projected_point *projected_out, orbm_window_query *queries_out);
/* SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)   src/ORBmatcher.cc:1529-1671
 * (th_high = TH_HIGH).  THE per-frame matcher call of Tracking::TrackWithMotionModel. */
int orbm_search_by_projection_last(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const float *Tlw, const orbm_points *last,
                                   const uint8_t *occupied, float th, int mono, int th_high, int check_orientation, int32_t *match_kp,
                                   int32_t *match_q, int *nmatches, orbm_window_query *queries_out);
/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
 * src/ORBmatcher.cc:1673-1800 (relocalisation refinement). */
int orbm_search_by_projection_keyframe(const orbm_frame *cur, const orbm_view *view, const float *Tcw, const orbm_points *kf,
                                       const uint8_t *occupied, float th, int orb_dist, int check_orientation, int32_t *match_kp,
                                       int32_t *match_q, int *nmatches, orbm_window_query *queries_out);
/* SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
 * src/ORBmatcher.cc:491-604 (loop closing; th_low = TH_LOW).  Scw 4 x 4. */
int orbm_search_by_projection_sim3(const orbm_frame *kf, const orbm_view *view, const float *Scw, const orbm_points *points,
                                   const uint8_t *occupied, int th, int th_low, int32_t *match_kp, int32_t *match_q, int *nmatches,
                                   orbm_window_query *queries_out);
/* SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   src/ORBmatcher.cc:1303-1527: both projections, both window searches
 * (levels [l-1, l], best from INT_MAX, <= th_high = TH_HIGH) and the agreement check.  points1 / points2 have one entry per
 * keypoint of kf1 / kf2 (valid folds vbAlreadyMatched1 / 2 in); T1w / T2w = the key frames' poses, R12 3 x 3, t12 3.
 * vnMatch1 / vnMatch2 (optional) as the reference's locals; match12[i1] = idx2 where both directions agree, else -1 (the entries
 * of vpMatches12 the reference overwrites, :1513-1521); *nfound. */
int orbm_search_by_sim3(const orbm_frame *kf1, const orbm_frame *kf2, const orbm_view *view, const float *T1w, const float *T2w, float s12,
                        const float *R12, const float *t12, const orbm_points *points1, const orbm_points *points2, float th, int th_high,
                        int32_t *vnMatch1, int32_t *vnMatch2, int32_t *match12, int *nfound, orbm_window_query *q12_out,
                        orbm_window_query *q21_out);

/* Test aid: on != 0 makes the whole-loop projection searches use the one-wave sequential resolver (k_resolve) instead of the
 * parallel fixed-point resolver (k_resolve_par), which otherwise only takes over when the latter does not converge.  Both
 * give the reference loop's result.  Process-wide; returns the previous setting. */
int orbm_debug_force_sequential_resolver(int on);
/* Iterations the parallel resolver needed in the last whole-loop projection search of this process (-1: it gave up after
 * its iteration limit and the sequential resolver produced the result). */
int orbm_debug_last_resolver_iterations(void);

/* Which kernel the all-pairs matchers (orbm_match_batch_dev, orbm_match_bruteforce) launch.  Both produce the same
 * integers.  ORBM_ALLPAIRS_AUTO (default): a matrix-core kernel (FP4 MFMA computes the selection keys, 4x
 * faster at 2000 x 2000; train tiles expanded once per workgroup and shared through LDS) for sets up to 32768 rows, the
 * XOR + popcount kernel above that; ORBM_ALLPAIRS_POPCOUNT: always the wavefront popcount + LDS-reduction kernel
 * BASELINE.json's north_star describes; ORBM_ALLPAIRS_MFMA: the matrix-core kernel that splits the train tiles over a
 * workgroup's waves (a little faster alone, more vector instructions).  Process-wide; returns the previous setting, or a
 * negative status for an unknown value. */
enum { ORBM_ALLPAIRS_AUTO = 0, ORBM_ALLPAIRS_POPCOUNT = 1, ORBM_ALLPAIRS_MFMA = 2 };
int orbm_set_allpairs_kernel(int kind);

/* HIP-event timing of the all-pairs kernel launched through orbm_match_batch_dev. */
int orbm_profile_enable(int on);
int orbm_profile_read(double *total_ms, int64_t *launches);
/* (orbm_profile_enable: bit 0 = the all-pairs matcher, bit 1 = orbm_bow_transform_batch_dev -- that kernel's own start and end events) */
int orbm_profile_read_bow(double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* ORBSLAM_HIP_H */
