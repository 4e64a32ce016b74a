// orbslam_hip.hpp -- C++ host classes over the C-ABI (header only).
//
// Same names, constructor arguments, getters and call semantics as the
// reference's classes (include/ORBextractor.h:46-112, include/ORBmatcher.h:37-111,
// Thirdparty/g2o/g2o/FEA/include/FEA2.h:94-304), with plain views instead of
// cv::Mat / std::vector<cv::KeyPoint> so the header needs no OpenCV.
// INTEGRATION.md shows the few lines that bind these to the cv:: signatures so
// that Tracking.cc / Frame.cc link unchanged.  Like the reference, nothing here
// throws; `status()` reports the last C-ABI status code.
#ifndef ORBSLAM_HIP_HPP
#define ORBSLAM_HIP_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "fem_hip.h"
#include "orbslam_hip.h"

namespace orbslam_hip {

// A CV_8UC1 image view: what cv::Mat::data / cols / rows / step carry.
struct ImageView {
    const uint8_t *data = nullptr;
    int cols = 0, rows = 0, step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

// One pyramid level on the host (mvImagePyramid[level]).
struct HostImage {
    std::vector<uint8_t> pixels;
    int cols = 0, rows = 0;
    ImageView view() const { return ImageView{pixels.data(), cols, rows, cols}; }
};

typedef orbx_keypoint KeyPoint; // 28-byte cv::KeyPoint layout

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 }; // ORBextractor.h:50

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    {
        orbx_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, /*blur_variant*/ 0, /*trig_variant*/ 0};
        mStatus = orbx_create(&p, &mHandle);
        if (mStatus == ORBX_OK) {
            mnLevels = nlevels;
            mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
            mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
            orbx_get_scale_factors(mHandle, mvScaleFactor.data());
            orbx_get_inv_scale_factors(mHandle, mvInvScaleFactor.data());
            orbx_get_level_sigma2(mHandle, mvLevelSigma2.data());
            orbx_get_inv_level_sigma2(mHandle, mvInvLevelSigma2.data());
            mvImagePyramid.resize(nlevels);
        }
    }
    ~ORBextractor() { orbx_destroy(mHandle); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image; the mask is ignored
    // (ORBextractor.cc:1051-1113).  descriptors: n rows of 32 bytes.
    void operator()(const ImageView &image, const ImageView & /*mask*/, std::vector<KeyPoint> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        if (image.empty()) return; // :1054-1055: outputs untouched
        mStatus = orbx_reserve(mHandle, image.cols, image.rows, 1);   // the capacity depends on the frame's aspect ratio (tiny quotas)
        if (mStatus != ORBX_OK) { keypoints.clear(); descriptors.clear(); return; }
        const int cap = orbx_keypoint_capacity(mHandle);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        mStatus = orbx_extract(mHandle, image.data, image.cols, image.rows, image.step, keypoints.data(),
                               descriptors.data(), cap, &n);
        if (mStatus != ORBX_OK) n = 0;
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32); // n == 0: descriptors.release() (:1072-1073)
        mPyramidStale = true;
    }

    void MarkPyramidStale() { mPyramidStale = true; }   // after a call that extracted through the handle (ExtractPair)
    int GetLevels() const { return mnLevels; }
    float GetScaleFactor() const { return mnLevels > 1 ? mvScaleFactor[1] : 1.0f; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() const { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() const { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }

    // Public in the reference (ORBextractor.h:86) and read by
    // Frame::ComputeStereoMatches: call SyncImagePyramid() before reading it.
    std::vector<HostImage> mvImagePyramid;
    void SyncImagePyramid()
    {
        if (!mPyramidStale) return;
        for (int l = 0; l < mnLevels; ++l) {
            int w = 0, h = 0;
            if (orbx_level_size(mHandle, l, &w, &h) != ORBX_OK) return;
            HostImage &im = mvImagePyramid[l];
            im.cols = w; im.rows = h; im.pixels.resize((size_t)w * h);
            mStatus = orbx_pyramid_level(mHandle, 0, l, im.pixels.data(), w);
        }
        mPyramidStale = false;
    }

    int status() const { return mStatus; }
    orbx_extractor *handle() { return mHandle; }

private:
    orbx_extractor *mHandle = nullptr;
    int mnLevels = 0, mStatus = ORBX_OK;
    bool mPyramidStale = true;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// The two ExtractORB calls of a stereo frame (Frame.cc:78-81: two threads, then join) from one thread: both images' kernel chains
// are enqueued before the host waits for either (orbx_extract_pair).  Outputs as from left(...) and right(...).
inline int ExtractPair(ORBextractor &left, const ImageView &image_left, std::vector<KeyPoint> &keys_left, std::vector<uint8_t> &desc_left,
                       ORBextractor &right, const ImageView &image_right, std::vector<KeyPoint> &keys_right, std::vector<uint8_t> &desc_right)
{
    if (image_left.empty() || image_right.empty() || image_left.cols != image_right.cols || image_left.rows != image_right.rows ||
        image_left.step != image_right.step)
        return ORBX_ERR_ARG;
    int rc = orbx_reserve(left.handle(), image_left.cols, image_left.rows, 1);
    if (rc == ORBX_OK) rc = orbx_reserve(right.handle(), image_right.cols, image_right.rows, 1);
    if (rc != ORBX_OK) return rc;
    const int capl = orbx_keypoint_capacity(left.handle()), capr = orbx_keypoint_capacity(right.handle());
    keys_left.resize(capl); desc_left.resize((size_t)capl * 32); keys_right.resize(capr); desc_right.resize((size_t)capr * 32);
    int nl = 0, nr = 0;
    rc = orbx_extract_pair(left.handle(), image_left.data, right.handle(), image_right.data, image_left.cols, image_left.rows, image_left.step,
                           keys_left.data(), desc_left.data(), capl, &nl, keys_right.data(), desc_right.data(), capr, &nr);
    if (rc != ORBX_OK) nl = nr = 0;
    keys_left.resize(nl); desc_left.resize((size_t)nl * 32); keys_right.resize(nr); desc_right.resize((size_t)nr * 32);
    left.MarkPyramidStale(); right.MarkPyramidStale();
    return rc;
}

// Frame::ComputeStereoMatches (src/Frame.cc:527-701) on the results the two
// extractors still hold on the device after operator() ran on the left / right image.
inline int ComputeStereoMatches(ORBextractor &left, ORBextractor &right, float mb, float mbf,
                                std::vector<float> &mvuRight, std::vector<float> &mvDepth)
{
    const int cap = orbx_keypoint_capacity(left.handle());
    mvuRight.assign(cap, -1.0f);
    mvDepth.assign(cap, -1.0f);
    int n = 0;
    int rc = orbx_stereo_match(left.handle(), right.handle(), mb, mbf, nullptr);
    if (rc == ORBX_OK) rc = orbx_stereo_download(left.handle(), 0, mvuRight.data(), mvDepth.data(), cap, &n);
    mvuRight.resize(rc == ORBX_OK ? n : 0);
    mvDepth.resize(rc == ORBX_OK ? n : 0);
    return rc;
}

namespace DBoW2 {
typedef std::map<unsigned int, double> BowVector;                          // DBoW2/BowVector.h:57
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;    // DBoW2/FeatureVector.h:23
}

class ORBmatcher {
public:
    static const int TH_LOW = 45, TH_HIGH = 95, TH_RELOC = 60, HISTO_LENGTH = 30; // ORBmatcher.cc:37-40

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // ORBmatcher::DescriptorDistance(a, b), a/b = 32-byte rows.
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b)
    {
        uint16_t d = 0;
        return orbm_hamming_matrix(a, 1, b, 1, &d) == ORBX_OK ? (int)d : -1;
    }

    // The selection loop of every Search* function over gated candidates
    // (candidate order = GetFeaturesInArea / BoW-member order), then the
    // acceptance test bestDist<=th && bestDist<bestDist2*mfNNratio.
    // Returns the number of accepted matches; vnMatches12[i] = index in B or -1.
    int MatchCandidates(const uint8_t *A, int nA, const uint8_t *B, int nB, const std::vector<int32_t> &candOff,
                        const std::vector<int32_t> &candIdx, int th, std::vector<int32_t> &vnMatches12,
                        std::vector<int32_t> *bestDist = nullptr)
    {
        std::vector<int32_t> best(nA), second(nA), idx(nA);
        vnMatches12.assign(nA, -1);
        if (orbm_match_candidates(A, nA, B, nB, candOff.data(), candIdx.data(), best.data(), second.data(), idx.data()) != ORBX_OK)
            return 0;
        int n = 0;
        orbm_match_filter(nA, best.data(), second.data(), idx.data(), th, mfNNratio, vnMatches12.data(), &n);
        if (bestDist) *bestDist = best;
        return n;
    }

    // Which kernel MatchBruteForce (and orbm_match_batch_dev) launch: ORBM_ALLPAIRS_AUTO / _POPCOUNT / _MFMA; same results.
    static int SetAllPairsKernel(int kind) { return orbm_set_allpairs_kernel(kind); }

    int MatchBruteForce(const uint8_t *A, int nA, const uint8_t *B, int nB, int th, std::vector<int32_t> &vnMatches12)
    {
        std::vector<int32_t> best(nA), second(nA), idx(nA);
        vnMatches12.assign(nA, -1);
        if (orbm_match_bruteforce(A, nA, B, nB, best.data(), second.data(), idx.data()) != ORBX_OK) return 0;
        int n = 0;
        orbm_match_filter(nA, best.data(), second.data(), idx.data(), th, mfNNratio, vnMatches12.data(), &n);
        return n;
    }

    // What the whole-loop searches read of a Frame / KeyFrame (src/Frame.h): mvKeysUn, mDescriptors (N x 32),
    // mvuRight (may be null: monocular), the grid bounds mnMinX..mnMaxY.
    struct FrameView {
        const orbx_keypoint *mvKeysUn = nullptr;
        const uint8_t *mDescriptors = nullptr;
        int N = 0;
        const float *mvuRight = nullptr;
        float mnMinX = 0.f, mnMinY = 0.f, mnMaxX = 0.f, mnMaxY = 0.f;
    };
    // One projected map point of the SearchByProjection loops: window (u, v, r), level range, xr = u - bf/z,
    // MapPoint::GetDescriptor(), the angle of its source keypoint, and whether the pointer it leaves in
    // mvpMapPoints makes later points skip that keypoint (Observations() > 0, or always).
    struct ProjectedPoint {
        orbm_window_query window;
        const uint8_t *descriptor;
        float angle;
        bool blocksSlot;
    };

    // ORBmatcher::SearchByProjection loops (ORBmatcher.cc:46-132, :491-604, :1529-1671, :1673-1800) over points
    // the caller has projected, in the caller's order.  occupied[j]: keypoint j is skipped from the start.
    // slotOwner[j] = point that ends in mvpMapPoints[j] (-1 untouched, -2 set to NULL by the rotation check).
    int SearchByProjection(const FrameView &F, const std::vector<ProjectedPoint> &points, const std::vector<uint8_t> &occupied,
                           int thAccept, bool ratioOnSameLevel, std::vector<int32_t> &slotOwner)
    {
        const int nq = (int)points.size();
        std::vector<orbm_window_query> q(nq);
        std::vector<uint8_t> qd((size_t)32 * nq), tk(nq);
        std::vector<float> qa(nq);
        for (int i = 0; i < nq; ++i) {
            q[i] = points[i].window; qa[i] = points[i].angle; tk[i] = points[i].blocksSlot;
            std::copy(points[i].descriptor, points[i].descriptor + 32, &qd[(size_t)32 * i]);
        }
        slotOwner.assign(F.N, -1);
        std::vector<int32_t> mq(nq);
        int nm = 0;
        mStatus = orbm_search_projection(q.data(), qd.data(), qa.data(), tk.data(), nq, F.mvKeysUn, F.mDescriptors, F.N,
                                         occupied.empty() ? nullptr : occupied.data(), F.mvuRight, F.mnMinX, F.mnMinY, F.mnMaxX,
                                         F.mnMaxY, thAccept, mfNNratio, ratioOnSameLevel, mbCheckOrientation, slotOwner.data(),
                                         mq.data(), &nm);
        return mStatus == ORBX_OK ? nm : 0;
    }

    // ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:606-721);
    // vbPrevMatched as x0, y0, x1, y1, ...
    int SearchForInitialization(const FrameView &F1, const FrameView &F2, std::vector<float> &vbPrevMatched,
                                std::vector<int32_t> &vnMatches12, int windowSize = 10)
    {
        vnMatches12.assign(F1.N, -1);
        int nm = 0;
        mStatus = orbm_search_for_initialization(F1.mvKeysUn, F1.mDescriptors, F1.N, F2.mvKeysUn, F2.mDescriptors, F2.N,
                                                 vbPrevMatched.data(), F2.mnMinX, F2.mnMinY, F2.mnMaxX, F2.mnMaxY, windowSize,
                                                 mfNNratio, mbCheckOrientation, vnMatches12.data(), &nm);
        return mStatus == ORBX_OK ? nm : 0;
    }

    // Candidate loop of ORBmatcher::Fuse (ORBmatcher.cc:1092-1146): best keypoint per projected map point under the
    // reprojection-error gate; invLevelSigma2 == nullptr: no gate (the Sim3 form).  The map update stays with the caller.
    int FuseCandidates(const FrameView &KF, const std::vector<ProjectedPoint> &points, const float *invLevelSigma2, int nLevels,
                       std::vector<int32_t> &bestDist, std::vector<int32_t> &bestIdx)
    {
        const int nq = (int)points.size();
        std::vector<orbm_window_query> q(nq);
        std::vector<uint8_t> qd((size_t)32 * nq);
        for (int i = 0; i < nq; ++i) { q[i] = points[i].window; std::copy(points[i].descriptor, points[i].descriptor + 32, &qd[(size_t)32 * i]); }
        bestDist.assign(nq, 256); bestIdx.assign(nq, -1);
        mStatus = orbm_search_fuse(q.data(), qd.data(), nq, KF.mvKeysUn, KF.mDescriptors, KF.N, KF.mvuRight, invLevelSigma2, nLevels,
                                   KF.mnMinX, KF.mnMinY, KF.mnMaxX, KF.mnMaxY, bestDist.data(), bestIdx.data());
        return mStatus;
    }

    // What the projection kernels read of a Frame / KeyFrame pose (src/Frame.cc:270-282, KeyFrame::GetRotation ...):
    // mRcw (row-major), mtcw, mOw as the float cv::Mat members hold them, the intrinsics, the grid bounds
    // mnMinX .. mnMaxY (in cam.grid_*), mbf, mfLogScaleFactor, mvScaleFactors.
    struct PoseView {
        float Rcw[9], tcw[3], Ow[3];
        orbm_camera cam;
        float mbf = 0.f, mfLogScaleFactor = 0.f;
        const float *mvScaleFactors = nullptr;
        const float *mvInvLevelSigma2 = nullptr;
        int mnScaleLevels = 0;
    };
    // Map points gathered by the caller from a vector<MapPoint*>: GetWorldPos(), GetNormal(), mfMinDistance,
    // mfMaxDistance, GetDescriptor(), one entry per list position (entries of NULL pointers are never read back).
    struct MapPointArrays {
        const float *pos = nullptr, *normal = nullptr, *mfMinDistance = nullptr, *mfMaxDistance = nullptr;
        const uint8_t *descriptors = nullptr;
        int n = 0;
    };

    // Frame::isInFrustum + MapPoint::PredictScale for all points at once (Frame.cc:284-340, MapPoint.cc:464-480):
    // out[i] = {mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos, dist, mnTrackScaleLevel, mbTrackInView}, windows[i] =
    // the GetFeaturesInArea query of SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:62-70).
    int ProjectInFrustum(const PoseView &F, const MapPointArrays &mps, float viewingCosLimit, float th,
                         std::vector<orbm_projected_point> &out, std::vector<orbm_window_query> &windows)
    {
        out.resize(mps.n); windows.resize(mps.n);
        mStatus = orbm_project_points(ORBM_PROJECT_FRUSTUM, mps.pos, mps.normal, mps.mfMinDistance, mps.mfMaxDistance, mps.n, F.Rcw,
                                      F.tcw, F.Ow, &F.cam, F.mbf, viewingCosLimit, F.mfLogScaleFactor, F.mvScaleFactors,
                                      F.mnScaleLevels, th, out.data(), windows.data());
        return mStatus;
    }

    // ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) as a whole (ORBmatcher.cc:1026-1176): projection of every
    // listed point (:1053-1094) and the gated candidate loop (:1092-1146) on the device, then the loop's skips (:1046-1053)
    // and its map update (:1149-1170) on the host in the list's order -- the candidate search of a point does not depend
    // on the map state, the skips and the update do.  `map` supplies the pointer-graph side through handles H (e.g.
    // MapPoint*): H at(int i) (vpMapPoints[i], a null handle for NULL), bool null(H), bool isBad(H), bool
    // isInKeyFrame(H), H slotOwner(int idx) (pKF->GetMapPoint), int observations(H), void replace(H dead, H heir)
    // (dead->Replace(heir)), void add(H, int idx) (AddObservation + AddMapPoint).
    template <class MapOps>
    int Fuse(const FrameView &KF, const PoseView &pose, const MapPointArrays &mps, float th, MapOps &map)
    {
        std::vector<orbm_projected_point> proj;
        std::vector<int32_t> best, idx;
        if (fuseCandidates(KF, pose, mps, th, false, proj, best, idx) != ORBX_OK) return 0;
        return fuseReplay(proj, best, idx, map);
    }
    // ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, vpReplacePoint) (ORBmatcher.cc:1178-1301):
    // pose = the Sim3 decomposed as :1188-1192 do; `map` needs at / null / isBad / slotOwner / add as above, plus
    // bool alreadyFound(H) (spAlreadyFound.count(pMP): the SNAPSHOT of pKF->GetMapPoints() from before the call) and
    // void recordReplace(int i, H) (vpReplacePoint[i] = pMPinKF); it never replaces.
    template <class MapOps>
    int FuseSim3(const FrameView &KF, const PoseView &pose, const MapPointArrays &mps, float th, MapOps &map)
    {
        std::vector<orbm_projected_point> proj;
        std::vector<int32_t> best, idx;
        if (fuseCandidates(KF, pose, mps, th, true, proj, best, idx) != ORBX_OK) return 0;
        return fuseReplaySim3(proj, best, idx, map);
    }
    int fuseCandidates(const FrameView &KF, const PoseView &pose, const MapPointArrays &mps, float th, bool sim3,
                       std::vector<orbm_projected_point> &proj, std::vector<int32_t> &best, std::vector<int32_t> &idx)
    {
        proj.resize(mps.n);
        std::vector<orbm_window_query> q(mps.n);
        mStatus = orbm_project_points(sim3 ? ORBM_PROJECT_FUSE_SIM3 : ORBM_PROJECT_FUSE, mps.pos, mps.normal, mps.mfMinDistance,
                                      mps.mfMaxDistance, mps.n, pose.Rcw, pose.tcw, pose.Ow, &pose.cam, pose.mbf, 0.f,
                                      pose.mfLogScaleFactor, pose.mvScaleFactors, pose.mnScaleLevels, th, proj.data(), q.data());
        if (mStatus != ORBX_OK) return mStatus;
        best.assign(mps.n, 256); idx.assign(mps.n, -1);
        mStatus = orbm_search_fuse(q.data(), mps.descriptors, mps.n, KF.mvKeysUn, KF.mDescriptors, KF.N, KF.mvuRight,
                                   sim3 ? nullptr : pose.mvInvLevelSigma2, pose.mnScaleLevels, KF.mnMinX, KF.mnMinY, KF.mnMaxX,
                                   KF.mnMaxY, best.data(), idx.data());
        return mStatus;
    }

    // Tail of the Fuse loop, ORBmatcher.cc:1046-1053 and :1149-1170, in list order.
    template <class MapOps>
    static int fuseReplay(const std::vector<orbm_projected_point> &proj, const std::vector<int32_t> &best,
                          const std::vector<int32_t> &idx, MapOps &map)
    {
        int nFused = 0;
        for (int i = 0; i < (int)proj.size(); ++i) {
            auto pMP = map.at(i);
            if (map.null(pMP)) continue;
            if (map.isBad(pMP) || map.isInKeyFrame(pMP)) continue;
            if (!proj[i].visible || idx[i] < 0) continue;      // projection tests failed / vIndices.empty() / no candidate
            if (best[i] <= TH_LOW) {
                auto pMPinKF = map.slotOwner(idx[i]);
                if (!map.null(pMPinKF)) {
                    if (!map.isBad(pMPinKF)) {
                        if (map.observations(pMPinKF) > map.observations(pMP)) map.replace(pMP, pMPinKF);
                        else map.replace(pMPinKF, pMP);
                    }
                } else {
                    map.add(pMP, idx[i]);
                }
                ++nFused;
            }
        }
        return nFused;
    }

    // Tail of the Sim3 form, ORBmatcher.cc:1194-1205 and :1279-1296: the skips read spAlreadyFound, a snapshot of
    // pKF->GetMapPoints() taken before the loop -- map.alreadyFound(H) must answer from that snapshot, not from the
    // state the loop is changing (a point the loop adds is processed again when the list repeats it); a taken slot is
    // only recorded (vpReplacePoint[iMP] = pMPinKF: map.recordReplace(i, pMPinKF)), a free one gets the point.
    template <class MapOps>
    static int fuseReplaySim3(const std::vector<orbm_projected_point> &proj, const std::vector<int32_t> &best,
                              const std::vector<int32_t> &idx, MapOps &map)
    {
        int nFused = 0;
        for (int i = 0; i < (int)proj.size(); ++i) {
            auto pMP = map.at(i);
            if (map.null(pMP) || map.isBad(pMP) || map.alreadyFound(pMP)) continue;
            if (!proj[i].visible || idx[i] < 0) continue;
            if (best[i] <= TH_LOW) {
                auto pMPinKF = map.slotOwner(idx[i]);
                if (!map.null(pMPinKF)) { if (!map.isBad(pMPinKF)) map.recordReplace(i, pMPinKF); }
                else map.add(pMP, idx[i]);
                ++nFused;
            }
        }
        return nFused;
    }

    // ORBmatcher::SearchBySim3 (ORBmatcher.cc:1303-1527) over projected points: points12[i1] = map point i1 of KF1 seen
    // in KF2 (window.r < 0: skipped), points21 the other way; levels [l-1, l], <= TH_HIGH, then the agreement check.
    int SearchBySim3(const FrameView &KF1, const FrameView &KF2, const std::vector<ProjectedPoint> &points12,
                     const std::vector<ProjectedPoint> &points21, std::vector<int32_t> &vnMatches12)
    {
        auto one_way = [&](const FrameView &dst, const std::vector<ProjectedPoint> &pts, std::vector<int32_t> &m) {
            const int nq = (int)pts.size();
            std::vector<orbm_window_query> q(nq);
            std::vector<uint8_t> qd((size_t)32 * nq);
            for (int i = 0; i < nq; ++i) { q[i] = pts[i].window; std::copy(pts[i].descriptor, pts[i].descriptor + 32, &qd[(size_t)32 * i]); }
            std::vector<int32_t> best(nq), bl(nq), second(nq), sl(nq), idx(nq);
            mStatus = orbm_search_window(q.data(), qd.data(), nq, dst.mvKeysUn, dst.mDescriptors, dst.N, nullptr, nullptr, dst.mnMinX,
                                         dst.mnMinY, dst.mnMaxX, dst.mnMaxY, INT32_MAX, best.data(), bl.data(), second.data(), sl.data(),
                                         idx.data());
            m.assign(nq, -1);
            for (int i = 0; i < nq && mStatus == ORBX_OK; ++i)
                if (idx[i] >= 0 && best[i] <= TH_HIGH) m[i] = idx[i];
        };
        std::vector<int32_t> m1, m2;
        one_way(KF2, points12, m1);
        if (mStatus == ORBX_OK) one_way(KF1, points21, m2);
        vnMatches12.assign(points12.size(), -1);
        int nFound = 0;
        for (size_t i1 = 0; i1 < m1.size() && mStatus == ORBX_OK; ++i1)
            if (m1[i1] >= 0 && m1[i1] < (int32_t)m2.size() && m2[m1[i1]] == (int32_t)i1) { vnMatches12[i1] = m1[i1]; ++nFound; }
        return nFound;
    }

    // ---- resident frames and the projection searches as whole functions ----------------------------------------------------
    // A Frame / KeyFrame kept in HBM between searches (orbm_frame): Frame::AssignFeaturesToGrid runs once, on the device.
    // Holds no reference to the arrays it was made from.  Movable, not copyable; any number of searches may read it at once.
    class ResidentFrame {
    public:
        ResidentFrame() = default;
        explicit ResidentFrame(const FrameView &F) { mStatus = orbm_frame_create(F.mvKeysUn, F.mDescriptors, F.N, F.mvuRight, F.mnMinX, F.mnMinY, F.mnMaxX, F.mnMaxY, &mH); sync(); }
        // straight from the extractor's device results (no trip over PCIe): xyUndistorted = mvKeysUn coordinates (x0, y0, ...) or
        // null when the camera has no distortion; uRight from the host or from the last ComputeStereoMatches on `ex`
        ResidentFrame(ORBextractor &ex, const float *xyUndistorted, const float *uRight, bool uRightFromStereo, float mnMinX, float mnMinY,
                      float mnMaxX, float mnMaxY)
        {
            mStatus = orbm_frame_from_extractor(ex.handle(), 0, xyUndistorted, uRight, uRightFromStereo, mnMinX, mnMinY, mnMaxX, mnMaxY, &mH);
            sync();
        }
        ResidentFrame(ResidentFrame &&o) noexcept : mH(o.mH), N(o.N), mStatus(o.mStatus) { o.mH = nullptr; }
        ResidentFrame &operator=(ResidentFrame &&o) noexcept { if (this != &o) { reset(); mH = o.mH; N = o.N; mStatus = o.mStatus; o.mH = nullptr; } return *this; }
        ResidentFrame(const ResidentFrame &) = delete;
        ResidentFrame &operator=(const ResidentFrame &) = delete;
        ~ResidentFrame() { reset(); }
        void reset() { if (mH) orbm_frame_destroy(mH); mH = nullptr; }
        const orbm_frame *handle() const { return mH; }
        int status() const { return mStatus; }
        int N = 0;
    private:
        void sync() { if (mStatus == ORBX_OK) orbm_frame_size(mH, &N, nullptr); else mH = nullptr; }
        orbm_frame *mH = nullptr;
        int mStatus = ORBX_OK;
    };
    // The flat form of a vector<MapPoint*> (orbm_points; see include/orbslam_hip.h for which form reads what): the binding fills it
    // in one pass over the pointer vector.
    struct PointList {
        std::vector<uint8_t> valid, desc, takes;
        std::vector<float> pos, normal, minDistance, maxDistance, angle;
        std::vector<int32_t> octave;
        void resize(size_t n, bool withRange, bool withNormal, bool withOctave)
        {
            valid.assign(n, 0); desc.resize(32 * n); takes.assign(n, 1); pos.resize(3 * n); angle.assign(n, 0.f);
            if (withRange) { minDistance.resize(n); maxDistance.resize(n); }
            if (withNormal) normal.resize(3 * n);
            if (withOctave) octave.assign(n, 0);
        }
        orbm_points view() const
        {
            orbm_points p;
            p.n = (int32_t)valid.size(); p.valid = valid.data(); p.pos = pos.data(); p.normal = normal.empty() ? nullptr : normal.data();
            p.min_distance = minDistance.empty() ? nullptr : minDistance.data(); p.max_distance = maxDistance.empty() ? nullptr : maxDistance.data();
            p.desc = desc.data(); p.takes = takes.empty() ? nullptr : takes.data(); p.octave = octave.empty() ? nullptr : octave.data();
            p.angle = angle.empty() ? nullptr : angle.data();
            return p;
        }
    };
    // Calibration + scale pyramid of the searched frame (Frame::fx .. mbf, mfLogScaleFactor, mvScaleFactors).
    struct Calibration {
        float fx = 0, fy = 0, cx = 0, cy = 0, mb = 0, mbf = 0, mfLogScaleFactor = 0;
        std::vector<float> mvScaleFactors;
        orbm_view view() const { return orbm_view{fx, fy, cx, cy, mb, mbf, mfLogScaleFactor, (int32_t)mvScaleFactors.size(), mvScaleFactors.data()}; }
    };
    // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) (ORBmatcher.cc:1529-1671),
    // whole.  Tcw / Tlw = mTcw of the two frames (4 x 4, row-major).  slotOwner[j] = entry of `last` whose point ends in
    // CurrentFrame.mvpMapPoints[j] (-1 untouched, -2 set to NULL by the rotation check); returns nmatches.
    int SearchByProjection(const ResidentFrame &Cur, const Calibration &K, const float *Tcw, const float *Tlw, const PointList &last,
                           const std::vector<uint8_t> &occupied, float th, bool bMono, std::vector<int32_t> &slotOwner)
    {
        const orbm_points p = last.view(); const orbm_view v = K.view();
        slotOwner.assign(Cur.N, -1); mScratch.resize(p.n);
        int nm = 0;
        mStatus = orbm_search_by_projection_last(Cur.handle(), &v, Tcw, Tlw, &p, occupied.empty() ? nullptr : occupied.data(), th, bMono, TH_HIGH,
                                                 mbCheckOrientation, slotOwner.data(), mScratch.data(), &nm, nullptr);
        return mStatus == ORBX_OK ? nm : 0;
    }
    // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist) (:1673-1800), whole.
    int SearchByProjection(const ResidentFrame &Cur, const Calibration &K, const float *Tcw, const PointList &kf,
                           const std::vector<uint8_t> &occupied, float th, int ORBdist, std::vector<int32_t> &slotOwner)
    {
        const orbm_points p = kf.view(); const orbm_view v = K.view();
        slotOwner.assign(Cur.N, -1); mScratch.resize(p.n);
        int nm = 0;
        mStatus = orbm_search_by_projection_keyframe(Cur.handle(), &v, Tcw, &p, occupied.empty() ? nullptr : occupied.data(), th, ORBdist,
                                                     mbCheckOrientation, slotOwner.data(), mScratch.data(), &nm, nullptr);
        return mStatus == ORBX_OK ? nm : 0;
    }
    // SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
    // (:491-604), whole.  slotOwner[j] = list entry written to vpMatched[j] or -1.
    int SearchByProjection(const ResidentFrame &KF, const Calibration &K, const float *Scw, const PointList &points,
                           const std::vector<uint8_t> &occupied, int th, std::vector<int32_t> &slotOwner)
    {
        const orbm_points p = points.view(); const orbm_view v = K.view();
        slotOwner.assign(KF.N, -1); mScratch.resize(p.n);
        int nm = 0;
        mStatus = orbm_search_by_projection_sim3(KF.handle(), &v, Scw, &p, occupied.empty() ? nullptr : occupied.data(), th, TH_LOW,
                                                 slotOwner.data(), mScratch.data(), &nm, nullptr);
        return mStatus == ORBX_OK ? nm : 0;
    }
    // Tracking::SearchLocalPoints' data plane: isInFrustum for every listed point + SearchByProjection(Frame&, vector<MapPoint*>&, th)
    // (:46-132), whole.  projected[i] = what isInFrustum leaves in the MapPoint.
    int SearchLocalPoints(const ResidentFrame &Cur, const Calibration &K, const float *Tcw, const PointList &points,
                          const std::vector<uint8_t> &occupied, float th, std::vector<int32_t> &slotOwner,
                          std::vector<orbm_projected_point> *projected = nullptr)
    {
        const orbm_points p = points.view(); const orbm_view v = K.view();
        slotOwner.assign(Cur.N, -1); mScratch.resize(p.n);
        if (projected) projected->resize(p.n);
        int nm = 0;
        mStatus = orbm_search_by_projection_points(Cur.handle(), &v, Tcw, &p, occupied.empty() ? nullptr : occupied.data(), th, 0.5f, TH_HIGH,
                                                   mfNNratio, slotOwner.data(), mScratch.data(), &nm, projected ? projected->data() : nullptr, nullptr);
        return mStatus == ORBX_OK ? nm : 0;
    }
    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (:1303-1527), whole: vnMatches12[i1] = idx2 where both directions
    // agree (the entries of vpMatches12 the reference overwrites), else -1; returns nFound.
    int SearchBySim3(const ResidentFrame &KF1, const ResidentFrame &KF2, const Calibration &K, const float *T1w, const float *T2w, float s12,
                     const float *R12, const float *t12, const PointList &points1, const PointList &points2, float th,
                     std::vector<int32_t> &vnMatches12)
    {
        const orbm_points p1 = points1.view(), p2 = points2.view(); const orbm_view v = K.view();
        vnMatches12.assign(KF1.N, -1);
        int nf = 0;
        mStatus = orbm_search_by_sim3(KF1.handle(), KF2.handle(), &v, T1w, T2w, s12, R12, t12, &p1, &p2, th, TH_HIGH, nullptr, nullptr,
                                      vnMatches12.data(), &nf, nullptr, nullptr);
        return mStatus == ORBX_OK ? nf : 0;
    }

    // DBoW2::FeatureVector flattened in std::map order.
    struct FeatureVector {
        std::vector<int32_t> nodes, off, items;
        // from the per-feature node ids of orbm_bow_transform; keep[i] == 0: feature i is not in the vector
        static FeatureVector FromNodeIds(const std::vector<int32_t> &nodeId, const std::vector<uint8_t> *keep = nullptr)
        {
            std::vector<int32_t> idx;
            for (int32_t i = 0; i < (int32_t)nodeId.size(); ++i)
                if (!keep || (*keep)[i]) idx.push_back(i);
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return nodeId[a] < nodeId[b]; });
            FeatureVector fv;
            for (size_t k = 0; k < idx.size(); ++k) {
                if (k == 0 || nodeId[idx[k]] != nodeId[idx[k - 1]]) { fv.nodes.push_back(nodeId[idx[k]]); fv.off.push_back((int32_t)k); }
                fv.items.push_back(idx[k]);
            }
            fv.off.push_back((int32_t)idx.size());
            return fv;
        }
        // from the map ORBVocabulary::transform fills (Frame::mFeatVec / KeyFrame::mFeatVec)
        static FeatureVector FromMap(const DBoW2::FeatureVector &m)
        {
            FeatureVector fv;
            for (const auto &e : m) {
                fv.nodes.push_back((int32_t)e.first); fv.off.push_back((int32_t)fv.items.size());
                for (unsigned int i : e.second) fv.items.push_back((int32_t)i);
            }
            fv.off.push_back((int32_t)fv.items.size());
            return fv;
        }
    };

    // ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:360-489; valid2 == nullptr) and
    // SearchByBoW(KeyFrame*, KeyFrame*, ...) (:723-856).  valid = "owns a good MapPoint".
    int SearchByBoW(const FeatureVector &fv1, const std::vector<uint8_t> &valid1, const FrameView &KF1, const FeatureVector &fv2,
                    const std::vector<uint8_t> *valid2, const FrameView &F2, std::vector<int32_t> &vnMatches12)
    {
        std::vector<float> a1(KF1.N), a2(F2.N);
        for (int i = 0; i < KF1.N; ++i) a1[i] = KF1.mvKeysUn[i].angle;
        for (int i = 0; i < F2.N; ++i) a2[i] = F2.mvKeysUn[i].angle;
        vnMatches12.assign(KF1.N, -1);
        int nm = 0;
        mStatus = orbm_search_by_bow(fv1.nodes.data(), fv1.off.data(), fv1.items.data(), (int)fv1.nodes.size(), valid1.data(),
                                     KF1.mDescriptors, a1.data(), KF1.N, fv2.nodes.data(), fv2.off.data(), fv2.items.data(),
                                     (int)fv2.nodes.size(), valid2 ? valid2->data() : nullptr, F2.mDescriptors, a2.data(), F2.N,
                                     TH_LOW, valid2 ? 1 : 0, mfNNratio, mbCheckOrientation, vnMatches12.data(), nullptr, &nm);
        return mStatus == ORBX_OK ? nm : 0;
    }

    // The same on two resident frames: nothing but the feature-vector lists travels with the call.
    int SearchByBoW(const FeatureVector &fv1, const std::vector<uint8_t> &valid1, const ResidentFrame &KF1, const FeatureVector &fv2,
                    const std::vector<uint8_t> *valid2, const ResidentFrame &F2, std::vector<int32_t> &vnMatches12)
    {
        vnMatches12.assign(KF1.N, -1);
        int nm = 0;
        mStatus = orbm_frame_search_by_bow(KF1.handle(), fv1.nodes.data(), fv1.off.data(), fv1.items.data(), (int)fv1.nodes.size(),
                                           valid1.data(), F2.handle(), fv2.nodes.data(), fv2.off.data(), fv2.items.data(),
                                           (int)fv2.nodes.size(), valid2 ? valid2->data() : nullptr, TH_LOW, valid2 ? 1 : 0, mfNNratio,
                                           mbCheckOrientation, vnMatches12.data(), nullptr, &nm);
        return mStatus == ORBX_OK ? nm : 0;
    }

    // SearchForTriangulation on two resident keyframes (stereo = the frame's right coordinate >= 0): the lists and the two
    // "owns a MapPoint" masks are all that travels.
    int SearchForTriangulation(const ResidentFrame &KF1, const FeatureVector &fv1, const std::vector<uint8_t> &hasMP1,
                               const ResidentFrame &KF2, const FeatureVector &fv2, const std::vector<uint8_t> &hasMP2, const float F12[9],
                               float ex, float ey, bool bOnlyStereo, const std::vector<float> &scaleFactors2,
                               const std::vector<float> &levelSigma2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs)
    {
        vMatchedPairs.clear();
        std::vector<int32_t> m12(KF1.N > 0 ? KF1.N : 1, -1);
        int nm = 0;
        mStatus = orbm_frame_search_for_triangulation(KF1.handle(), fv1.nodes.data(), fv1.off.data(), fv1.items.data(), (int)fv1.nodes.size(),
                                                      hasMP1.data(), KF2.handle(), fv2.nodes.data(), fv2.off.data(), fv2.items.data(),
                                                      (int)fv2.nodes.size(), hasMP2.data(), bOnlyStereo ? 1 : 0, F12, ex, ey,
                                                      scaleFactors2.data(), levelSigma2.data(), (int)scaleFactors2.size(),
                                                      mbCheckOrientation, m12.data(), &nm);
        if (mStatus != ORBX_OK) return 0;
        for (int i = 0; i < KF1.N; ++i)
            if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);
        return (int)vMatchedPairs.size();
    }

    // ORBmatcher::SearchForTriangulation (ORBmatcher.cc:858-1024).  The FeatureVector co-iteration builds the candidate
    // list of every keypoint of KF1 (the members of the same vocabulary node in KF2, in member order), the device
    // runs the loop with its epipolar gates, the rotation histogram and the pair list are finished here.
    // hasMP / stereo: "pKF->GetMapPoint(idx) exists" and "mvuRight[idx] >= 0", one byte per keypoint.
    int SearchForTriangulation(const FrameView &KF1, const FeatureVector &fv1, const std::vector<uint8_t> &hasMP1,
                               const std::vector<uint8_t> &stereo1, const FrameView &KF2, const FeatureVector &fv2,
                               const std::vector<uint8_t> &hasMP2, const std::vector<uint8_t> &stereo2, const float F12[9],
                               float ex, float ey, bool bOnlyStereo, const std::vector<float> &scaleFactors2,
                               const std::vector<float> &levelSigma2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs)
    {
        vMatchedPairs.clear();
        std::vector<int32_t> node_of(KF1.N, -1);       // position in fv2.nodes of keypoint i's node, if both frames have it
        for (size_t a = 0, b = 0; a < fv1.nodes.size() && b < fv2.nodes.size();) {
            if (fv1.nodes[a] == fv2.nodes[b]) {
                for (int32_t k = fv1.off[a]; k < fv1.off[a + 1]; ++k) node_of[fv1.items[k]] = (int32_t)b;
                ++a; ++b;
            } else if (fv1.nodes[a] < fv2.nodes[b]) ++a;
            else ++b;
        }
        std::vector<int32_t> off(KF1.N + 1, 0), idx;
        for (int i = 0; i < KF1.N; ++i) {
            if (node_of[i] >= 0) idx.insert(idx.end(), fv2.items.begin() + fv2.off[node_of[i]], fv2.items.begin() + fv2.off[node_of[i] + 1]);
            off[i + 1] = (int32_t)idx.size();
        }
        std::vector<int32_t> m12(KF1.N, -1), bd(KF1.N, 0);
        mStatus = orbm_match_triangulation(KF1.mvKeysUn, KF1.mDescriptors, KF1.N, KF2.mvKeysUn, KF2.mDescriptors, KF2.N, off.data(),
                                           idx.data(), hasMP1.data(), hasMP2.data(), stereo1.data(), stereo2.data(), bOnlyStereo ? 1 : 0,
                                           F12, ex, ey, scaleFactors2.data(), levelSigma2.data(), (int)scaleFactors2.size(), m12.data(),
                                           bd.data());
        if (mStatus != ORBX_OK) return 0;
        int nmatches = 0;
        for (int i = 0; i < KF1.N; ++i) nmatches += m12[i] >= 0;
        if (mbCheckOrientation) {                      // :965-976, :992-1011
            std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
            const float factor = 1.0f / HISTO_LENGTH;
            for (int i = 0; i < KF1.N; ++i) {
                if (m12[i] < 0) continue;
                float rot = KF1.mvKeysUn[i].angle - KF2.mvKeysUn[m12[i]].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(i);
            }
            int ind1, ind2, ind3;
            ComputeThreeMaxima(rotHist, ind1, ind2, ind3);
            for (int b = 0; b < HISTO_LENGTH; ++b) {
                if (b == ind1 || b == ind2 || b == ind3) continue;
                for (int i : rotHist[b]) { m12[i] = -1; --nmatches; }
            }
        }
        for (int i = 0; i < KF1.N; ++i)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return nmatches;
    }

    // ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1802-1843) on the bin sizes.
    static void ComputeThreeMaxima(const std::vector<std::vector<int>> &histo, int &ind1, int &ind2, int &ind3)
    {
        int max1 = 0, max2 = 0, max3 = 0;
        ind1 = ind2 = ind3 = -1;
        for (int i = 0; i < (int)histo.size(); ++i) {
            const int s = (int)histo[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) ind3 = -1;
    }

    // The fork's SearchByProjection(Frame&, Map*, Rcw, tcw, ...) over every map point (ORBmatcher.cc:134-222): isInFrustum,
    // level prediction, windowed search and acceptance on the device.  mp*: one entry per map point (position and normal
    // xyz, distance invariance limits, MapPoint::GetDescriptor()); vMatchedMPs[j] = map point index for keypoint j or -1.
    int SearchByProjection(const FrameView &F, const std::vector<uint8_t> &hasMapPoint, const float *mpPos, const float *mpNormal,
                           const float *mpMinDistance, const float *mpMaxDistance, const uint8_t *mpDescriptors, int nMapPoints,
                           const double Rcw[9], const double tcw[3], const orbm_camera &cam, const std::vector<float> &scaleFactors,
                           float th, std::vector<int32_t> &vMatchedMPs)
    {
        vMatchedMPs.assign(F.N, -1);
        int nm = 0;
        mStatus = orbm_search_by_projection_map(F.mvKeysUn, F.mDescriptors, F.N, hasMapPoint.data(), mpPos, mpNormal, mpMinDistance,
                                                mpMaxDistance, mpDescriptors, nMapPoints, Rcw, tcw, &cam, scaleFactors.data(),
                                                (int)scaleFactors.size(), th, mfNNratio, TH_RELOC, vMatchedMPs.data(), &nm, nullptr);
        return mStatus == ORBX_OK ? nm : 0;
    }

    // MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:305-370) for many map points at once: the observed descriptors of
    // point i are rows off[i] .. off[i+1) of desc; best[i] = the row (within the point) with the least median distance.
    static int ComputeDistinctiveDescriptors(const uint8_t *desc, const std::vector<int32_t> &off, std::vector<int32_t> &best)
    {
        best.assign(off.empty() ? 0 : off.size() - 1, -1);
        return best.empty() ? ORBX_OK : orbm_distinctive_descriptors(desc, off.data(), (int)best.size(), best.data());
    }

    int status() const { return mStatus; }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    int mStatus = ORBX_OK;
    std::vector<int32_t> mScratch;     // per-entry output of the whole-function searches (not re-entrant per OBJECT; instances are stack-local, as in the reference)
};

// ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ORBVocabulary.h:31) as the
// front end uses it: loadFromTextFile (System.cc:69) and transform (Frame.cc:410-417).  The tree lives in HBM
// behind orbm_vocab_create; the descent of all features of a frame is one device call, the two std::map
// results are filled here in feature order like TemplatedVocabulary.h:1127-1160.
class ORBVocabulary {
    typedef DBoW2::BowVector BowVector;
    typedef DBoW2::FeatureVector FeatureVector;


public:
    ORBVocabulary() {}
    ~ORBVocabulary() { clear(); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // Text layout of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1420): header `k L scoring
    // weighting`, then per node `parent isLeaf d0..d31 weight`; node ids are line numbers, word ids count the leaves.
    bool loadFromTextFile(const std::string &filename) {
        clear();
        FILE *f = std::fopen(filename.c_str(), "r");
        if (!f) return false;
        int k = 0, L = 0, n1 = 0, n2 = 0;
        bool ok = std::fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) == 4 && !(k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3);
        std::vector<int32_t> parent(1, 0), word(1, -1);
        std::vector<uint8_t> desc(32, 0);
        std::vector<double> weight(1, 0.0);
        int nwords = 0;
        while (ok) {
            int pid, leaf;
            if (std::fscanf(f, "%d %d", &pid, &leaf) != 2) break;            // end of file
            const int id = (int)parent.size();
            if (pid < 0 || pid >= id) { ok = false; break; }
            unsigned b[32];
            for (int i = 0; i < 32 && ok; ++i) ok = std::fscanf(f, "%u", &b[i]) == 1 && b[i] < 256;
            double w = 0;
            ok = ok && std::fscanf(f, "%lf", &w) == 1;
            if (!ok) break;
            parent.push_back(pid); word.push_back(leaf > 0 ? nwords++ : -1); weight.push_back(w);
            for (int i = 0; i < 32; ++i) desc.push_back((uint8_t)b[i]);
        }
        std::fclose(f);
        if (!ok) return false;
        const int n = (int)parent.size();
        std::vector<int32_t> off(n + 1, 0), ids(n - 1), fill(n, 0);
        for (int i = 1; i < n; ++i) ++off[parent[i] + 1];
        for (int i = 0; i < n; ++i) off[i + 1] += off[i];
        for (int i = 1; i < n; ++i) ids[off[parent[i]] + fill[parent[i]]++] = i;   // children in order of appearance
        m_k = k; m_L = L; m_scoring = n1; m_weighting = n2; m_nwords = nwords; m_nnodes = n;
        mStatus = orbm_vocab_create(off.data(), ids.data(), desc.data(), word.data(), weight.data(), n, L, &mVoc);
        return mStatus == ORBX_OK;
    }

    bool empty() const { return m_nwords == 0; }
    unsigned int size() const { return (unsigned int)m_nwords; }
    int getBranchingFactor() const { return m_k; }
    int getDepthLevels() const { return m_L; }

    // features: n x 32 bytes (Converter::toDescriptorVector of mDescriptors is this, row by row).
    void transform(const uint8_t *features, int n, BowVector &v, FeatureVector &fv, int levelsup) {
        v.clear(); fv.clear();
        if (!mVoc || n <= 0) return;
        mWord.resize(n); mNode.resize(n); mW.resize(n);
        mStatus = orbm_bow_transform(mVoc, features, n, levelsup, mWord.data(), mNode.data(), mW.data());
        if (mStatus != ORBX_OK) return;
        const bool tf = m_weighting == 0 || m_weighting == 1;                // TF_IDF / TF accumulate, IDF / BINARY keep the first
        for (int i = 0; i < n; ++i)
            if (mW[i] > 0) {                                                 // not stopped
                if (tf) v[(unsigned)mWord[i]] += mW[i];                      // BowVector::addWeight
                else v.insert(BowVector::value_type((unsigned)mWord[i], mW[i]));   // BowVector::addIfNotExist
                fv[(unsigned)mNode[i]].push_back((unsigned)i);               // FeatureVector::addFeature
            }
        const bool must = m_scoring != 5;                                    // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
        if (tf && !v.empty() && !must) {
            const double nd = (double)v.size();
            for (auto &e : v) e.second /= nd;
        }
        if (must) {                                                          // BowVector::normalize: L2 for L2_NORM, else L1
            double norm = 0;
            if (m_scoring == 1) { for (auto &e : v) norm += e.second * e.second; norm = std::sqrt(norm); }
            else for (auto &e : v) norm += std::fabs(e.second);
            if (norm > 0) for (auto &e : v) e.second /= norm;
        }
    }

    orbm_vocabulary *handle() const { return mVoc; }
    int status() const { return mStatus; }

private:
    void clear() { if (mVoc) orbm_vocab_destroy(mVoc); mVoc = nullptr; m_nwords = m_nnodes = 0; }
    orbm_vocabulary *mVoc = nullptr;
    int m_k = 0, m_L = 0, m_scoring = 0, m_weighting = 0, m_nwords = 0, m_nnodes = 0, mStatus = ORBX_OK;
    std::vector<int32_t> mWord, mNode;
    std::vector<double> mW;
};

// Numeric core of FEA2: the members and methods g2o and Optimizer touch
// (FEA2.h: K-size, vva, vvf, sE, nsE; Set/Compute* calls at
// optimization_algorithm_levenberg.cpp:159-199).
class FEA2 {
public:
    FEA2(unsigned int input_E, float input_nu, float input_h, float input_fg1, int input_nElType)
        : E(input_E), nu(input_nu), h(input_h), fg(input_fg1), nElType(input_nElType) {}
    ~FEA2() { fem_destroy(mModel); }
    FEA2(const FEA2 &) = delete;
    FEA2 &operator=(const FEA2 &) = delete;

    // SetSecondLayer + Set_u0 + MatrixAssemblyC3D8/6 + ImposeDirichletEncastre_K
    // (the numeric half of FEA2::Compute(1), FEA2.cc:92-103) for a top-layer mesh:
    // faces = triangles (nElType 2) or quads (nElType 1).
    bool Compute(const std::vector<float> &topXYZ, const std::vector<int32_t> &faces)
    {
        const int nTop = (int)topXYZ.size() / 3, nv = nElType == 1 ? 4 : 3, nf = (int)faces.size() / nv;
        std::vector<float> nodes((size_t)6 * nTop);
        fem_second_layer(topXYZ.data(), nTop, h, nodes.data());
        u0 = nodes;
        std::vector<int32_t> elems((size_t)nf * 2 * nv);
        for (int f = 0; f < nf; ++f)
            for (int k = 0; k < nv; ++k) {
                elems[(size_t)f * 2 * nv + k] = faces[(size_t)f * nv + k];
                elems[(size_t)f * 2 * nv + nv + k] = faces[(size_t)f * nv + k] + nTop; // FEA2.cc:1392-1399
            }
        fem_destroy(mModel);
        mModel = nullptr;
        mTrialReady = false;
        mStatus = fem_create(nElType == 1 ? FEM_C3D8 : FEM_C3D6, nodes.data(), 1, 2 * nTop, elems.data(), nf, E, nu, fg, &mModel);
        if (mStatus != ORBX_OK) return false; // Ksize<=3: assembly refuses (:1386)
        Ksize = 6 * nTop;
        vDir.resize(nTop);
        for (int i = 0; i < nTop; ++i) vDir[i] = nTop + i; // vvDir_t, :1198
        if ((mStatus = fem_assemble(mModel)) != ORBX_OK) return false;
        mStatus = fem_dirichlet_penalty(mModel, vDir.data(), nTop, 100000000.0f);
        return mStatus == ORBX_OK;
    }

    // Set_uf (top layer moved, bottom layer kept) + ComputeDisplacement (:1732-1808)
    void ComputeDisplacement(const std::vector<float> &topXYZ_new)
    {
        std::vector<float> uf = u0;
        std::memcpy(uf.data(), topXYZ_new.data(), sizeof(float) * topXYZ_new.size());
        vva.resize(Ksize);
        mStatus = fem_displacement(mModel, uf.data(), u0.data(), vDir.data(), (int)vDir.size(), 100000000.0f, vva.data());
    }
    void ComputeForces() { vvf.resize(Ksize); mStatus = fem_matvec(mModel, vva.data(), vvf.data()); }
    float ComputeStrainEnergy() { mStatus = fem_strain_energy(mModel, vva.data(), &sE, &nsE); return sE; }
    float NormalizeStrainEnergy() const { return nsE; }

    // The whole per-trial sequence of the LM hook (optimization_algorithm_levenberg.cpp:159-175)
    // in one call, K / u0 / Dirichlet list resident: vertex estimates in, nsE out.
    float TrialEnergy(const std::vector<double> &vertexXYZ)
    {
        if (!mTrialReady) {
            mStatus = fem_trial_setup(mModel, u0.data(), vDir.data(), (int)vDir.size(), 100000000.0f, (int)vertexXYZ.size() / 3, nullptr, 0);
            mTrialReady = mStatus == ORBX_OK;
        }
        if (mTrialReady) mStatus = fem_trial_energy(mModel, vertexXYZ.data(), nullptr, &sE, &nsE);
        return nsE;
    }

    unsigned int E;
    float nu, h, fg;
    int nElType, Ksize = 0;
    float sE = 0.f, nsE = 0.f;
    std::vector<float> u0, vva, vvf;
    std::vector<int32_t> vDir;
    int status() const { return mStatus; }
    fem_model *model() { return mModel; }

private:
    fem_model *mModel = nullptr;
    int mStatus = ORBX_OK;
    bool mTrialReady = false;
};

// The FEM side of Optimizer::PoseOptimizationNR (src/Optimizer.cc:478-834) as one compiled sequence:
//   FEA2 fea2(mnId, 3500, 0.495, 0.5, 0.577350269, nElType)                       :480
//   fea2.Compute(1)                                                               :723   -> Compute()
//   4 x { optimizer.initializeOptimization(0); optimizer.optimize(10); classify } :733-790 -> Optimize()
// and, inside every Levenberg iteration of optimize(), g2o's trial loop with this fork's hook
// (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:63-232; the hook :159-199) -> Iteration() / Hook().
// What g2o computes -- the reprojection chi2, the linear solve, the vertex estimates, push / pop -- stays g2o's and
// comes in through `Problem` (a g2o::SparseOptimizer + BlockSolver in the real build, a scripted stand-in in the tests):
//   void   initializeOptimization(int round)          optimizer.initializeOptimization(0)              Optimizer.cc:738
//   double activeRobustChi2()                         computeActiveErrors() + activeRobustChi2()       levenberg.cpp:80-99
//   void   buildSystem()                              _solver->buildSystem()                           :102
//   double computeLambdaInit()                        :242-256
//   void   push(), pop(), discardTop()                _optimizer->push() / pop() / discardTop()        :123, 222, 216
//   bool   solveAndUpdate(double lambda)              setLambda, solve, update(x), restoreDiagonal     :130-146 (-> ok2)
//   double computeScale(double lambda)                :258-267
//   void   pointEstimates(std::vector<double> &xyz)   GetPointCoordinates(pFEA2->vVertices)            :293-311
//   bool   terminate()                                _optimizer->terminate()
//   void   classifyOutliers(int round)                the inlier / outlier pass after each round       Optimizer.cc:752-790
class PoseOptimizationNR_fem {
public:
    struct Trial { float sE, nsE; double tempChi, currentChi, rho, lambda; int qmax, accepted; };
    enum SolverResult { Terminate = 2, OK = 1, Fail = -1 };   // OptimizationAlgorithm::SolverResult

    explicit PoseOptimizationNR_fem(int nElType) : fea2(3500, 0.495f, 0.5f, 0.577350269f, nElType) {}

    // fea2.Compute(1), Optimizer.cc:723 (its numeric half: the PCL meshing in front stays the reference's), and the
    // state of the hook parked on the device: u0, the Dirichlet list, the optimiser's nVertices point vertices and
    // vNewPointsBase (derived[nDerived][4] = {2 | 3, i0, i1, i2}; nVertices + nDerived = top-layer nodes).
    bool Compute(const std::vector<float> &topXYZ, const std::vector<int32_t> &faces, int nVertices,
                 const std::vector<int32_t> &derived = std::vector<int32_t>())
    {
        if (!fea2.Compute(topXYZ, faces)) return false;
        mStatus = fem_trial_setup(fea2.model(), fea2.u0.data(), fea2.vDir.data(), (int)fea2.vDir.size(), 100000000.0f, nVertices,
                                  derived.empty() ? nullptr : derived.data(), (int)derived.size() / 4);
        mNV = nVertices;
        return mStatus == ORBX_OK;
    }

    // The hook, levenberg.cpp:159-199: GetPointCoordinates + Set_uf + ComputeDisplacement + ComputeForces +
    // ComputeStrainEnergy + NormalizeStrainEnergy in one device call, then g2o's weighting, verbatim.
    double Hook(const std::vector<double> &vertexXYZ, int qmax, double tempChi, double &currentChi, float &sE, float &nsE)
    {
        sE = 0.0f; nsE = 0.0f;
        mStatus = fem_trial_energy(fea2.model(), vertexXYZ.data(), nullptr, &sE, &nsE);     // :164-171
        float w_rE = 1.0;                                                                    // :184
        float w_sE = 5.0;                                                                    // :185
        if (qmax == 0) {                                                                     // :186-193
            w_rE = 1.0;
            w_sE = 2.0;
            currentChi += nsE;
        }
        return w_rE * tempChi + w_sE * nsE;                                                  // :198
    }

    // One Levenberg iteration = OptimizationAlgorithmLevenberg::solve(iteration), levenberg.cpp:63-232, bInFEA set.
    template <class Problem>
    SolverResult Iteration(Problem &g2o, int iteration, std::vector<Trial> *log = nullptr)
    {
        double currentChi = g2o.activeRobustChi2();            // :80-97
        double tempChi = currentChi;
        const double iniChi = currentChi;
        g2o.buildSystem();                                     // :102
        if (iteration == 0) {                                  // :109-114
            mLambda = g2o.computeLambdaInit();
            mNi = 2;
            mNBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        std::vector<double> pts;
        do {
            g2o.push();                                        // :123
            const bool ok2 = g2o.solveAndUpdate(mLambda);      // :130-146
            tempChi = g2o.activeRobustChi2();                  // :148-157
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            float sE, nsE;
            g2o.pointEstimates(pts);
            if ((int)pts.size() != 3 * mNV) { mStatus = ORBX_ERR_ARG; return Fail; }
            tempChi = Hook(pts, qmax, tempChi, currentChi, sE, nsE);   // :159-199
            if (mStatus != ORBX_OK) return Fail;
            rho = (currentChi - tempChi);                      // :201
            double scale = g2o.computeScale(mLambda);
            scale += 1e-3;
            rho /= scale;
            const bool good = rho > 0 && std::isfinite(tempChi);
            if (good) {                                        // :207-217
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = (std::min)(alpha, mGoodStepUpperScale);
                const double scaleFactor = (std::max)(mGoodStepLowerScale, alpha);
                mLambda *= scaleFactor;
                mNi = 2;
                currentChi = tempChi;
                g2o.discardTop();
            } else {                                           // :218-223
                mLambda *= mNi;
                mNi *= 2;
                g2o.pop();
            }
            if (log) log->push_back(Trial{sE, nsE, tempChi, currentChi, rho, mLambda, qmax, good ? 1 : 0});
            qmax++;
        } while (rho < 0 && qmax < mMaxTrialsAfterFailure && !g2o.terminate());   // :226
        if (qmax == mMaxTrialsAfterFailure || rho == 0) return Terminate;           // :228-229
        if ((iniChi - currentChi) * 1e3 < iniChi) mNBad++;                           // :232-235 (the stop criterion added in ORB-SLAM2's g2o)
        else mNBad = 0;
        if (mNBad >= 3) return Terminate;
        return OK;
    }

    // The four rounds of Optimizer.cc:733-790: its[] = {10, 10, 10, 10}; SparseOptimizer::optimize's loop stops at the
    // first result that is not OK.  Returns the number of Levenberg iterations run.
    template <class Problem>
    int Optimize(Problem &g2o, std::vector<Trial> *log = nullptr, std::vector<int> *results = nullptr)
    {
        static const int its[4] = {10, 10, 10, 10};            // :730
        int total = 0;
        for (int round = 0; round < 4; ++round) {
            g2o.initializeOptimization(round);                 // :738
            for (int i = 0; i < its[round] && !g2o.terminate(); ++i) {
                const SolverResult r = Iteration(g2o, i, log);
                ++total;
                if (results) results->push_back((int)r);
                if (r != OK) break;
            }
            g2o.classifyOutliers(round);                       // :752-790
        }
        return total;
    }

    FEA2 fea2;
    int status() const { return mStatus == ORBX_OK ? fea2.status() : mStatus; }
    double lambda() const { return mLambda; }

private:
    int mStatus = ORBX_OK, mNV = 0;
    double mLambda = -1., mNi = 2., mGoodStepLowerScale = 1. / 3., mGoodStepUpperScale = 2. / 3.;   // levenberg.cpp:48-56
    int mNBad = 0, mMaxTrialsAfterFailure = 10;                                                     // :50
};

} // namespace orbslam_hip
#endif // ORBSLAM_HIP_HPP
