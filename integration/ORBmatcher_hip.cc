// ORBmatcher_hip.cc -- ORB_SLAM2::ORBmatcher over liborbslam_hip.so.  Replaces src/ORBmatcher.cc in the reference build; the class
// declaration (include/ORBmatcher.h) is unchanged.  Every Search* / Fuse method keeps its walk over the pointer graph (MapPoint*,
// KeyFrame*: tests a device cannot make) and hands the rest -- projection, GetFeaturesInArea, descriptor distances, best / second
// selection, the coupling between list entries, acceptance, rotation histogram -- to ONE C-ABI call on the frame's resident handle
// (Frame::mpHipFrame / KeyFrame::mpHipFrame, integration/hip_frame.h).  Results equal the reference's sequential loops.
#include "ORBmatcher.h"

#include <limits.h>
#include <stdint.h>
#include <string.h>

#include <opencv2/core/core.hpp>

#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "hip_frame.h"
#include "orbslam_hip.h"

using namespace std;

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 95;      // src/ORBmatcher.cc:37-40
const int ORBmatcher::TH_LOW = 45;
const int ORBmatcher::TH_RELOC = 60;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// ---- flattening helpers ------------------------------------------------------------------------------------------------------

void HipPointList::set(size_t i, MapPoint *pMP)
{
    valid[i] = 1;
    const cv::Mat P = pMP->GetWorldPos(), Pn = pMP->GetNormal(), d = pMP->GetDescriptor();
    for (int k = 0; k < 3; ++k) { pos[3 * i + k] = P.at<float>(k); normal[3 * i + k] = Pn.at<float>(k); }
    memcpy(&desc[32 * i], d.ptr<uint8_t>(), 32);
    minDistance[i] = pMP->mfMinDistance;          // (friend access, reference.patch: the 0.8 / 1.2 of Get{Min,Max}DistanceInvariance
    maxDistance[i] = pMP->mfMaxDistance;          //  are applied by the library)
    takes[i] = pMP->Observations() > 0;
}

static orbm_view ViewOf(const Frame &F)
{
    return orbm_view{Frame::fx, Frame::fy, Frame::cx, Frame::cy, F.mb, F.mbf, F.mfLogScaleFactor, F.mnScaleLevels, F.mvScaleFactors.data()};
}
static orbm_view ViewOf(const KeyFrame *pKF)
{
    return orbm_view{pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mb, pKF->mbf, pKF->mfLogScaleFactor, pKF->mnScaleLevels, pKF->mvScaleFactors.data()};
}

// DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) in map order
struct FlatFeatVec {
    vector<int32_t> nodes, off, items;
    explicit FlatFeatVec(const DBoW2::FeatureVector &fv)
    {
        off.push_back(0);
        for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            nodes.push_back((int32_t)it->first);
            items.insert(items.end(), it->second.begin(), it->second.end());
            off.push_back((int32_t)items.size());
        }
    }
};

// ---- single pair (src/ORBmatcher.cc:1848-1864).  Kept for the odd caller; loops over pairs belong in the batched entries
// (orbm_hamming_matrix, orbm_distinctive_descriptors for MapPoint::ComputeDistinctiveDescriptors).
int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    uint16_t d = 0;
    orbm_hamming_matrix(a.ptr<uint8_t>(), 1, b.ptr<uint8_t>(), 1, &d);
    return d;
}

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 3.0 : 4.5; }      // :332-338

// ---- SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)   :46-132
// Tracking::SearchLocalPoints has run Frame::isInFrustum on the points (mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel,
// mTrackViewCos): the windows are formed from those fields, as the reference does at :62-70.
int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
{
    const bool bFactor = th != 1.0;
    const size_t n = vpMapPoints.size();
    vector<orbm_window_query> q(n);
    vector<uint8_t> qd(32 * n, 0), takes(n, 1), occ(F.N, 0);
    for (size_t iMP = 0; iMP < n; ++iMP) {
        MapPoint *pMP = vpMapPoints[iMP];
        q[iMP] = orbm_window_query{0.f, 0.f, -1.f, 0.f, 0, -1};                  // r < 0: skipped entry
        if (!pMP->mbTrackInView) continue;                                      // :57-58
        if (pMP->isBad()) continue;                                             // :60-61
        const int &nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);                       // :66
        if (bFactor) r *= th;
        q[iMP] = orbm_window_query{pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], pMP->mTrackProjXR,
                                   nPredictedLevel - 1, nPredictedLevel};
        memcpy(&qd[32 * iMP], pMP->GetDescriptor().ptr<uint8_t>(), 32);
        takes[iMP] = pMP->Observations() > 0;
    }
    for (int j = 0; j < F.N; ++j) occ[j] = F.mvpMapPoints[j] && F.mvpMapPoints[j]->Observations() > 0;   // :87-89
    vector<int32_t> slot(F.N > 0 ? F.N : 1), chosen(n ? n : 1);
    int nmatches = 0;
    if (orbm_frame_search_projection(F.mpHipFrame.get(), q.data(), qd.data(), nullptr, takes.data(), (int)n, occ.data(), TH_HIGH, mfNNratio,
                                     /*ratio_same_level=*/1, /*check_orientation=*/0, slot.data(), chosen.data(), &nmatches) != ORBX_OK)
        return 0;
    for (int j = 0; j < F.N; ++j)
        if (slot[j] >= 0) F.mvpMapPoints[j] = vpMapPoints[slot[j]];             // :126
    return nmatches;
}

// ---- the fork's whole-map form   :134-222 (isInFrustum :262-330, ComputeDistance :224-260 run on the device)
int ORBmatcher::SearchByProjection(Frame &F, Map *pMap, double mCamRcw[3][3], double mCamtcw[3], vector<MapPoint*> &vMatchedMPs,
                                   vector<cv::KeyPoint> &vMatchedKPs, vector<bool> &vbMatched, const float th)
{
    vector<MapPoint*> vpAllMapPoints = pMap->GetAllMapPoints();
    const int nFeatures = F.mvKeysUn.size(), m = (int)vpAllMapPoints.size();
    vMatchedMPs = vector<MapPoint*>(nFeatures, static_cast<MapPoint*>(NULL));
    vMatchedKPs.resize(nFeatures);              // (the reference writes into a merely reserved vector, :142,214)
    vbMatched = vector<bool>(nFeatures, false);
    HipPointList pts(m);
    for (int i = 0; i < m; ++i) pts.set(i, vpAllMapPoints[i]);
    vector<uint8_t> has(nFeatures, 0);
    for (int j = 0; j < nFeatures; ++j) has[j] = F.mvpMapPoints[j] != NULL;       // :186-187
    const vector<int> b = F.GetImageBounds();
    const orbm_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, b[0], b[1], b[2], b[3], Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    vector<int32_t> matched(nFeatures > 0 ? nFeatures : 1);
    vector<float> proj(4 * (size_t)(m ? m : 1));
    int nmatches = 0;
    if (orbm_frame_search_by_projection_map(F.mpHipFrame.get(), has.data(), pts.pos.data(), pts.normal.data(), pts.minDistance.data(),
                                            pts.maxDistance.data(), pts.desc.data(), m, &mCamRcw[0][0], mCamtcw, &cam, F.mvScaleFactors.data(),
                                            F.mnScaleLevels, th, mfNNratio, TH_RELOC, matched.data(), &nmatches, proj.data()) != ORBX_OK)
        return 0;
    for (int i = 0; i < m; ++i) {                // what isInFrustum leaves in the MapPoint (:263,325-328)
        MapPoint *pMP = vpAllMapPoints[i];
        if (proj[4 * i + 3] < 0) continue;                   // not in the frustum: the reference returns before writing
        pMP->mRelocProjX = proj[4 * i]; pMP->mRelocProjY = proj[4 * i + 1]; pMP->mRelocViewCos = proj[4 * i + 2];
        pMP->mnRelocScaleLevel = (int)proj[4 * i + 3];
    }
    for (int j = 0; j < nFeatures; ++j)
        if (matched[j] >= 0) { vMatchedMPs[j] = vpAllMapPoints[matched[j]]; vMatchedKPs[j] = F.mvKeys[j]; vbMatched[j] = true; }   // :212-216
    return nmatches;
}

// ---- SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)   :1529-1671
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    HipPointList last(LastFrame.N);
    for (int i = 0; i < LastFrame.N; ++i) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;                          // :1555-1558
        last.set(i, pMP);
        last.octave[i] = LastFrame.mvKeys[i].octave;                            // :1578
        last.angle[i] = LastFrame.mvKeysUn[i].angle;                            // :1642
    }
    vector<uint8_t> occ(CurrentFrame.N, 0);
    for (int j = 0; j < CurrentFrame.N; ++j)
        occ[j] = CurrentFrame.mvpMapPoints[j] && CurrentFrame.mvpMapPoints[j]->Observations() > 0;    // :1603-1605
    float Tcw[16], Tlw[16];
    HipPose(CurrentFrame.mTcw, Tcw); HipPose(LastFrame.mTcw, Tlw);
    const orbm_view view = ViewOf(CurrentFrame);
    const orbm_points pl = last.view();
    vector<int32_t> slot(CurrentFrame.N > 0 ? CurrentFrame.N : 1), chosen(LastFrame.N > 0 ? LastFrame.N : 1);
    int nmatches = 0;
    if (orbm_search_by_projection_last(CurrentFrame.mpHipFrame.get(), &view, Tcw, Tlw, &pl, occ.data(), th, bMono, TH_HIGH, mbCheckOrientation,
                                       slot.data(), chosen.data(), &nmatches, nullptr) != ORBX_OK)
        return 0;
    for (int j = 0; j < CurrentFrame.N; ++j) {
        if (slot[j] >= 0) CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[slot[j]];             // :1634
        else if (slot[j] == -2) CurrentFrame.mvpMapPoints[j] = static_cast<MapPoint*>(NULL);          // :1664
    }
    return nmatches;
}

// ---- SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)   :1673-1800
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
{
    const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    HipPointList kf(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); ++i) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;         // :1695-1700
        kf.set(i, pMP);
        kf.angle[i] = pKF->mvKeysUn[i].angle;                                   // :1771
    }
    vector<uint8_t> occ(CurrentFrame.N, 0);
    for (int j = 0; j < CurrentFrame.N; ++j) occ[j] = CurrentFrame.mvpMapPoints[j] != NULL;           // :1741-1742
    float Tcw[16];
    HipPose(CurrentFrame.mTcw, Tcw);
    const orbm_view view = ViewOf(CurrentFrame);
    const orbm_points pk = kf.view();
    vector<int32_t> slot(CurrentFrame.N > 0 ? CurrentFrame.N : 1), chosen(vpMPs.size() ? vpMPs.size() : 1);
    int nmatches = 0;
    if (orbm_search_by_projection_keyframe(CurrentFrame.mpHipFrame.get(), &view, Tcw, &pk, occ.data(), th, ORBdist, mbCheckOrientation,
                                           slot.data(), chosen.data(), &nmatches, nullptr) != ORBX_OK)
        return 0;
    for (int j = 0; j < CurrentFrame.N; ++j) {
        if (slot[j] >= 0) CurrentFrame.mvpMapPoints[j] = vpMPs[slot[j]];                              // :1763
        else if (slot[j] == -2) CurrentFrame.mvpMapPoints[j] = NULL;                                  // :1793
    }
    return nmatches;
}

// ---- SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)   :491-604
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
{
    set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());          // :507-508
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    HipPointList pts(vpPoints.size());
    for (size_t iMP = 0; iMP < vpPoints.size(); ++iMP) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;                // :516-517
        pts.set(iMP, pMP);
    }
    vector<uint8_t> occ(pKF->N, 0);
    for (int j = 0; j < pKF->N; ++j) occ[j] = vpMatched[j] != NULL;             // :574-575
    float S[16];
    HipPose(Scw, S);
    const orbm_view view = ViewOf(pKF);
    const orbm_points pp = pts.view();
    vector<int32_t> slot(pKF->N > 0 ? pKF->N : 1), chosen(vpPoints.size() ? vpPoints.size() : 1);
    int nmatches = 0;
    if (orbm_search_by_projection_sim3(pKF->mpHipFrame.get(), &view, S, &pp, occ.data(), th, TH_LOW, slot.data(), chosen.data(), &nmatches,
                                       nullptr) != ORBX_OK)
        return 0;
    for (int j = 0; j < pKF->N; ++j)
        if (slot[j] >= 0) vpMatched[j] = vpPoints[slot[j]];                     // :597
    return nmatches;
}

// ---- SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)   :360-489
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)
{
    const vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    const FlatFeatVec f1(pKF->mFeatVec), f2(F.mFeatVec);
    vector<uint8_t> valid1(pKF->N, 0);
    for (int i = 0; i < pKF->N; ++i) valid1[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();       // :395-399
    vector<int32_t> match12(pKF->N > 0 ? pKF->N : 1);
    int nmatches = 0;                 // both frames are resident: the feature vectors and the mask are all that travels
    if (orbm_frame_search_by_bow(pKF->mpHipFrame.get(), f1.nodes.data(), f1.off.data(), f1.items.data(), (int)f1.nodes.size(), valid1.data(),
                                 F.mpHipFrame.get(), f2.nodes.data(), f2.off.data(), f2.items.data(), (int)f2.nodes.size(), nullptr,
                                 TH_LOW, /*strict_th=*/0, mfNNratio, mbCheckOrientation, match12.data(), nullptr, &nmatches) != ORBX_OK)
        return 0;
    for (int i = 0; i < pKF->N; ++i)
        if (match12[i] >= 0) vpMapPointMatches[match12[i]] = vpMapPointsKF[i];                          // :433
    return nmatches;
}

// ---- SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)   :723-856
int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)
{
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    vector<uint8_t> valid1(pKF1->N, 0), valid2(pKF2->N, 0);
    for (int i = 0; i < pKF1->N; ++i) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();         // :763-767
    for (int i = 0; i < pKF2->N; ++i) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();         // :782-786
    vector<int32_t> match12(pKF1->N > 0 ? pKF1->N : 1);
    int nmatches = 0;
    if (orbm_frame_search_by_bow(pKF1->mpHipFrame.get(), f1.nodes.data(), f1.off.data(), f1.items.data(), (int)f1.nodes.size(), valid1.data(),
                                 pKF2->mpHipFrame.get(), f2.nodes.data(), f2.off.data(), f2.items.data(), (int)f2.nodes.size(), valid2.data(),
                                 TH_LOW, /*strict_th=*/1, mfNNratio, mbCheckOrientation, match12.data(), nullptr, &nmatches) != ORBX_OK)
        return 0;
    for (int i = 0; i < pKF1->N; ++i)
        if (match12[i] >= 0) vpMatches12[i] = vpMapPoints2[match12[i]];                                 // :803
    return nmatches;
}

// ---- SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)   :606-721
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
{
    vnMatches12 = vector<int>(F1.mvKeysUn.size(), -1);
    int nmatches = 0;                 // cv::KeyPoint has orbx_keypoint's layout, cv::Point2f[] is float[][2]
    if (orbm_frame_search_for_initialization(F2.mpHipFrame.get(), reinterpret_cast<const orbx_keypoint *>(F2.mvKeysUn.data()),
                                             reinterpret_cast<const orbx_keypoint *>(F1.mvKeysUn.data()), F1.mDescriptors.data, F1.N,
                                             reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio, mbCheckOrientation,
                                             vnMatches12.data(), &nmatches) != ORBX_OK)
        return 0;
    return nmatches;
}

// ---- SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t>> &vMatchedPairs, bOnlyStereo)   :858-1024
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
{
    // epipole in the second image (:863-871)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    const int n1 = pKF1->N, n2 = pKF2->N;
    vector<uint8_t> has1(n1, 0), has2(n2, 0);      // "stereo" (mvuRight >= 0, :911 / :931) is read from the resident frames
    for (int i = 0; i < n1; ++i) has1[i] = pKF1->GetMapPoint(i) != NULL;                                           // :900-907
    for (int i = 0; i < n2; ++i) has2[i] = pKF2->GetMapPoint(i) != NULL;                                           // :921-927
    float F[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
    vector<int32_t> match12(n1 > 0 ? n1 : 1);
    int nmatches = 0;
    if (orbm_frame_search_for_triangulation(pKF1->mpHipFrame.get(), f1.nodes.data(), f1.off.data(), f1.items.data(), (int)f1.nodes.size(),
                                            has1.data(), pKF2->mpHipFrame.get(), f2.nodes.data(), f2.off.data(), f2.items.data(),
                                            (int)f2.nodes.size(), has2.data(), bOnlyStereo, F, ex, ey, pKF2->mvScaleFactors.data(),
                                            pKF2->mvLevelSigma2.data(), pKF2->mnScaleLevels, mbCheckOrientation, match12.data(),
                                            &nmatches) != ORBX_OK)
        return 0;
    vMatchedPairs.clear();
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; ++i)
        if (match12[i] >= 0) vMatchedPairs.push_back(make_pair((size_t)i, (size_t)match12[i]));          // :1014-1021
    return nmatches;
}

// ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   :1303-1527
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th)
{
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = vpMapPoints1.size(), N2 = vpMapPoints2.size();
    vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                              // :1332-1343
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    HipPointList p1(N1), p2(N2);
    for (int i = 0; i < N1; ++i) { MapPoint *pMP = vpMapPoints1[i]; if (pMP && !vbAlreadyMatched1[i] && !pMP->isBad()) p1.set(i, pMP); }   // :1352-1358
    for (int i = 0; i < N2; ++i) { MapPoint *pMP = vpMapPoints2[i]; if (pMP && !vbAlreadyMatched2[i] && !pMP->isBad()) p2.set(i, pMP); }   // :1434-1440
    float T1w[16], T2w[16], R[9], t[3];
    HipPose(pKF1->GetPose(), T1w); HipPose(pKF2->GetPose(), T2w);               // GetRotation / GetTranslation = the pose's blocks
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = R12.at<float>(r, c); t[r] = t12.at<float>(r); }
    const orbm_view view = ViewOf(pKF1);
    const orbm_points q1 = p1.view(), q2 = p2.view();
    vector<int32_t> match12(N1 > 0 ? N1 : 1);
    int nFound = 0;
    if (orbm_search_by_sim3(pKF1->mpHipFrame.get(), pKF2->mpHipFrame.get(), &view, T1w, T2w, s12, R, t, &q1, &q2, th, TH_HIGH, nullptr, nullptr,
                            match12.data(), &nFound, nullptr, nullptr) != ORBX_OK)
        return 0;
    for (int i1 = 0; i1 < N1; ++i1)
        if (match12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match12[i1]];       // :1518
    return nFound;
}

// ---- Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th)   :1026-1176
// Projection + gated candidate search for every listed point on the device (they do not depend on the map state); the loop's skips and
// its map update, which do, replayed in list order.
int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th)
{
    const int nMPs = vpMapPoints.size();
    HipPointList pts(nMPs);
    for (int i = 0; i < nMPs; ++i) if (vpMapPoints[i]) pts.set(i, vpMapPoints[i]);
    float Tcw[16], Rcw[9], tcw[3], Ow[3];
    HipPose(pKF->GetPose(), Tcw);
    const cv::Mat O = pKF->GetCameraCenter();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rcw[3 * r + c] = Tcw[4 * r + c]; tcw[r] = Tcw[4 * r + 3]; Ow[r] = O.at<float>(r); }
    const orbm_camera cam = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0, 0, 0, 0, (float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY};
    vector<orbm_projected_point> proj(nMPs ? nMPs : 1);
    vector<orbm_window_query> q(nMPs ? nMPs : 1);
    vector<int32_t> best(nMPs ? nMPs : 1), idx(nMPs ? nMPs : 1);
    if (orbm_project_points(ORBM_PROJECT_FUSE, pts.pos.data(), pts.normal.data(), pts.minDistance.data(), pts.maxDistance.data(), nMPs, Rcw, tcw, Ow,
                            &cam, pKF->mbf, 0.f, pKF->mfLogScaleFactor, pKF->mvScaleFactors.data(), pKF->mnScaleLevels, th, proj.data(), q.data()) != ORBX_OK ||
        orbm_frame_search_fuse(pKF->mpHipFrame.get(), q.data(), pts.desc.data(), nMPs, pKF->mvInvLevelSigma2.data(), pKF->mnScaleLevels, best.data(),
                               idx.data()) != ORBX_OK)
        return 0;
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;                                                     // :1043-1044
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                   // :1046-1047
        if (!proj[i].visible || idx[i] < 0) continue;                           // projection tests, empty window, no candidate passed the gate
        if (best[i] <= TH_LOW) {                                                // :1149
            MapPoint *pMPinKF = pKF->GetMapPoint(idx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, idx[i]);
                pKF->AddMapPoint(pMP, idx[i]);
            }
            nFused++;
        }
    }
    return nFused;
}

// ---- Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint)   :1178-1301
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint)
{
    // :1186-1192
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();                  // :1194: a snapshot
    const int nPoints = vpPoints.size();
    HipPointList pts(nPoints);
    for (int i = 0; i < nPoints; ++i) pts.set(i, vpPoints[i]);
    float R[9], t[3], O[3];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = Rcw.at<float>(r, c); t[r] = tcw.at<float>(r); O[r] = Ow.at<float>(r); }
    const orbm_camera cam = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0, 0, 0, 0, (float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY};
    vector<orbm_projected_point> proj(nPoints ? nPoints : 1);
    vector<orbm_window_query> q(nPoints ? nPoints : 1);
    vector<int32_t> best(nPoints ? nPoints : 1), idx(nPoints ? nPoints : 1);
    if (orbm_project_points(ORBM_PROJECT_FUSE_SIM3, pts.pos.data(), pts.normal.data(), pts.minDistance.data(), pts.maxDistance.data(), nPoints, R, t, O,
                            &cam, pKF->mbf, 0.f, pKF->mfLogScaleFactor, pKF->mvScaleFactors.data(), pKF->mnScaleLevels, th, proj.data(), q.data()) != ORBX_OK ||
        orbm_frame_search_fuse(pKF->mpHipFrame.get(), q.data(), pts.desc.data(), nPoints, nullptr, 0, best.data(), idx.data()) != ORBX_OK)
        return 0;
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;                // :1203-1205
        if (!proj[iMP].visible || idx[iMP] < 0) continue;
        if (best[iMP] <= TH_LOW) {                                              // :1279
            MapPoint *pMPinKF = pKF->GetMapPoint(idx[iMP]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, idx[iMP]);
                pKF->AddMapPoint(pMP, idx[iMP]);
            }
            nFused++;
        }
    }
    return nFused;
}

// ---- protected helpers the header still declares; nothing on the hot path calls them any more (their work is inside the entries above)
void ORBmatcher::ComputeThreeMaxima(vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)   // :1802-1843
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2)   // :341-358
{
    const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
    const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
    const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

} // namespace ORB_SLAM2
