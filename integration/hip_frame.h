// hip_frame.h -- the resident-frame handle a Frame / KeyFrame carries, and the flattening of the pointer graph the whole-function
// searches take (include/orbslam_hip.h: orbm_frame, orbm_points, orbm_view).  Reference-side code: compiles in the ORB_SLAM2_E tree.
#ifndef HIP_FRAME_H
#define HIP_FRAME_H

#include <memory>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>

#include "orbslam_hip.h"

namespace ORB_SLAM2 {

class MapPoint;

// Frame is copied by value (mLastFrame = Frame(mCurrentFrame), src/Tracking.cc) and a KeyFrame is made from a Frame: the device data is
// shared, never copied.
typedef std::shared_ptr<orbm_frame> HipFramePtr;

inline HipFramePtr HipFrameAdopt(orbm_frame *f) { return HipFramePtr(f, [](orbm_frame *p) { orbm_frame_destroy(p); }); }

// End of a Frame constructor (after UndistortKeyPoints / ComputeStereoMatches / AssignFeaturesToGrid): keypoints and descriptors are
// taken where the extractor left them, in HBM; only the undistorted coordinates (when the camera has distortion) and mvuRight travel.
inline HipFramePtr HipFrameFromExtractor(orbx_extractor *ex, const std::vector<cv::KeyPoint> &mvKeys, const std::vector<cv::KeyPoint> &mvKeysUn,
                                         const cv::Mat &mDistCoef, const std::vector<float> &mvuRight, bool stereoOnDevice, float mnMinX,
                                         float mnMinY, float mnMaxX, float mnMaxY)
{
    std::vector<float> xy;
    if (mDistCoef.at<float>(0) != 0.0) {                 // Frame::UndistortKeyPoints (src/Frame.cc:419-424): otherwise mvKeysUn = mvKeys
        xy.resize(2 * mvKeysUn.size());
        for (size_t i = 0; i < mvKeysUn.size(); ++i) { xy[2 * i] = mvKeysUn[i].pt.x; xy[2 * i + 1] = mvKeysUn[i].pt.y; }
    }
    (void)mvKeys;
    bool anyRight = false;
    for (size_t i = 0; i < mvuRight.size() && !anyRight; ++i) anyRight = mvuRight[i] >= 0;
    orbm_frame *f = nullptr;
    const int rc = orbm_frame_from_extractor(ex, 0, xy.empty() ? nullptr : xy.data(),
                                             (!stereoOnDevice && anyRight) ? mvuRight.data() : nullptr, stereoOnDevice ? 1 : 0,
                                             mnMinX, mnMinY, mnMaxX, mnMaxY, &f);
    return rc == ORBX_OK ? HipFrameAdopt(f) : HipFramePtr();
}

// KeyFrame::KeyFrame(Frame &F, ...): the Frame's grid, the KeyFrame's int-valued bounds (include/KeyFrame.h:201-204)
inline HipFramePtr HipFrameForKeyFrame(const HipFramePtr &frame, int mnMinX, int mnMinY, int mnMaxX, int mnMaxY)
{
    orbm_frame *f = nullptr;
    if (!frame || orbm_frame_alias(frame.get(), (float)mnMinX, (float)mnMinY, (float)mnMaxX, (float)mnMaxY, &f) != ORBX_OK) return HipFramePtr();
    return HipFrameAdopt(f);
}

// A vector<MapPoint*> flattened for orbm_points.  The caller's loop decides valid[i] (the reference's pointer tests) and calls set().
struct HipPointList {
    std::vector<uint8_t> valid, desc, takes;
    std::vector<float> pos, normal, minDistance, maxDistance, angle;
    std::vector<int32_t> octave;
    explicit HipPointList(size_t n) : valid(n, 0), desc(32 * n, 0), takes(n, 1), pos(3 * n, 0.f), normal(3 * n, 0.f), minDistance(n, 0.f),
                                      maxDistance(n, 0.f), angle(n, 0.f), octave(n, 0) {}
    void set(size_t i, MapPoint *pMP);           // defined in ORBmatcher_hip.cc (needs MapPoint.h)
    orbm_points view() const
    {
        orbm_points p;
        p.n = (int32_t)valid.size(); p.valid = valid.data(); p.pos = pos.data(); p.normal = normal.data();
        p.min_distance = minDistance.data(); p.max_distance = maxDistance.data(); p.desc = desc.data(); p.takes = takes.data();
        p.octave = octave.data(); p.angle = angle.data();
        return p;
    }
};

inline void HipPose(const cv::Mat &T, float out[16])      // a 4 x 4 (or 3 x 4) CV_32F pose, row-major
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = r < T.rows ? T.at<float>(r, c) : (r == c ? 1.f : 0.f);
}

} // namespace ORB_SLAM2

#endif // HIP_FRAME_H
