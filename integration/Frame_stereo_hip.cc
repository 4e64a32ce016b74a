// Frame_stereo_hip.cc -- Frame::ComputeStereoMatches (src/Frame.cc:527-701) over liborbslam_hip.so; the reference's definition is compiled
// out by integration/reference.patch (ORBSLAM_STEREO_ON_HIP).  Both extractors still hold this frame's keypoints, descriptors and pyramids
// in HBM, so nothing is uploaded: row-band candidates, Hamming best match, the 11-shift L1 refinement on the keypoint's pyramid level,
// the parabola and the median cut run there; mvuRight / mvDepth come back in one small block.
#include "Frame.h"

#include "orbslam_hip.h"

namespace ORB_SLAM2 {

void Frame::ComputeStereoMatches()
{
    mvuRight = vector<float>(N, -1.0f);
    mvDepth = vector<float>(N, -1.0f);
    if (N == 0) return;
    if (orbx_stereo_match(mpORBextractorLeft->mHip, mpORBextractorRight->mHip, mb, mbf, nullptr) != ORBX_OK) return;
    int n = 0;
    orbx_stereo_download(mpORBextractorLeft->mHip, 0, mvuRight.data(), mvDepth.data(), N, &n);
}

} // namespace ORB_SLAM2
