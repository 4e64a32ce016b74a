// ORBextractor_hip.cc -- ORB_SLAM2::ORBextractor over liborbslam_hip.so.  Replaces src/ORBextractor.cc in the reference build
// (integration/reference.patch adds the member `orbx_extractor *mHip;` and a destructor to include/ORBextractor.h).
// Everything the reference computes in ORBextractor.cc:410-470 (constructor tables) and :1051-1140 (operator(), ComputePyramid,
// ComputeKeyPointsOctTree, computeOrientation, GaussianBlur, computeDescriptors) happens on the device behind orbx_extract.
#include "ORBextractor.h"

#include <cassert>

#include "orbslam_hip.h"

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint is 28 bytes: pt.x, pt.y, size, angle, response, octave, class_id");

namespace ORB_SLAM2 {

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mHip(nullptr)
{
    orbx_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, /*blur_variant*/ 0, /*trig_variant*/ 0};
    if (orbx_create(&p, &mHip) != ORBX_OK) { mHip = nullptr; return; }       // orbx_last_error() has the text
    mvScaleFactor.resize(nlevels);      orbx_get_scale_factors(mHip, mvScaleFactor.data());
    mvInvScaleFactor.resize(nlevels);   orbx_get_inv_scale_factors(mHip, mvInvScaleFactor.data());
    mvLevelSigma2.resize(nlevels);      orbx_get_level_sigma2(mHip, mvLevelSigma2.data());
    mvInvLevelSigma2.resize(nlevels);   orbx_get_inv_level_sigma2(mHip, mvInvLevelSigma2.data());
    mnFeaturesPerLevel.resize(nlevels); orbx_get_features_per_level(mHip, mnFeaturesPerLevel.data());
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { if (mHip) orbx_destroy(mHip); }

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty()) return;                                            // ORBextractor.cc:1054-1055
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);                                       // :1058
    if (!mHip || orbx_reserve(mHip, image.cols, image.rows, 1) != ORBX_OK) { _keypoints.clear(); _descriptors.release(); return; }
    const int cap = orbx_keypoint_capacity(mHip);                          // nfeatures + 3 nlevels, more on very wide frames
    _keypoints.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    const int rc = orbx_extract(mHip, image.data, image.cols, image.rows, (int)image.step,
                                reinterpret_cast<orbx_keypoint *>(_keypoints.data()), desc.data, cap, &n);
    if (rc != ORBX_OK) n = 0;
    _keypoints.resize(n);
    if (n == 0) _descriptors.release();                                    // :1072-1073
    else desc.rowRange(0, n).copyTo(_descriptors);                         // :1076
    mbPyramidOnHost = false;                                               // mvImagePyramid is fetched when somebody reads it
}

// The public mvImagePyramid (include/ORBextractor.h:86) is read only by the stereo path (src/Frame.cc:534,624,636,641), which the shell of
// Frame::ComputeStereoMatches no longer needs (orbx_stereo_match works on the device-resident pyramids).  For any other reader:
void ORBextractor::SyncImagePyramid()
{
    if (mbPyramidOnHost || !mHip) return;
    for (int l = 0; l < nlevels; ++l) {
        int w = 0, h = 0;
        orbx_level_size(mHip, l, &w, &h);
        cv::Mat padded(h + 38, w + 38, CV_8U);                             // the 19-px REFLECT_101 border of ComputePyramid (:1123-1137)
        orbx_pyramid_level_padded(mHip, 0, l, padded.data, (int)padded.step);
        mvImagePyramid[l] = padded(cv::Rect(19, 19, w, h));
    }
    mbPyramidOnHost = true;
}

} // namespace ORB_SLAM2
