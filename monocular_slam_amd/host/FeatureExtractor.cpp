#include "FeatureExtractor.h"

FeatureExtractor::FeatureExtractor() : ProcessingNode("FeatureExtractor") { mslam_orb_default_params(&params); }

void FeatureExtractor::extract(const cv::Mat& img, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors)
{
    const int cap = params.nfeatures + capacitySlack;
    MslamRuntime& rt = MslamRuntime::instance();
    rt.ensure(img.cols, img.rows, cap);
    std::vector<mslam_keypoint> kps((size_t)cap);
    cv::Mat desc(cap, 32, cv::CV_8U);
    int n = 0;
    MslamRuntime::check(mslam_orb_extract(rt.ctx(), img.data, img.cols, img.rows, (int)img.step, img.channels(), &params,
                                          kps.data(), desc.data, cap, &n), "mslam_orb_extract");
    keypoints.resize((size_t)n);
    descriptors.create(n, 32, cv::CV_8U);
    for (int i = 0; i < n; i++) {
        cv::KeyPoint& k = keypoints[(size_t)i];
        k.pt = cv::Point2f(kps[(size_t)i].x, kps[(size_t)i].y); k.size = kps[(size_t)i].size; k.angle = kps[(size_t)i].angle;
        k.response = kps[(size_t)i].response; k.octave = kps[(size_t)i].octave;
    }
    if (n > 0) std::memcpy(descriptors.data, desc.data, (size_t)n * 32);
}

void FeatureExtractor::process(DataManager& data, int frameIdx)
{
    Frame& frame = data.frames[(size_t)frameIdx];
    std::vector<cv::KeyPoint> keypoints;
    cv::Mat descriptor;
    extract(frame.frameBuffer, keypoints, descriptor);
    frame.features.descriptors = descriptor;
    for (const cv::KeyPoint& k : keypoints) {
        frame.features.positions.push_back(Point2d(k.pt));      // Point2f -> Point2d widening
        frame.features.scales.push_back((double)k.size);
    }
    frame.features.mapPointsIndices.resize(keypoints.size(), -1);
}

bool FeatureExtractor::validationCheck(DataManager&, int) { return true; }
