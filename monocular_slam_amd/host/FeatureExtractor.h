// FeatureExtractor -- the reference's stage (src/FeatureExtractor.h:16-25, src/FeatureExtractor.cpp:13-31) with its two
// OpenCV ORB members replaced by the C ABI.  `extract` is the new entry point north_star names; process() calls it and
// fills frame.features exactly as the reference does (:20-25).
#pragma once
#include "ProcessingNode.h"
#include "Runtime.h"

class FeatureExtractor : public ProcessingNode {
public:
    FeatureExtractor();
    // detector.detect + extractor.compute (:17,:19): keypoints and N x 32 descriptors of one frame
    void extract(const cv::Mat& frameBuffer, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors);
    virtual void process(DataManager& data, int frameIdx) override;
    virtual bool validationCheck(DataManager& data, int frameIdx) override;
    mslam_orb_params params;   // defaults = default-constructed cv::ORB (500 features, 1.2, 8 levels, 31, HARRIS, 31)
    int capacitySlack = 1024;  // room for response ties (retainBest keeps every tie)
};
