#include "FeatureExtractor.h"

FeatureExtractor::FeatureExtractor() : ProcessingNode("FeatureExtractor") { mslam_orb_default_params(&params); }

void FeatureExtractor::extract(const cv::Mat& img, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors)
{
    const int cap = params.nfeatures + capacitySlack;
    MslamRuntime& rt = MslamRuntime::instance();
    rt.ensure(img.cols, img.rows, cap);
    mslam_ctx* ctx = rt.ctx();
    // the frame goes up once; keypoints and descriptors are produced on the device, the descriptors stay there as the
    // twin of `descriptors` (so that matching never uploads them again) and a host copy is taken for the data model
    const size_t img_bytes = img.step * (size_t)img.rows;
    uint8_t* dimg = (uint8_t*)rt.scratch(0, img_bytes + 256);
    mslam_keypoint* dk = (mslam_keypoint*)rt.scratch(1, sizeof(mslam_keypoint) * (size_t)cap + 256);
    int32_t* dstatus = (int32_t*)((uint8_t*)dk + sizeof(mslam_keypoint) * (size_t)cap);
    std::shared_ptr<DeviceFeatures> twin = rt.newDeviceFeatures(cap);
    MslamRuntime::check(mslam_memcpy_h2d_async(ctx, dimg, img.data, img_bytes), "mslam_memcpy_h2d_async");
    MslamRuntime::check(mslam_orb_extract_batch_dev(ctx, dimg, 1, img.cols, img.rows, (int)img.step, (int64_t)((img_bytes + 255) & ~(size_t)255),
                                                    img.channels(), &params, dk, twin->desc, cap, twin->n_dev, dstatus), "mslam_orb_extract_batch_dev");
    int32_t hn = 0, hs = 0;
    MslamRuntime::check(mslam_memcpy_d2h_async(ctx, &hn, twin->n_dev, sizeof(hn)), "mslam_memcpy_d2h_async");
    MslamRuntime::check(mslam_memcpy_d2h(ctx, &hs, dstatus, sizeof(hs)), "mslam_memcpy_d2h");   // syncs: hn and hs are valid
    MslamRuntime::check(hs, "mslam_orb_extract");
    const int n = hn;
    twin->n = n;
    std::vector<mslam_keypoint> kps((size_t)n);
    descriptors.create(n, 32, cv::CV_8U);
    if (n > 0) {
        MslamRuntime::check(mslam_memcpy_d2h_async(ctx, kps.data(), dk, sizeof(mslam_keypoint) * (size_t)n), "mslam_memcpy_d2h_async");
        MslamRuntime::check(mslam_memcpy_d2h(ctx, descriptors.data, twin->desc, (size_t)n * 32), "mslam_memcpy_d2h");
    }
    twin->content = contentHash(descriptors.data, (size_t)n * 32);
    descriptors.device = twin;
    keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        cv::KeyPoint& k = keypoints[(size_t)i];
        k.pt = cv::Point2f(kps[(size_t)i].x, kps[(size_t)i].y); k.size = kps[(size_t)i].size; k.angle = kps[(size_t)i].angle;
        k.response = kps[(size_t)i].response; k.octave = kps[(size_t)i].octave;
    }
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
