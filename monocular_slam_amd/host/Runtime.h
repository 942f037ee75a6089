// Runtime.h -- process-wide handle on the C ABI context (one per GPU) shared by the stages, plus the device-side
// resources the single-stream stages keep between calls so that features do not bounce through the host:
// a frame's descriptors stay on the GPU (DeviceFeatures, attached to Features::descriptors as cv::Mat::device) from
// FeatureExtractor::extract until the last cv::Mat copy that refers to them dies; CameraPoseEstimator::matchFeatures
// matches two such twins with mslam_match_knn2_batch_dev without any upload (the reference keeps everything in host
// cv::Mats, src/Frame.h:22-34 -- there is no device there to keep anything on).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mslam.h"

// descriptors of one frame on the device: cap x 32 bytes + the row count as a device int32 (what the _dev entry points take)
struct DeviceFeatures {
    uint64_t gen = 0;          // runtime generation this buffer belongs to (a re-created context invalidates older ones)
    uint8_t* desc = nullptr;   // cap x 32, 256-byte aligned
    int32_t* n_dev = nullptr;  // 1 int32: number of rows
    int n = 0, cap = 0;
    uint64_t content = 0;      // contentHash of the host rows the twin was attached to (cv::Mat hands out mutable access:
                               // a caller may edit descriptors in place; the twin is used only while the host rows still hash to this)
};

// 64-bit content hash of n bytes (n a multiple of 8 for descriptor rows; a tail is folded bytewise): four independent
// multiply-rotate lanes, a few microseconds for a 2000 x 32 descriptor matrix
inline uint64_t contentHash(const uint8_t* p, size_t n)
{
    uint64_t h[4] = { 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull };
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        for (int k = 0; k < 4; k++) {
            uint64_t w; __builtin_memcpy(&w, p + i + 8 * k, 8);
            h[k] = (h[k] ^ w) * 0xFF51AFD7ED558CCDull;
            h[k] = (h[k] << 29) | (h[k] >> 35);
        }
    uint64_t r = n;
    for (; i < n; i++) r = (r ^ p[i]) * 0x100000001B3ull;
    for (int k = 0; k < 4; k++) { r ^= h[k]; r *= 0xC4CEB9FE1A85EC53ull; r ^= r >> 31; }
    return r;
}

class MslamRuntime {
public:
    // device from $MSLAM_DEVICE (default 0); sized for the largest frame the pipeline will see
    static MslamRuntime& instance();
    mslam_ctx* ctx() { return ctx_; }
    void ensure(int w, int h, int max_kpts);
    static void check(int status, const char* what) {
        if (status != MSLAM_OK) throw std::runtime_error(std::string(what) + ": " + mslam_strerror(status));
    }
    // a descriptor buffer from the pool (returned to it when the last owner lets go)
    std::shared_ptr<DeviceFeatures> newDeviceFeatures(int cap);
    // the twin behind a cv::Mat::device handle if it is still valid on the current context, else nullptr
    DeviceFeatures* deviceFeatures(const std::shared_ptr<void>& handle) const;
    // grow-only device scratch of the single-stream stages (slot 0: image upload, 1: keypoints + status, 2: match outputs)
    void* scratch(int slot, size_t bytes);
    ~MslamRuntime();
private:
    MslamRuntime() = default;
    void release();
    mslam_ctx* ctx_ = nullptr;
    int w_ = 0, h_ = 0, kp_ = 0;
    uint64_t gen_ = 1;
    struct Pool { int cap; std::vector<void*> free; };
    std::vector<Pool> pools_;
    std::vector<void*> owned_;            // every pool buffer ever allocated on this context
    void* scratch_[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t scratch_bytes_[4] = { 0, 0, 0, 0 };
    static bool alive_;
    friend struct DeviceFeaturesDeleter;
};
