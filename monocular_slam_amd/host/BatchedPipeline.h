// BatchedPipeline -- ProcessingPipeline(FeatureExtractor -> CameraPoseEstimator) for B independent sequences at once.
// The reference runs one sequence, one frame at a time, single-threaded (src/main.cpp:40-51, src/Pipeline.h:49-65); frame i
// of a sequence depends on frame i-1 of the same sequence only (src/CameraPoseEstimator.cpp:388,405), so B sequences can
// share every launch.  process(i) does for all of them what ProcessingPipeline::process(dm, i) does for one:
//   FeatureExtractor::process  (src/FeatureExtractor.cpp:13-31)   one mslam_orb_extract_batch_dev over B frames; features stay
//                                                                  on the device in a ring of the last 1 + numBackTraverse frames
//   CameraPoseEstimator::process (src/CameraPoseEstimator.cpp:517-530)
//     matchFeatures :200-213       mslam_match_knn2_batch_dev on the device-resident descriptors (no upload, ratio test on device)
//     computeFundamentalMatrix :545-586   mslam_fmat_ransac8_batch_dev over the B match sets
//     solvePnPRansac :472          mslam_pnp_ransac_batch_dev over the B 3D-2D sets
//   association, essential-matrix decomposition, cheirality vote and map bookkeeping run per stream on the host, as in the
//   reference (:218-243, :316-356, :432-452).
// Every stream gets exactly what the single-stream ProcessingPipeline of this repository gives it (same kernels, same
// parameters, same sample tables): tests/test_pipeline_gpu.py compares the two.  Frame.features.descriptors is filled lazily
// (descriptors()/syncFeatures()) -- nothing in the batched stages needs it on the host.
#pragma once
#include <memory>
#include <vector>
#include "MultiStreamRunner.h"
#include "Stages.h"

class BatchedPipeline {
public:
    // the stage objects carry the parameters (same defaults as the single-stream stages)
    BatchedPipeline(std::vector<DataManager*> streams, const FeatureExtractor& fe = FeatureExtractor(),
                    const CameraPoseEstimator& cpe = CameraPoseEstimator(), int device = 0);
    ~BatchedPipeline();
    int streams() const { return (int)dms_.size(); }
    // one tick: frames[frameIdx] of every stream (all must have the same size / channel count)
    void process(int frameIdx);
    // host copy of frame `frameIdx`'s descriptors for one stream (downloads on first use while the frame is still in the ring)
    const cv::Mat& descriptors(int stream, int frameIdx);
    std::vector<int> lastStatus;       // per stream: MSLAM_OK or the status that made the pose stage keep the previous pose
    MultiStreamRunner* runner() { return runner_.get(); }
    bool eagerDescriptors = false;     // true: fill Frame.features.descriptors for every frame right after extraction
    // true: while frame i is being extracted, frame i + 1 (if every stream already holds it, as after FrameLoader::load --
    // src/main.cpp:35-37) is staged and copied to the device on a copy stream.  A frame buffer that is replaced between
    // process(i) and process(i + 1) would be missed: switch this off for such a caller.
    bool prefetchNextFrame = true;
    // host wall time per phase, accumulated over the ticks (milliseconds; device work counts where the host waits for it)
    enum Phase { PH_STAGE_FRAMES, PH_EXTRACT, PH_FEATURES_DOWN, PH_MATCH, PH_FMAT_PREP, PH_FMAT, PH_ASSOCIATE, PH_PNP_PREP, PH_PNP, PH_TRIANGULATE, PH_COUNT };
    double phaseMs[PH_COUNT] = { 0 };
    static const char* phaseName(int p);

private:
    void extractAll(int frameIdx);
    void poseInitial(int frameIdx);
    void posePnp(int frameIdx);
    // match lists + epipolar inlier masks of N = sets x B problems (problem k * B + b = set k, stream b), viewed in the pinned
    // download buffer: valid until the next matchFilterMany
    struct Filtered {
        int cap = 0;
        const int32_t *count = nullptr, *query = nullptr, *train = nullptr;   // N, N x cap, N x cap
        const float* dist = nullptr;                                           // N x cap
        const uint8_t* status = nullptr;                                       // N x cap
        const double* F = nullptr;                                             // N x 9
        std::vector<int> rc;                                                   // N: MSLAM_OK / TOO_FEW / DEGENERATE
        uint8_t inlier(int q, int i) const { return rc[(size_t)q] == MSLAM_OK ? status[(size_t)q * cap + i] : 0; }
    };
    void matchFilterMany(int slotQ, const std::vector<int>& slotsT, float ratio, const std::vector<uint64_t>& seeds, Filtered& out);
    int slotOf(int frameIdx) const { return frameIdx % ring_; }
    void* scratch(int slot, size_t bytes);

    std::vector<DataManager*> dms_;
    FeatureExtractor fe_;
    CameraPoseEstimator cpe_;
    int device_, ring_, w_ = 0, h_ = 0, ch_ = 0;
    std::unique_ptr<MultiStreamRunner> runner_;
    void* pinned_ = nullptr; size_t pinned_bytes_ = 0;   // two staging buffers of B frames (frame i uses half i & 1)
    int prefetched_ = -1;                                 // frame index already on its way to the device
    void stageFrames(int frameIdx);
    bool scratch_sized_ = false;
    std::vector<int> slotFeatures_;    // per ring slot: the largest feature count over the streams
    void* arena_ = nullptr; size_t arena_bytes_ = 0;   // pinned up/download buffer of matchFilterMany
    void* dscratch_[4] = { nullptr, nullptr, nullptr, nullptr }; size_t dscratch_bytes_[4] = { 0, 0, 0, 0 };
};
