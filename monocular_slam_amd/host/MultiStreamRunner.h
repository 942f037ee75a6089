// MultiStreamRunner -- the batched, device-resident form of the tracking front end for C++ hosts: B independent streams
// (one DataManager each in BatchedPipeline) advance one frame per tick through
//     mslam_orb_extract_batch_dev -> mslam_match_knn2_batch_dev -> mslam_pnp_ransac_batch_dev
// with keypoints and descriptors kept on the device between stages, in a ring of the last `ring` frames per stream (the
// reference matches a frame against up to 5 previous ones, src/CameraPoseEstimator.cpp:388,405-409).  Nothing is
// downloaded unless asked for.  It is what one ProcessingPipeline::process(dm, i) tick of the reference
// (src/Pipeline.h:49-65; FeatureExtractor::process src/FeatureExtractor.cpp:13-31, matchFeatures
// src/CameraPoseEstimator.cpp:200-213, solvePnPRansac :472) becomes when B streams share one GPU.
// Two HIP streams (contexts): extract+match on one, the PnP chain on the other, so that the f64 RANSAC kernels overlap the
// integer image kernels of the same tick (outputs are unchanged by the overlap).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/mslam.h"

class MultiStreamRunner {
public:
    struct Config {
        int device = 0, streams = 32, w = 1920, h = 1080, channels = 1;
        mslam_orb_params orb;          // defaults of mslam_orb_default_params
        int capacitySlack = 1024;      // room for response ties (retainBest keeps every tie)
        int ring = 2;                  // feature slots kept per stream (current + previous ones)
        int frameSets = 2;             // device-resident frame buffers (ingest double-buffering / alternating sets)
        int matchSets = 1;             // match result buffers: match(.., set) calls of one tick can be enqueued back to back and read with one download
        bool pnpOnSecondStream = true;
        Config() { mslam_orb_default_params(&orb); }
    };
    explicit MultiStreamRunner(const Config& cfg);
    ~MultiStreamRunner();
    MultiStreamRunner(const MultiStreamRunner&) = delete;
    MultiStreamRunner& operator=(const MultiStreamRunner&) = delete;

    int streams() const { return cfg_.streams; }
    int cap() const { return cap_; }
    mslam_ctx* ctx() { return ctx_; }
    mslam_ctx* pnpCtx() { return pnp_ctx_; }

    // ---- frames: B images of h x stride bytes, contiguous; `set` selects the resident buffer
    void uploadFrames(int set, const uint8_t* host_frames);            // synchronous
    void uploadFramesAsync(int set, const uint8_t* pinned_frames);     // enqueued on the extract stream
    // enqueued on a copy stream of its own (created on first use): overlaps the kernels running on the extract stream; the
    // next extract(set, ..) waits for it on the device (event), the host does not
    void uploadFramesPrefetch(int set, const uint8_t* pinned_frames);
    // ---- stages (enqueue only; no host synchronisation)
    void extract(int set, int slot);                                   // frames[set] -> feature ring slot
    void match(int slot_query, int slot_train, float ratio, int set = 0); // kNN-2 + ratio test + ordered compaction, results into match set `set`
    // B PnP problems resident on the device: X (B x capP x 3), x (B x capP x 2), P (B), K (B x 9), dist (B x 8), samples (B x H x 4)
    // samples == nullptr: the tables are drawn on the device from `seeds` (B values; mslam_sample_tables_dev == geom::sampleTable(P[b], H, 4, seeds[b]))
    void setPnpProblems(int capP, int H, const double* X, const double* x, const int32_t* P, const double* K, const double* dist,
                        const int32_t* samples, const mslam_pnp_params& prm, const uint64_t* seeds = nullptr);
    void pnp();
    void sync();

    // ---- results (synchronise the stream they come from)
    void downloadCounts(int slot, std::vector<int32_t>& n, std::vector<int32_t>* status = nullptr);
    void downloadFeatures(int slot, int stream, int n, std::vector<mslam_keypoint>& kps, uint8_t* desc /* n x 32 or null */);
    void downloadKeypoints(int slot, std::vector<mslam_keypoint>& kps /* B x cap, one copy for all streams */);
    // the same through a pinned buffer owned by the runner (valid until the next call): B x cap keypoints
    const mslam_keypoint* downloadKeypointsPinned(int slot);
    // the first `nsets` match sets: count (nsets x B), query / train / dist (nsets x B x cap)
    void downloadMatches(std::vector<int32_t>& count, std::vector<int32_t>& query, std::vector<int32_t>& train, std::vector<float>& dist, int nsets = 1);
    void downloadKnn(int stream, int nq, std::vector<int32_t>& idx, std::vector<int32_t>& dist);
    void downloadPnp(std::vector<double>& Rt /*B x 12*/, std::vector<int32_t>& best, std::vector<int32_t>& ninl, std::vector<int32_t>* counts = nullptr /*B x H*/,
                     std::vector<uint8_t>* inliers = nullptr /*B x capP*/);
    // device views for callers that chain their own _dev calls
    const uint8_t* descDev(int slot) const { return desc_[slot]; }
    const mslam_keypoint* kpsDev(int slot) const { return kps_[slot]; }
    // match results, all sets contiguous: set s, stream b at (s * B + b) [* cap]
    const int32_t* matchQueryDev() const { return mq_; }
    const int32_t* matchTrainDev() const { return mt_; }
    const float* matchDistDev() const { return md_; }
    const int32_t* matchCountDev() const { return mc_; }
    const int32_t* countDev(int slot) const { return n_[slot]; }
    const int32_t* pnpCountsDev() const { return p_counts_; }
    int pnpHyps() const { return H_; }

private:
    void check(int status, const char* what, mslam_ctx* c) const;
    Config cfg_;
    int cap_ = 0, capP_ = 0, H_ = 0;
    mslam_ctx *ctx_ = nullptr, *pnp_ctx_ = nullptr, *copy_ctx_ = nullptr;
    std::vector<void*> copy_ev_;        // per frame set: recorded behind the prefetch copy
    std::vector<char> copy_pending_;
    std::vector<uint8_t*> frames_;
    std::vector<mslam_keypoint*> kps_;
    std::vector<uint8_t*> desc_;
    std::vector<int32_t*> n_;
    int32_t *status_ = nullptr, *idx_ = nullptr, *dist_ = nullptr, *mq_ = nullptr, *mt_ = nullptr, *mc_ = nullptr;
    float* md_ = nullptr;
    double *pX_ = nullptr, *px_ = nullptr, *pK_ = nullptr, *pD_ = nullptr, *p_Rt_ = nullptr;
    uint64_t* pSeed_ = nullptr;
    int32_t *pP_ = nullptr, *pS_ = nullptr, *p_counts_ = nullptr, *p_best_ = nullptr, *p_ninl_ = nullptr;
    uint8_t* p_inl_ = nullptr;
    mslam_pnp_params pnp_prm_{};
    std::vector<void*> owned_, owned_pnp_;
    void* kps_pinned_ = nullptr;
};
