#include "MultiStreamRunner.h"
#include <stdexcept>
#include <string>

void MultiStreamRunner::check(int status, const char* what, mslam_ctx* c) const
{
    if (status == MSLAM_OK) return;
    std::string m = std::string(what) + ": " + mslam_strerror(status);
    if (status == MSLAM_ERR_HIP && c) m += std::string(" (") + mslam_last_hip_error(c) + ")";
    throw std::runtime_error(m);
}

MultiStreamRunner::MultiStreamRunner(const Config& cfg) : cfg_(cfg)
{
    if (cfg.streams < 1 || cfg.ring < 2 || cfg.frameSets < 1 || cfg.matchSets < 1) throw std::runtime_error("MultiStreamRunner: bad configuration");
    const int B = cfg.streams;
    cap_ = cfg.orb.nfeatures + cfg.capacitySlack;
    const int max_kpts = cap_ < 4096 ? 4096 : cap_;
    // no fallback: without a gfx950 device this throws
    check(mslam_create(&ctx_, cfg.device, cfg.w, cfg.h, max_kpts, B), "mslam_create", nullptr);
    if (cfg.pnpOnSecondStream) check(mslam_create(&pnp_ctx_, cfg.device, cfg.w, cfg.h, max_kpts, B), "mslam_create", nullptr);
    else pnp_ctx_ = ctx_;
    auto alloc = [&](size_t bytes) { void* p = nullptr; check(mslam_dev_alloc(ctx_, &p, bytes), "mslam_dev_alloc", ctx_); owned_.push_back(p); return p; };
    const size_t frame_bytes = (size_t)cfg.w * cfg.channels * cfg.h;
    for (int s = 0; s < cfg.frameSets; s++) frames_.push_back((uint8_t*)alloc(frame_bytes * B));
    for (int r = 0; r < cfg.ring; r++) {
        kps_.push_back((mslam_keypoint*)alloc(sizeof(mslam_keypoint) * (size_t)B * cap_));
        desc_.push_back((uint8_t*)alloc((size_t)B * cap_ * 32));
        n_.push_back((int32_t*)alloc(sizeof(int32_t) * B));
        std::vector<int32_t> z((size_t)B, 0);
        check(mslam_memcpy_h2d(ctx_, n_.back(), z.data(), sizeof(int32_t) * B), "mslam_memcpy_h2d", ctx_);
    }
    status_ = (int32_t*)alloc(sizeof(int32_t) * B);
    const size_t S = (size_t)cfg.matchSets;
    idx_ = (int32_t*)alloc(sizeof(int32_t) * S * B * cap_ * 2);
    dist_ = (int32_t*)alloc(sizeof(int32_t) * S * B * cap_ * 2);
    mq_ = (int32_t*)alloc(sizeof(int32_t) * S * B * cap_);
    mt_ = (int32_t*)alloc(sizeof(int32_t) * S * B * cap_);
    md_ = (float*)alloc(sizeof(float) * S * B * cap_);
    mc_ = (int32_t*)alloc(sizeof(int32_t) * S * B);
}

MultiStreamRunner::~MultiStreamRunner()
{
    if (ctx_) (void)mslam_sync(ctx_);
    if (pnp_ctx_ && pnp_ctx_ != ctx_) (void)mslam_sync(pnp_ctx_);
    if (copy_ctx_) {
        (void)mslam_sync(copy_ctx_);
        for (void* e : copy_ev_) if (e) (void)mslam_event_destroy(copy_ctx_, e);
        mslam_destroy(copy_ctx_);
    }
    if (kps_pinned_) (void)mslam_host_free(ctx_, kps_pinned_);
    for (void* p : owned_pnp_) (void)mslam_dev_free(pnp_ctx_, p);
    for (void* p : owned_) (void)mslam_dev_free(ctx_, p);
    if (pnp_ctx_ && pnp_ctx_ != ctx_) mslam_destroy(pnp_ctx_);
    if (ctx_) mslam_destroy(ctx_);
}

void MultiStreamRunner::uploadFrames(int set, const uint8_t* host_frames)
{
    check(mslam_memcpy_h2d(ctx_, frames_.at((size_t)set), host_frames, (size_t)cfg_.w * cfg_.channels * cfg_.h * cfg_.streams), "mslam_memcpy_h2d", ctx_);
}

void MultiStreamRunner::uploadFramesAsync(int set, const uint8_t* pinned_frames)
{
    check(mslam_memcpy_h2d_async(ctx_, frames_.at((size_t)set), pinned_frames, (size_t)cfg_.w * cfg_.channels * cfg_.h * cfg_.streams), "mslam_memcpy_h2d_async", ctx_);
}

void MultiStreamRunner::uploadFramesPrefetch(int set, const uint8_t* pinned_frames)
{
    if (!copy_ctx_) {
        check(mslam_create(&copy_ctx_, cfg_.device, cfg_.w, cfg_.h, 4096, 1), "mslam_create", nullptr);   // a stream; its kernels' buffers are never touched
        copy_ev_.assign((size_t)cfg_.frameSets, nullptr);
        copy_pending_.assign((size_t)cfg_.frameSets, 0);
        for (void*& e : copy_ev_) check(mslam_event_create(copy_ctx_, &e), "mslam_event_create", copy_ctx_);
    }
    check(mslam_memcpy_h2d_async(copy_ctx_, frames_.at((size_t)set), pinned_frames, (size_t)cfg_.w * cfg_.channels * cfg_.h * cfg_.streams), "mslam_memcpy_h2d_async", copy_ctx_);
    check(mslam_event_record(copy_ctx_, copy_ev_[(size_t)set]), "mslam_event_record", copy_ctx_);
    copy_pending_[(size_t)set] = 1;
}

void MultiStreamRunner::extract(int set, int slot)
{
    if (copy_ctx_ && copy_pending_.at((size_t)set)) {   // the frames of this set arrive on the copy stream
        check(mslam_stream_wait_event(ctx_, copy_ev_[(size_t)set]), "mslam_stream_wait_event", ctx_);
        copy_pending_[(size_t)set] = 0;
    }
    const int stride = cfg_.w * cfg_.channels;
    check(mslam_orb_extract_batch_dev(ctx_, frames_.at((size_t)set), cfg_.streams, cfg_.w, cfg_.h, stride, (int64_t)stride * cfg_.h, cfg_.channels,
                                      &cfg_.orb, kps_.at((size_t)slot), desc_.at((size_t)slot), cap_, n_.at((size_t)slot), status_),
          "mslam_orb_extract_batch_dev", ctx_);
}

void MultiStreamRunner::match(int slot_query, int slot_train, float ratio, int set)
{
    if (set < 0 || set >= cfg_.matchSets) throw std::runtime_error("MultiStreamRunner::match: no such match set");
    const size_t o = (size_t)set * cfg_.streams * cap_;
    check(mslam_match_knn2_batch_dev(ctx_, desc_.at((size_t)slot_query), n_.at((size_t)slot_query), desc_.at((size_t)slot_train), n_.at((size_t)slot_train),
                                     cap_, cfg_.streams, idx_ + 2 * o, dist_ + 2 * o, ratio, mq_ + o, mt_ + o, md_ + o, mc_ + (size_t)set * cfg_.streams),
          "mslam_match_knn2_batch_dev", ctx_);
}

void MultiStreamRunner::setPnpProblems(int capP, int H, const double* X, const double* x, const int32_t* P, const double* K, const double* dist,
                                       const int32_t* samples, const mslam_pnp_params& prm, const uint64_t* seeds)
{
    if (!samples && !seeds) throw std::runtime_error("MultiStreamRunner::setPnpProblems: neither sample tables nor seeds");
    const int B = cfg_.streams;
    check(mslam_sync(pnp_ctx_), "mslam_sync", pnp_ctx_);
    if (capP != capP_ || H != H_) {
        for (void* p : owned_pnp_) (void)mslam_dev_free(pnp_ctx_, p);
        owned_pnp_.clear();
        auto alloc = [&](size_t bytes) { void* p = nullptr; check(mslam_dev_alloc(pnp_ctx_, &p, bytes), "mslam_dev_alloc", pnp_ctx_); owned_pnp_.push_back(p); return p; };
        pX_ = (double*)alloc(sizeof(double) * (size_t)B * capP * 3); px_ = (double*)alloc(sizeof(double) * (size_t)B * capP * 2);
        pK_ = (double*)alloc(sizeof(double) * B * 9); pD_ = (double*)alloc(sizeof(double) * B * 8);
        pP_ = (int32_t*)alloc(sizeof(int32_t) * B); pS_ = (int32_t*)alloc(sizeof(int32_t) * (size_t)B * H * 4);
        p_Rt_ = (double*)alloc(sizeof(double) * B * 12); p_counts_ = (int32_t*)alloc(sizeof(int32_t) * (size_t)B * H);
        p_best_ = (int32_t*)alloc(sizeof(int32_t) * B); p_ninl_ = (int32_t*)alloc(sizeof(int32_t) * B);
        p_inl_ = (uint8_t*)alloc((size_t)B * capP); pSeed_ = (uint64_t*)alloc(sizeof(uint64_t) * B);
        capP_ = capP; H_ = H;
    }
    auto up = [&](void* d, const void* s, size_t n) { check(mslam_memcpy_h2d(pnp_ctx_, d, s, n), "mslam_memcpy_h2d", pnp_ctx_); };
    up(pX_, X, sizeof(double) * (size_t)B * capP * 3); up(px_, x, sizeof(double) * (size_t)B * capP * 2);
    up(pK_, K, sizeof(double) * B * 9); up(pD_, dist, sizeof(double) * B * 8); up(pP_, P, sizeof(int32_t) * B);
    if (samples) up(pS_, samples, sizeof(int32_t) * (size_t)B * H * 4);
    else {
        up(pSeed_, seeds, sizeof(uint64_t) * B);
        check(mslam_sample_tables_dev(pnp_ctx_, pP_, pSeed_, H, 4, B, pS_), "mslam_sample_tables_dev", pnp_ctx_);
    }
    pnp_prm_ = prm;
}

void MultiStreamRunner::pnp()
{
    if (!capP_) throw std::runtime_error("MultiStreamRunner::pnp: no problems set");
    check(mslam_pnp_ransac_batch_dev(pnp_ctx_, pX_, px_, pP_, capP_, pK_, pD_, pS_, H_, cfg_.streams, &pnp_prm_, p_Rt_, nullptr, nullptr, p_inl_,
                                     p_counts_, p_best_, p_ninl_), "mslam_pnp_ransac_batch_dev", pnp_ctx_);
}

void MultiStreamRunner::sync()
{
    check(mslam_sync(ctx_), "mslam_sync", ctx_);
    if (pnp_ctx_ != ctx_) check(mslam_sync(pnp_ctx_), "mslam_sync", pnp_ctx_);
    if (copy_ctx_) check(mslam_sync(copy_ctx_), "mslam_sync", copy_ctx_);
}

void MultiStreamRunner::downloadCounts(int slot, std::vector<int32_t>& n, std::vector<int32_t>* status)
{
    n.resize((size_t)cfg_.streams);
    check(mslam_memcpy_d2h(ctx_, n.data(), n_.at((size_t)slot), sizeof(int32_t) * cfg_.streams), "mslam_memcpy_d2h", ctx_);
    if (status) { status->resize((size_t)cfg_.streams); check(mslam_memcpy_d2h(ctx_, status->data(), status_, sizeof(int32_t) * cfg_.streams), "mslam_memcpy_d2h", ctx_); }
}

void MultiStreamRunner::downloadFeatures(int slot, int stream, int n, std::vector<mslam_keypoint>& kps, uint8_t* desc)
{
    kps.resize((size_t)n);
    if (n <= 0) return;
    check(mslam_memcpy_d2h(ctx_, kps.data(), kps_.at((size_t)slot) + (size_t)stream * cap_, sizeof(mslam_keypoint) * (size_t)n), "mslam_memcpy_d2h", ctx_);
    if (desc) check(mslam_memcpy_d2h(ctx_, desc, desc_.at((size_t)slot) + (size_t)stream * cap_ * 32, (size_t)n * 32), "mslam_memcpy_d2h", ctx_);
}

void MultiStreamRunner::downloadKeypoints(int slot, std::vector<mslam_keypoint>& kps)
{
    kps.resize((size_t)cfg_.streams * cap_);
    check(mslam_memcpy_d2h(ctx_, kps.data(), kps_.at((size_t)slot), sizeof(mslam_keypoint) * kps.size()), "mslam_memcpy_d2h", ctx_);
}

const mslam_keypoint* MultiStreamRunner::downloadKeypointsPinned(int slot)
{
    const size_t bytes = sizeof(mslam_keypoint) * (size_t)cfg_.streams * cap_;
    if (!kps_pinned_) check(mslam_host_alloc(ctx_, &kps_pinned_, bytes), "mslam_host_alloc", ctx_);
    check(mslam_memcpy_d2h_async(ctx_, kps_pinned_, kps_.at((size_t)slot), bytes), "mslam_memcpy_d2h_async", ctx_);
    check(mslam_sync(ctx_), "mslam_sync", ctx_);
    return (const mslam_keypoint*)kps_pinned_;
}

void MultiStreamRunner::downloadMatches(std::vector<int32_t>& count, std::vector<int32_t>& query, std::vector<int32_t>& train, std::vector<float>& dist, int nsets)
{
    if (nsets < 1 || nsets > cfg_.matchSets) throw std::runtime_error("MultiStreamRunner::downloadMatches: no such match set");
    const size_t B = (size_t)cfg_.streams * (size_t)nsets, n = B * cap_;
    count.resize(B); query.resize(n); train.resize(n); dist.resize(n);
    check(mslam_memcpy_d2h_async(ctx_, query.data(), mq_, sizeof(int32_t) * n), "mslam_memcpy_d2h_async", ctx_);
    check(mslam_memcpy_d2h_async(ctx_, train.data(), mt_, sizeof(int32_t) * n), "mslam_memcpy_d2h_async", ctx_);
    check(mslam_memcpy_d2h_async(ctx_, dist.data(), md_, sizeof(float) * n), "mslam_memcpy_d2h_async", ctx_);
    check(mslam_memcpy_d2h(ctx_, count.data(), mc_, sizeof(int32_t) * B), "mslam_memcpy_d2h", ctx_);
}

void MultiStreamRunner::downloadKnn(int stream, int nq, std::vector<int32_t>& idx, std::vector<int32_t>& dist)
{
    idx.resize((size_t)nq * 2); dist.resize((size_t)nq * 2);
    if (nq <= 0) return;
    check(mslam_memcpy_d2h_async(ctx_, idx.data(), idx_ + (size_t)stream * cap_ * 2, sizeof(int32_t) * (size_t)nq * 2), "mslam_memcpy_d2h_async", ctx_);
    check(mslam_memcpy_d2h(ctx_, dist.data(), dist_ + (size_t)stream * cap_ * 2, sizeof(int32_t) * (size_t)nq * 2), "mslam_memcpy_d2h", ctx_);
}

void MultiStreamRunner::downloadPnp(std::vector<double>& Rt, std::vector<int32_t>& best, std::vector<int32_t>& ninl, std::vector<int32_t>* counts,
                                    std::vector<uint8_t>* inliers)
{
    const size_t B = (size_t)cfg_.streams;
    Rt.resize(B * 12); best.resize(B); ninl.resize(B);
    check(mslam_memcpy_d2h_async(pnp_ctx_, Rt.data(), p_Rt_, sizeof(double) * B * 12), "mslam_memcpy_d2h_async", pnp_ctx_);
    check(mslam_memcpy_d2h_async(pnp_ctx_, best.data(), p_best_, sizeof(int32_t) * B), "mslam_memcpy_d2h_async", pnp_ctx_);
    if (counts) { counts->resize(B * H_); check(mslam_memcpy_d2h_async(pnp_ctx_, counts->data(), p_counts_, sizeof(int32_t) * B * H_), "mslam_memcpy_d2h_async", pnp_ctx_); }
    if (inliers) { inliers->resize(B * capP_); check(mslam_memcpy_d2h_async(pnp_ctx_, inliers->data(), p_inl_, B * capP_), "mslam_memcpy_d2h_async", pnp_ctx_); }
    check(mslam_memcpy_d2h(pnp_ctx_, ninl.data(), p_ninl_, sizeof(int32_t) * B), "mslam_memcpy_d2h", pnp_ctx_);
}
