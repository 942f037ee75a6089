#include "BatchedPipeline.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include "Geometry.h"

namespace {

// run fn(b) for b in [0, n) on the host cores: the per-stream work between two batched launches (sample tables, coordinate
// gathers, association) is independent per stream.  One pool of workers for the process: a tick calls this several times
// and starting threads each time costs more than the work.
class StreamPool {
public:
    StreamPool()
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = std::min<unsigned>(hw ? hw : 1u, 16u);
        for (unsigned t = 1; t < nt; t++) th_.emplace_back([this]() { loop(); });
    }
    ~StreamPool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    void run(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int b = 0; b < n; b++) fn(b); return; }
        std::lock_guard<std::mutex> one(run_m_);   // one batch at a time
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; next_ = 0; pending_ = (int)th_.size(); err_ = nullptr; gen_++; }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&]() { return pending_ == 0; });
        if (err_) std::rethrow_exception(err_);
    }

private:
    void work()
    {
        try { for (int b = next_++; b < n_; b = next_++) (*fn_)(b); }
        catch (...) { std::lock_guard<std::mutex> l(m_); if (!err_) err_ = std::current_exception(); next_ = n_; }
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&]() { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            work();
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{ 0 };
    int n_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    std::exception_ptr err_;
};
template <typename F> void parallelStreams(int n, F fn)
{
    static StreamPool pool;
    pool.run(n, std::function<void(int)>(fn));
}

struct PhaseTimer {   // adds the scope's wall time to one accumulator
    double& acc; std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer() { acc += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

void chk(int status, const char* what)
{
    if (status != MSLAM_OK) throw std::runtime_error(std::string(what) + ": " + mslam_strerror(status));
}
void matTo(const cv::Mat& m, double* out, int n) { for (int i = 0; i < n; i++) out[i] = m.at<double>(i / m.cols, i % m.cols); }
cv::Mat matFrom(const double* v, int r, int c) { cv::Mat m(r, c, cv::CV_64F); for (int i = 0; i < r * c; i++) m.at<double>(i / c, i % c) = v[i]; return m; }

// associateFeatureWithMapPoint / registerNewMapPoint (src/CameraPoseEstimator.cpp:218-243)
void associate(DataManager& data, int mapPointIndex, Frame& frame, int featureIdx)
{
    frame.features.mapPointsIndices[(size_t)featureIdx] = mapPointIndex;
    data.mapPoints[(size_t)mapPointIndex].addObservingFrame(frame.meta.frameID, featureIdx);
}
void registerNewMapPoint(DataManager& data, Point3d pos, Frame& f1, int i1, Frame& f2, int i2)
{
    const int id = (int)data.mapPoints.size();
    data.mapPoints.push_back(MapPoint(pos, id));
    associate(data, id, f1, i1);
    associate(data, id, f2, i2);
}

} // namespace

BatchedPipeline::BatchedPipeline(std::vector<DataManager*> streams, const FeatureExtractor& fe, const CameraPoseEstimator& cpe, int device)
    : dms_(std::move(streams)), fe_(fe), cpe_(cpe), device_(device), ring_(cpe.numBackTraverse + 1)
{
    if (dms_.empty()) throw std::runtime_error("BatchedPipeline: no streams");
    if (ring_ < 2) ring_ = 2;
    lastStatus.assign(dms_.size(), MSLAM_OK);
}

const char* BatchedPipeline::phaseName(int p)
{
    static const char* const names[PH_COUNT] = { "stage_frames", "extract", "features_down", "match", "fmat_prep", "fmat", "associate", "pnp_prep", "pnp", "triangulate" };
    return (p >= 0 && p < PH_COUNT) ? names[p] : "";
}

BatchedPipeline::~BatchedPipeline()
{
    if (runner_) {
        runner_->sync();
        for (int i = 0; i < 4; i++) if (dscratch_[i]) (void)mslam_dev_free(runner_->ctx(), dscratch_[i]);
        if (pinned_) (void)mslam_host_free(runner_->ctx(), pinned_);
        if (arena_) (void)mslam_host_free(runner_->ctx(), arena_);
    }
}

void* BatchedPipeline::scratch(int slot, size_t bytes)
{
    if (bytes > dscratch_bytes_[slot]) {
        if (dscratch_[slot]) chk(mslam_dev_free(runner_->ctx(), dscratch_[slot]), "mslam_dev_free");
        dscratch_[slot] = nullptr; dscratch_bytes_[slot] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        chk(mslam_dev_alloc(runner_->ctx(), &dscratch_[slot], want), "mslam_dev_alloc");
        dscratch_bytes_[slot] = want;
    }
    return dscratch_[slot];
}

// frames[frameIdx] of every stream, rows packed, into the pinned half frameIdx & 1
void BatchedPipeline::stageFrames(int frameIdx)
{
    const int B = streams();
    const size_t row = (size_t)w_ * ch_, fb = row * h_;
    uint8_t* dst = (uint8_t*)pinned_ + (size_t)(frameIdx & 1) * pinned_bytes_;
    for (int b = 0; b < B; b++) {
        const cv::Mat& img = dms_[(size_t)b]->frames.at((size_t)frameIdx).frameBuffer;
        if (img.cols != w_ || img.rows != h_ || img.channels() != ch_) throw std::runtime_error("BatchedPipeline: all streams must share one frame geometry");
    }
    parallelStreams(B, [&](int b) {
        const cv::Mat& img = dms_[(size_t)b]->frames[(size_t)frameIdx].frameBuffer;
        for (int y = 0; y < h_; y++) std::memcpy(dst + (size_t)b * fb + (size_t)y * row, img.data + img.step * (size_t)y, row);
    });
}

// ---- FeatureExtractor::process for all streams (src/FeatureExtractor.cpp:13-31)
void BatchedPipeline::extractAll(int frameIdx)
{
    const int B = streams();
    const cv::Mat& f0 = dms_[0]->frames.at((size_t)frameIdx).frameBuffer;
    if (!runner_) {
        w_ = f0.cols; h_ = f0.rows; ch_ = f0.channels();
        MultiStreamRunner::Config cfg;
        cfg.device = device_; cfg.streams = B; cfg.w = w_; cfg.h = h_; cfg.channels = ch_; cfg.orb = fe_.params;
        cfg.capacitySlack = fe_.capacitySlack; cfg.ring = ring_; cfg.frameSets = 2; cfg.matchSets = ring_ - 1; cfg.pnpOnSecondStream = false;
        runner_.reset(new MultiStreamRunner(cfg));
        pinned_bytes_ = (size_t)w_ * ch_ * h_ * B;
        chk(mslam_host_alloc(runner_->ctx(), &pinned_, 2 * pinned_bytes_), "mslam_host_alloc");
    }
    const int set = frameIdx & 1, slot = slotOf(frameIdx);
    uint8_t* half[2] = { (uint8_t*)pinned_, (uint8_t*)pinned_ + pinned_bytes_ };
    std::vector<int32_t> n, st;
    if (prefetched_ != frameIdx) {
        { PhaseTimer pt(phaseMs[PH_STAGE_FRAMES]); stageFrames(frameIdx); }
        PhaseTimer pt(phaseMs[PH_EXTRACT]);
        runner_->uploadFramesAsync(set, half[set]);
    }
    {
        PhaseTimer pt(phaseMs[PH_EXTRACT]);
        runner_->extract(set, slot);   // waits on the device for a prefetched set
    }
    bool next = prefetchNextFrame;
    for (int b = 0; b < B && next; b++) {
        const std::vector<Frame>& fr = dms_[(size_t)b]->frames;
        next = (size_t)frameIdx + 1 < fr.size() && fr[(size_t)frameIdx + 1].frameBuffer.cols == w_ && fr[(size_t)frameIdx + 1].frameBuffer.rows == h_ &&
               fr[(size_t)frameIdx + 1].frameBuffer.channels() == ch_;
    }
    if (next) {   // host staging and the copy both overlap the extraction just enqueued
        { PhaseTimer pt(phaseMs[PH_STAGE_FRAMES]); stageFrames(frameIdx + 1); }
        runner_->uploadFramesPrefetch(set ^ 1, half[set ^ 1]);
        prefetched_ = frameIdx + 1;
    }
    {
        PhaseTimer pt(phaseMs[PH_EXTRACT]);
        runner_->downloadCounts(slot, n, &st);   // syncs the extract stream
    }
    PhaseTimer pt(phaseMs[PH_FEATURES_DOWN]);
    const int cap = runner_->cap();
    if (slotFeatures_.size() != (size_t)ring_) slotFeatures_.assign((size_t)ring_, 0);
    slotFeatures_[(size_t)slot] = 0;
    for (int b = 0; b < B; b++) slotFeatures_[(size_t)slot] = std::max(slotFeatures_[(size_t)slot], n[(size_t)b]);
    for (int b = 0; b < B; b++) chk(st[(size_t)b], "mslam_orb_extract_batch_dev (per-frame status)");
    const mslam_keypoint* all = runner_->downloadKeypointsPinned(slot);   // one copy for the B streams
    parallelStreams(B, [&](int b) {
        Frame& frame = dms_[(size_t)b]->frames[(size_t)frameIdx];
        const int nb = n[(size_t)b];
        const mslam_keypoint* kps = all + (size_t)b * cap;
        frame.features.positions.reserve((size_t)nb); frame.features.scales.reserve((size_t)nb);
        for (int i = 0; i < nb; i++) {
            frame.features.positions.push_back(Point2d(cv::Point2f(kps[i].x, kps[i].y)));   // Point2f -> Point2d widening
            frame.features.scales.push_back((double)kps[i].size);
        }
        frame.features.mapPointsIndices.resize((size_t)nb, -1);
    });
    for (int b = 0; b < B; b++) {
        Frame& frame = dms_[(size_t)b]->frames[(size_t)frameIdx];
        cv::Mat desc;   // empty unless eager: descriptors(stream, frame) fills it on demand
        if (eagerDescriptors) {
            std::vector<mslam_keypoint> kps;
            desc.create(n[(size_t)b], 32, cv::CV_8U);
            runner_->downloadFeatures(slot, b, n[(size_t)b], kps, desc.data);
        }
        frame.features.descriptors = desc;
    }
}

const cv::Mat& BatchedPipeline::descriptors(int stream, int frameIdx)
{
    Frame& frame = dms_.at((size_t)stream)->frames.at((size_t)frameIdx);
    const int n = (int)frame.features.positions.size();
    if (frame.features.descriptors.rows != n) {
        const int newest = (int)dms_[(size_t)stream]->relative_frame_poses.size() - 1;
        if (!runner_ || frameIdx > newest || frameIdx <= newest - ring_) throw std::runtime_error("BatchedPipeline::descriptors: frame no longer resident on the device");
        cv::Mat d(n, 32, cv::CV_8U);
        std::vector<mslam_keypoint> kps;
        runner_->downloadFeatures(slotOf(frameIdx), stream, n, kps, d.data);
        frame.features.descriptors = d;
    }
    return frame.features.descriptors;
}

// matchFeatures (src/CameraPoseEstimator.cpp:200-213) + computeFundamentalMatrix (:545-586) of slot slotQ against each of
// slotsT, for all streams: N = slotsT.size() x B problems in ONE launch sequence with ONE synchronisation.  Nothing is
// assembled on the host: the matched coordinates are gathered from the resident keypoints (mslam_gather_matches_dev), the
// sample tables are drawn on the device (mslam_sample_tables_dev == geom::sampleTable), and the match lists, inlier masks
// and F come back through one pinned buffer.
void BatchedPipeline::matchFilterMany(int slotQ, const std::vector<int>& slotsT, float ratio, const std::vector<uint64_t>& seeds, Filtered& out)
{
    const int B = streams(), cap = runner_->cap(), H = cpe_.fmatHypotheses, nb = (int)slotsT.size(), N = nb * B;
    const int kf = cpe_.fmatSamplePoints == 7 ? 7 : 8;   // points per F sample (Stages.h)
    mslam_ctx* ctx = runner_->ctx();
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nl = (size_t)N * cap;
    if (!scratch_sized_) {   // size the buffers for the full back-traverse depth once: no reallocation while the ring fills up
        const size_t Nmax = (size_t)(ring_ - 1) * B, nlmax = Nmax * cap;
        (void)scratch(0, 2 * up256(sizeof(double) * nlmax * 2) + up256(sizeof(int32_t) * Nmax * H * 8) + up256(sizeof(int32_t) * Nmax) + up256(sizeof(uint64_t) * Nmax));
        (void)scratch(1, up256(sizeof(double) * 9 * Nmax) + up256(nlmax) + up256(sizeof(int32_t) * 2 * Nmax));
        scratch_sized_ = true;
    }
    // device scratch: x1, x2, tables, M, seeds | F, status, best, ninl
    const size_t o_x2 = up256(sizeof(double) * nl * 2), o_S = 2 * o_x2, o_M = o_S + up256(sizeof(int32_t) * (size_t)N * H * 8), o_sd = o_M + up256(sizeof(int32_t) * N);
    uint8_t* in = (uint8_t*)scratch(0, o_sd + up256(sizeof(uint64_t) * N));
    double* dx1 = (double*)in; double* dx2 = (double*)(in + o_x2);
    int32_t* dS = (int32_t*)(in + o_S); int32_t* dM = (int32_t*)(in + o_M); uint64_t* dseed = (uint64_t*)(in + o_sd);
    const size_t o_st = up256(sizeof(double) * 9 * N), o_best = o_st + up256(nl);
    uint8_t* outb = (uint8_t*)scratch(1, o_best + up256(sizeof(int32_t) * 2 * N));
    double* dF = (double*)outb; uint8_t* dst = outb + o_st; int32_t* dbest = (int32_t*)(outb + o_best); int32_t* dninl = dbest + N;
    // pinned arena: seeds (up) | query, train, dist, count, status, F, best (down)
    const size_t h_q = up256(sizeof(uint64_t) * N), h_t = h_q + up256(4 * nl), h_d = h_t + up256(4 * nl), h_c = h_d + up256(4 * nl), h_st = h_c + up256(4 * (size_t)N),
                 h_F = h_st + up256(nl), h_b = h_F + up256(sizeof(double) * 9 * N), h_end = h_b + up256(4 * (size_t)N);
    if (h_end > arena_bytes_) {
        const size_t Nmax = (size_t)(ring_ - 1) * B;
        const size_t want = std::max(h_end, (h_end / (size_t)N + 1) * Nmax);
        if (arena_) chk(mslam_host_free(ctx, arena_), "mslam_host_free");
        arena_ = nullptr; arena_bytes_ = 0;
        chk(mslam_host_alloc(ctx, &arena_, want + want / 8), "mslam_host_alloc");
        arena_bytes_ = want + want / 8;
    }
    uint8_t* hp = (uint8_t*)arena_;
    {
        PhaseTimer pt(phaseMs[PH_MATCH]);
        std::memcpy(hp, seeds.data(), sizeof(uint64_t) * N);
        chk(mslam_memcpy_h2d_async(ctx, dseed, hp, sizeof(uint64_t) * N), "mslam_memcpy_h2d_async");
        for (int k = 0; k < nb; k++) runner_->match(slotQ, slotsT[(size_t)k], ratio, k);   // enqueued back to back
        for (int k = 0; k < nb; k++) {
            const size_t o = (size_t)k * B * cap;
            chk(mslam_gather_matches_dev(ctx, runner_->kpsDev(slotQ), runner_->kpsDev(slotsT[(size_t)k]), cap, runner_->matchQueryDev() + o, runner_->matchTrainDev() + o,
                                         runner_->matchCountDev() + (size_t)k * B, B, kf, dx1 + o * 2, dx2 + o * 2, dM + (size_t)k * B), "mslam_gather_matches_dev");
        }
        chk(mslam_sample_tables_dev(ctx, dM, dseed, H, kf, N, dS), "mslam_sample_tables_dev");
    }
    PhaseTimer pt(phaseMs[PH_FMAT]);
    if (kf == 7) chk(mslam_fmat_ransac7_batch_dev(ctx, dx1, dx2, dM, cap, dS, H, N, &cpe_.fmat, 0, dF, dst, nullptr, dbest, dninl), "mslam_fmat_ransac7_batch_dev");
    else chk(mslam_fmat_ransac8_batch_dev(ctx, dx1, dx2, dM, cap, dS, H, N, &cpe_.fmat, dF, dst, nullptr, dbest, dninl), "mslam_fmat_ransac8_batch_dev");
    // a match list holds at most as many entries as the query frame has features: only those columns of the cap-wide tables travel
    const size_t used = (size_t)std::min(cap, std::max(slotFeatures_[(size_t)slotQ], 1));
    chk(mslam_memcpy_d2h_2d_async(ctx, hp + h_q, 4 * (size_t)cap, runner_->matchQueryDev(), 4 * (size_t)cap, 4 * used, (size_t)N), "mslam_memcpy_d2h_2d_async");
    chk(mslam_memcpy_d2h_2d_async(ctx, hp + h_t, 4 * (size_t)cap, runner_->matchTrainDev(), 4 * (size_t)cap, 4 * used, (size_t)N), "mslam_memcpy_d2h_2d_async");
    chk(mslam_memcpy_d2h_2d_async(ctx, hp + h_d, 4 * (size_t)cap, runner_->matchDistDev(), 4 * (size_t)cap, 4 * used, (size_t)N), "mslam_memcpy_d2h_2d_async");
    chk(mslam_memcpy_d2h_async(ctx, hp + h_c, runner_->matchCountDev(), 4 * (size_t)N), "mslam_memcpy_d2h_async");
    chk(mslam_memcpy_d2h_2d_async(ctx, hp + h_st, (size_t)cap, dst, (size_t)cap, used, (size_t)N), "mslam_memcpy_d2h_2d_async");
    chk(mslam_memcpy_d2h_async(ctx, hp + h_F, dF, sizeof(double) * 9 * N), "mslam_memcpy_d2h_async");
    chk(mslam_memcpy_d2h_async(ctx, hp + h_b, dbest, 4 * (size_t)N), "mslam_memcpy_d2h_async");
    chk(mslam_sync(ctx), "mslam_sync");   // the only synchronisation of the stage
    out.cap = cap;
    out.query = (const int32_t*)(hp + h_q); out.train = (const int32_t*)(hp + h_t); out.dist = (const float*)(hp + h_d);
    out.count = (const int32_t*)(hp + h_c); out.status = hp + h_st; out.F = (const double*)(hp + h_F);
    const int32_t* best = (const int32_t*)(hp + h_b);
    out.rc.assign((size_t)N, MSLAM_OK);
    for (int q = 0; q < N; q++) {
        if (out.count[q] < kf) out.rc[(size_t)q] = MSLAM_ERR_TOO_FEW;          // status of such a set is not meaningful: inlier() says 0
        else if (best[q] < 0) out.rc[(size_t)q] = MSLAM_ERR_DEGENERATE;
    }
}

// initialPoseEstimation (src/CameraPoseEstimator.cpp:264-376)
void BatchedPipeline::poseInitial(int frameIdx)
{
    const int B = streams(), cap = runner_->cap();
    Filtered ms;
    std::vector<uint64_t> seeds((size_t)B, cpe_.seed + 1000ull * (uint64_t)frameIdx);
    matchFilterMany(slotOf(frameIdx - 1), { slotOf(frameIdx) }, cpe_.ratioInitial, seeds, ms);   // query = previous frame, train = current (:280)
    const std::vector<int>& rc = ms.rc;
    const double* F = ms.F;
    // per stream: inlier coordinates and the four (R, t) candidates in the reference's order (R1|t1, R1|t2, R2|t1, R2|t2);
    // ONE batched triangulation for all 4 B candidate groups, then the vote per stream (:337-354)
    struct Init { bool ok = false; std::vector<double> x1, x2; double K1[9], K2[9], cand[4][12]; };
    std::vector<Init> in((size_t)B);
    const double I34[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    for (int b = 0; b < B; b++) {
        DataManager& data = *dms_[(size_t)b];
        Frame& cur = data.frames[(size_t)frameIdx];
        Frame& pre = data.frames[(size_t)frameIdx - 1];
        cur.Rt = pre.Rt.clone(); // kept if the geometry below cannot be estimated
        if (rc[(size_t)b] != MSLAM_OK) { lastStatus[(size_t)b] = rc[(size_t)b]; continue; }
        Init& s0 = in[(size_t)b];
        const int m = ms.count[(size_t)b];
        for (int i = 0; i < m; i++)
            if (ms.inlier(b, i)) {
                const Point2d& a = pre.features.positions[(size_t)ms.query[(size_t)b * cap + i]];
                const Point2d& c = cur.features.positions[(size_t)ms.train[(size_t)b * cap + i]];
                s0.x1.push_back(a.x); s0.x1.push_back(a.y); s0.x2.push_back(c.x); s0.x2.push_back(c.y);
            }
        double E[9], R[2][9], t[2][3];
        matTo(pre.K, s0.K1, 9); matTo(cur.K, s0.K2, 9);
        geom::essentialFromF(&F[(size_t)b * 9], s0.K1, s0.K2, E);
        geom::extractRTfromE(E, R, t);
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) s0.cand[c][i * 4 + j] = R[c / 2][i * 3 + j]; s0.cand[c][i * 4 + 3] = t[c % 2][i]; }
        s0.ok = true;
    }
    std::vector<double> gx1, gx2, gRt1, gRt2, gK1, gK2;
    std::vector<int32_t> ofs(1, 0), gstream;
    for (int b = 0; b < B; b++) {
        if (!in[(size_t)b].ok) continue;
        const Init& s0 = in[(size_t)b];
        for (int c = 0; c < 4; c++) {
            gx1.insert(gx1.end(), s0.x1.begin(), s0.x1.end()); gx2.insert(gx2.end(), s0.x2.begin(), s0.x2.end());
            gRt1.insert(gRt1.end(), I34, I34 + 12); gRt2.insert(gRt2.end(), s0.cand[c], s0.cand[c] + 12);
            gK1.insert(gK1.end(), s0.K1, s0.K1 + 9); gK2.insert(gK2.end(), s0.K2, s0.K2 + 9);
            ofs.push_back(ofs.back() + (int32_t)(s0.x1.size() / 2));
            gstream.push_back(b);
        }
    }
    const int G = (int)gstream.size();
    std::vector<double> gX((size_t)ofs.back() * 3);
    std::vector<int32_t> gfront((size_t)std::max(G, 1), 0);
    if (G > 0 && ofs.back() > 0)
        chk(mslam_triangulate_batch(runner_->ctx(), gx1.data(), gx2.data(), ofs.data(), G, gRt1.data(), gRt2.data(), gK1.data(), gK2.data(), gX.data(), nullptr,
                                    gfront.data()), "mslam_triangulate_batch");
    for (int g0 = 0; g0 < G; g0 += 4) {
        const int b = gstream[(size_t)g0];
        DataManager& data = *dms_[(size_t)b];
        Frame& cur = data.frames[(size_t)frameIdx];
        Frame& pre = data.frames[(size_t)frameIdx - 1];
        const Init& s0 = in[(size_t)b];
        int bestIdx = -1, maxCount = -1;
        for (int c = 0; c < 4; c++)
            if (maxCount < gfront[(size_t)g0 + c]) { maxCount = gfront[(size_t)g0 + c]; bestIdx = c; }
        if (bestIdx < 0) { lastStatus[(size_t)b] = MSLAM_ERR_DEGENERATE; continue; }
        const double* bestX = gX.data() + (size_t)ofs[(size_t)g0 + bestIdx] * 3;
        double preRt[12], out[12];
        matTo(pre.Rt, preRt, 12);
        geom::concatenateRts(preRt, s0.cand[bestIdx], out);
        cur.Rt = matFrom(out, 3, 4);
        int count = 0;
        const int m = ms.count[(size_t)b];
        for (int i = 0; i < m; i++)
            if (ms.inlier(b, i) == 1) {
                registerNewMapPoint(data, Point3d(bestX[3 * (size_t)count], bestX[3 * (size_t)count + 1], bestX[3 * (size_t)count + 2]), pre,
                                    ms.query[(size_t)b * cap + i], cur, ms.train[(size_t)b * cap + i]);
                count++;
            }
    }
}

// pnpPoseEstimation (src/CameraPoseEstimator.cpp:384-515)
void BatchedPipeline::posePnp(int frameIdx)
{
    const int B = streams(), cap = runner_->cap();
    struct PerStream {
        std::vector<bool> matched; std::vector<Point3d> mapPoints; std::vector<Point2d> imagePoints;
        std::vector<std::vector<cv::DMatch>> cachedMatches; int count = 0, numCur = 0; bool done = false;
    };
    std::vector<PerStream> ps((size_t)B);
    for (int b = 0; b < B; b++) {
        ps[(size_t)b].numCur = (int)dms_[(size_t)b]->frames[(size_t)frameIdx].features.positions.size();
        ps[(size_t)b].matched.assign((size_t)ps[(size_t)b].numCur, false);
    }
    // The match and the epipolar filter of the current frame against each back frame do not depend on the association
    // state, so all of them go out in one launch sequence (N = back frames x streams problems); the association below
    // then walks the back frames in the reference's order per stream and stops where the reference's loop stops (:456).
    std::vector<int> backs, backSlots;
    for (int i = frameIdx - 1; i >= 0 && i >= frameIdx - cpe_.numBackTraverse; i--) { backs.push_back(i); backSlots.push_back(slotOf(i)); }
    const int nb = (int)backs.size(), N = nb * B;
    std::vector<uint64_t> seeds((size_t)N);
    for (int k = 0; k < nb; k++)
        for (int b = 0; b < B; b++) seeds[(size_t)k * B + b] = cpe_.seed + 1000ull * (uint64_t)frameIdx + (uint64_t)(frameIdx - backs[(size_t)k]);
    Filtered ms;   // query = current, train = previous frame i (:405-409); epipolar filter (FILTERING_WITH_F, :412-427): F itself is discarded
    matchFilterMany(slotOf(frameIdx), backSlots, cpe_.ratioTracking, seeds, ms);
    const std::vector<int>& rc = ms.rc;
    {
        PhaseTimer pt(phaseMs[PH_ASSOCIATE]);
        parallelStreams(B, [&](int b) {
            PerStream& s = ps[(size_t)b];
            DataManager& data = *dms_[(size_t)b];
            Frame& cur = data.frames[(size_t)frameIdx];
            for (int k = 0; k < nb && !s.done; k++) {
                const size_t q = (size_t)k * B + b;
                Features& preF = data.frames[(size_t)backs[(size_t)k]].features;
                if (rc[q] != MSLAM_OK) lastStatus[(size_t)b] = rc[q];
                std::vector<cv::DMatch> raw;
                for (int j = 0; j < ms.count[q]; j++)
                    if (ms.inlier((int)q, j)) {
                        cv::DMatch m; m.queryIdx = ms.query[q * cap + j]; m.trainIdx = ms.train[q * cap + j]; m.distance = ms.dist[q * cap + j];
                        raw.push_back(m);
                    }
                s.cachedMatches.push_back(raw);
                for (size_t j = 0; j < raw.size(); j++) {
                    const int cf = raw[j].queryIdx, pf = raw[j].trainIdx;
                    if (preF.mapPointsIndices[(size_t)pf] != -1 && !s.matched[(size_t)cf]) {
                        s.count++;
                        s.matched[(size_t)cf] = true;
                        const int mp = preF.mapPointsIndices[(size_t)pf];
                        associate(data, mp, cur, cf);
                        s.mapPoints.push_back(data.mapPoints[(size_t)mp].worldPosition);
                        s.imagePoints.push_back(cur.features.positions[(size_t)cf]);
                        if (s.count == s.numCur) break;
                    }
                }
                if (s.count == s.numCur) s.done = true;
            }
        });
    }
    // ---- solvePnPRansac for all streams (:472)
    const int H = cpe_.pnpHypotheses;
    int capP = 4;
    for (int b = 0; b < B; b++) capP = std::max(capP, (int)ps[(size_t)b].mapPoints.size());
    std::vector<double> X((size_t)B * capP * 3, 0.0), x((size_t)B * capP * 2, 0.0), K((size_t)B * 9), D((size_t)B * 8);
    std::vector<int32_t> P((size_t)B, 0);
    std::vector<uint64_t> pnpSeeds((size_t)B, cpe_.seed + 1000ull * (uint64_t)frameIdx + 99);
    const auto tprep = std::chrono::steady_clock::now();
    parallelStreams(B, [&](int b) {
        const PerStream& s = ps[(size_t)b];
        Frame& cur = dms_[(size_t)b]->frames[(size_t)frameIdx];
        cur.Rt = dms_[(size_t)b]->frames[(size_t)frameIdx - 1].Rt.clone(); // kept if PnP cannot be solved
        matTo(cur.K, &K[(size_t)b * 9], 9);
        std::memcpy(&D[(size_t)b * 8], cpe_.distCoeff, sizeof(double) * 8);
        const int p = (int)s.mapPoints.size();
        if (p < 4) { lastStatus[(size_t)b] = MSLAM_ERR_TOO_FEW; return; }
        P[(size_t)b] = p;
        for (int i = 0; i < p; i++) {
            X[((size_t)b * capP + i) * 3] = s.mapPoints[(size_t)i].x; X[((size_t)b * capP + i) * 3 + 1] = s.mapPoints[(size_t)i].y; X[((size_t)b * capP + i) * 3 + 2] = s.mapPoints[(size_t)i].z;
            x[((size_t)b * capP + i) * 2] = s.imagePoints[(size_t)i].x; x[((size_t)b * capP + i) * 2 + 1] = s.imagePoints[(size_t)i].y;
        }
    });
    phaseMs[PH_PNP_PREP] += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tprep).count();
    std::vector<double> Rt;
    std::vector<int32_t> best, ninl;
    {
        PhaseTimer pt(phaseMs[PH_PNP]);
        runner_->setPnpProblems(capP, H, X.data(), x.data(), P.data(), K.data(), D.data(), nullptr, cpe_.pnp, pnpSeeds.data());   // tables drawn on the device
        runner_->pnp();
        runner_->downloadPnp(Rt, best, ninl);
    }
    for (int b = 0; b < B; b++) {
        if (P[(size_t)b] < 4) continue;
        if (best[(size_t)b] >= 0 && ninl[(size_t)b] >= cpe_.pnp.min_inliers) dms_[(size_t)b]->frames[(size_t)frameIdx].Rt = matFrom(&Rt[(size_t)b * 12], 3, 4);
        else lastStatus[(size_t)b] = MSLAM_ERR_DEGENERATE;
    }
    // ---- triangulate the matches not yet tied to a map point, against each previous frame (:488-512).  Which pairs get a new
    // point depends on the association state only (it advances from one back frame to the next), not on the 3D positions:
    // each stream registers its new points first (position pending), then ONE batched triangulation over all
    // (stream, back frame) groups fills the positions in.
    PhaseTimer ptTri(phaseMs[PH_TRIANGULATE]);
    struct Group { int back; size_t firstPoint, n; };
    struct NewPoints { std::vector<Group> groups; std::vector<double> x1, x2; };
    std::vector<NewPoints> np((size_t)B);
    parallelStreams(B, [&](int b) {
        DataManager& data = *dms_[(size_t)b];
        Frame& cur = data.frames[(size_t)frameIdx];
        Features& curF = cur.features;
        NewPoints& out = np[(size_t)b];
        for (size_t mi = 0; mi < ps[(size_t)b].cachedMatches.size(); mi++) {   // fewer than nb entries: the stream left the loop earlier (:456)
            const int i = frameIdx - 1 - (int)mi;
            Frame& pre = data.frames[(size_t)i];
            Features& preF = pre.features;
            Group g{ i, data.mapPoints.size(), 0 };
            for (const cv::DMatch& m : ps[(size_t)b].cachedMatches[mi]) {
                const int cf = m.queryIdx, pf = m.trainIdx;
                if (preF.mapPointsIndices[(size_t)pf] == -1 && curF.mapPointsIndices[(size_t)cf] == -1) {   // both ends unassociated (:493)
                    registerNewMapPoint(data, Point3d(0, 0, 0), pre, pf, cur, cf);
                    out.x1.push_back(preF.positions[(size_t)pf].x); out.x1.push_back(preF.positions[(size_t)pf].y);
                    out.x2.push_back(curF.positions[(size_t)cf].x); out.x2.push_back(curF.positions[(size_t)cf].y);
                    g.n++;
                }
            }
            if (g.n) out.groups.push_back(g);
        }
    });
    std::vector<double> gx1, gx2, gRt1, gRt2, gK1, gK2;
    std::vector<int32_t> ofs(1, 0);
    for (int b = 0; b < B; b++) {
        const NewPoints& o = np[(size_t)b];
        if (o.groups.empty()) continue;
        DataManager& data = *dms_[(size_t)b];
        Frame& cur = data.frames[(size_t)frameIdx];
        gx1.insert(gx1.end(), o.x1.begin(), o.x1.end()); gx2.insert(gx2.end(), o.x2.begin(), o.x2.end());
        for (const Group& g : o.groups) {
            Frame& pre = data.frames[(size_t)g.back];
            double m12[12], k9[9];
            matTo(pre.Rt, m12, 12); gRt1.insert(gRt1.end(), m12, m12 + 12);
            matTo(cur.Rt, m12, 12); gRt2.insert(gRt2.end(), m12, m12 + 12);
            matTo(pre.K, k9, 9); gK1.insert(gK1.end(), k9, k9 + 9);
            matTo(cur.K, k9, 9); gK2.insert(gK2.end(), k9, k9 + 9);
            ofs.push_back(ofs.back() + (int32_t)g.n);
        }
    }
    const int G = (int)ofs.size() - 1;
    if (G == 0) return;
    std::vector<double> gX((size_t)ofs.back() * 3);
    chk(mslam_triangulate_batch(runner_->ctx(), gx1.data(), gx2.data(), ofs.data(), G, gRt1.data(), gRt2.data(), gK1.data(), gK2.data(), gX.data(), nullptr, nullptr),
        "mslam_triangulate_batch");
    size_t g0 = 0;
    for (int b = 0; b < B; b++)
        for (const Group& g : np[(size_t)b].groups) {
            const double* Xn = gX.data() + (size_t)ofs[g0++] * 3;
            DataManager& data = *dms_[(size_t)b];
            for (size_t k = 0; k < g.n; k++) data.mapPoints[g.firstPoint + k].worldPosition = Point3d(Xn[3 * k], Xn[3 * k + 1], Xn[3 * k + 2]);
        }
}

void BatchedPipeline::process(int frameIdx)
{
    const int B = streams();
    extractAll(frameIdx);
    std::fill(lastStatus.begin(), lastStatus.end(), (int)MSLAM_OK);
    if (frameIdx == 0) { for (int b = 0; b < B; b++) dms_[(size_t)b]->frames[0].Rt = cv::Mat::eye(3, 4, cv::CV_64F); }
    else if (frameIdx == 1) poseInitial(frameIdx);
    else posePnp(frameIdx);
    for (int b = 0; b < B; b++) dms_[(size_t)b]->relative_frame_poses.push_back(dms_[(size_t)b]->frames[(size_t)frameIdx].Rt);   // (:528)
}
