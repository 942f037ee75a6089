#include "BatchedPipeline.h"
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <thread>
#include "Geometry.h"

namespace {

// run fn(b) for b in [0, n) on the host cores: the per-stream preparation between two batched launches (sample tables,
// coordinate gathers) is independent per stream
template <typename F> void parallelStreams(int n, F fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<unsigned>(std::min<unsigned>(hw ? hw : 1u, 16u), (unsigned)n);
    if (nt <= 1) { for (int b = 0; b < n; b++) fn(b); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&]() { for (int b = next++; b < n; b = next++) fn(b); });
    for (std::thread& t : th) t.join();
}

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

BatchedPipeline::~BatchedPipeline()
{
    if (runner_) {
        runner_->sync();
        for (int i = 0; i < 4; i++) if (dscratch_[i]) (void)mslam_dev_free(runner_->ctx(), dscratch_[i]);
        if (pinned_) (void)mslam_host_free(runner_->ctx(), pinned_);
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

// ---- FeatureExtractor::process for all streams (src/FeatureExtractor.cpp:13-31)
void BatchedPipeline::extractAll(int frameIdx)
{
    const int B = streams();
    const cv::Mat& f0 = dms_[0]->frames.at((size_t)frameIdx).frameBuffer;
    if (!runner_) {
        w_ = f0.cols; h_ = f0.rows; ch_ = f0.channels();
        MultiStreamRunner::Config cfg;
        cfg.device = device_; cfg.streams = B; cfg.w = w_; cfg.h = h_; cfg.channels = ch_; cfg.orb = fe_.params;
        cfg.capacitySlack = fe_.capacitySlack; cfg.ring = ring_; cfg.frameSets = 2; cfg.pnpOnSecondStream = false;
        runner_.reset(new MultiStreamRunner(cfg));
        pinned_bytes_ = (size_t)w_ * ch_ * h_ * B;
        chk(mslam_host_alloc(runner_->ctx(), &pinned_, pinned_bytes_), "mslam_host_alloc");
    }
    const size_t row = (size_t)w_ * ch_, fb = row * h_;
    for (int b = 0; b < B; b++) {
        const cv::Mat& img = dms_[(size_t)b]->frames.at((size_t)frameIdx).frameBuffer;
        if (img.cols != w_ || img.rows != h_ || img.channels() != ch_) throw std::runtime_error("BatchedPipeline: all streams must share one frame geometry");
        for (int y = 0; y < h_; y++) std::memcpy((uint8_t*)pinned_ + (size_t)b * fb + (size_t)y * row, img.data + img.step * (size_t)y, row);
    }
    const int set = frameIdx & 1, slot = slotOf(frameIdx);
    runner_->uploadFramesAsync(set, (const uint8_t*)pinned_);
    runner_->extract(set, slot);
    std::vector<int32_t> n, st;
    runner_->downloadCounts(slot, n, &st);   // syncs: the pinned buffer may be refilled from here on
    const int cap = runner_->cap();
    std::vector<mslam_keypoint> kps;
    for (int b = 0; b < B; b++) {
        chk(st[(size_t)b], "mslam_orb_extract_batch_dev (per-frame status)");
        Frame& frame = dms_[(size_t)b]->frames[(size_t)frameIdx];
        const int nb = n[(size_t)b];
        cv::Mat desc;
        if (eagerDescriptors) { desc.create(nb, 32, cv::CV_8U); runner_->downloadFeatures(slot, b, nb, kps, desc.data); }
        else runner_->downloadFeatures(slot, b, nb, kps, nullptr);
        frame.features.descriptors = desc;   // empty unless eager: descriptors(stream, frame) fills it on demand
        for (int i = 0; i < nb; i++) {
            frame.features.positions.push_back(Point2d(cv::Point2f(kps[(size_t)i].x, kps[(size_t)i].y)));   // Point2f -> Point2d widening
            frame.features.scales.push_back((double)kps[(size_t)i].size);
        }
        frame.features.mapPointsIndices.resize((size_t)nb, -1);
        (void)cap;
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

void BatchedPipeline::matchAll(int slotQ, int slotT, float ratio, MatchSet& out)
{
    runner_->match(slotQ, slotT, ratio);
    runner_->downloadMatches(out.count, out.query, out.train, out.dist);
}

// computeFundamentalMatrix (src/CameraPoseEstimator.cpp:545-586) for the B match sets in one launch sequence
void BatchedPipeline::fmatAll(const std::vector<const Frame*>& fq, const std::vector<const Frame*>& ft, const MatchSet& ms,
                              const std::vector<uint64_t>& seeds, std::vector<double>& F, std::vector<std::vector<uint8_t>>& status, std::vector<int>& rc)
{
    const int B = streams(), cap = runner_->cap(), H = cpe_.fmatHypotheses;
    int capM = 8;
    for (int b = 0; b < B; b++) capM = std::max(capM, ms.count[(size_t)b]);
    std::vector<double> x1((size_t)B * capM * 2, 0.0), x2((size_t)B * capM * 2, 0.0);
    std::vector<int32_t> M((size_t)B, 0), S((size_t)B * H * 8, 0);
    rc.assign((size_t)B, MSLAM_OK);
    parallelStreams(B, [&](int b) {
        const int m = ms.count[(size_t)b];
        if (m < 8) { rc[(size_t)b] = MSLAM_ERR_TOO_FEW; return; }   // the kernels see M = 0 for this stream and skip it
        M[(size_t)b] = m;
        for (int i = 0; i < m; i++) {
            const Point2d& a = fq[(size_t)b]->features.positions[(size_t)ms.query[(size_t)b * cap + i]];
            const Point2d& c = ft[(size_t)b]->features.positions[(size_t)ms.train[(size_t)b * cap + i]];
            x1[((size_t)b * capM + i) * 2] = a.x; x1[((size_t)b * capM + i) * 2 + 1] = a.y;
            x2[((size_t)b * capM + i) * 2] = c.x; x2[((size_t)b * capM + i) * 2 + 1] = c.y;
        }
        const std::vector<int32_t> t = geom::sampleTable(m, H, 8, seeds[(size_t)b]);
        std::memcpy(&S[(size_t)b * H * 8], t.data(), sizeof(int32_t) * t.size());
    });
    mslam_ctx* ctx = runner_->ctx();
    const size_t nx = sizeof(double) * (size_t)B * capM * 2, nS = sizeof(int32_t) * (size_t)B * H * 8;
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint8_t* in = (uint8_t*)scratch(0, up256(nx) * 2 + up256(nS) + 256 + 256 * (size_t)B);
    double* dx1 = (double*)in; double* dx2 = (double*)(in + up256(nx));
    int32_t* dS = (int32_t*)(in + 2 * up256(nx)); int32_t* dM = (int32_t*)(in + 2 * up256(nx) + up256(nS));
    uint8_t* outb = (uint8_t*)scratch(1, up256(sizeof(double) * 9 * B) + up256((size_t)B * capM) + 1024 * (size_t)B + 4096);
    double* dF = (double*)outb; uint8_t* dst = outb + up256(sizeof(double) * 9 * B);
    int32_t* dbest = (int32_t*)(dst + up256((size_t)B * capM)); int32_t* dninl = dbest + B;
    chk(mslam_memcpy_h2d(ctx, dx1, x1.data(), nx), "mslam_memcpy_h2d");
    chk(mslam_memcpy_h2d(ctx, dx2, x2.data(), nx), "mslam_memcpy_h2d");
    chk(mslam_memcpy_h2d(ctx, dS, S.data(), nS), "mslam_memcpy_h2d");
    chk(mslam_memcpy_h2d(ctx, dM, M.data(), sizeof(int32_t) * B), "mslam_memcpy_h2d");
    chk(mslam_fmat_ransac8_batch_dev(ctx, dx1, dx2, dM, capM, dS, H, B, &cpe_.fmat, dF, dst, nullptr, dbest, dninl), "mslam_fmat_ransac8_batch_dev");
    F.assign((size_t)B * 9, 0.0);
    std::vector<uint8_t> st((size_t)B * capM);
    std::vector<int32_t> best((size_t)B);
    chk(mslam_memcpy_d2h_async(ctx, F.data(), dF, sizeof(double) * 9 * B), "mslam_memcpy_d2h_async");
    chk(mslam_memcpy_d2h_async(ctx, st.data(), dst, (size_t)B * capM), "mslam_memcpy_d2h_async");
    chk(mslam_memcpy_d2h(ctx, best.data(), dbest, sizeof(int32_t) * B), "mslam_memcpy_d2h");
    status.assign((size_t)B, {});
    for (int b = 0; b < B; b++) {
        const int m = ms.count[(size_t)b];
        status[(size_t)b].assign((size_t)m, 0);
        if (rc[(size_t)b] != MSLAM_OK) continue;
        if (best[(size_t)b] < 0) { rc[(size_t)b] = MSLAM_ERR_DEGENERATE; continue; }
        std::memcpy(status[(size_t)b].data(), &st[(size_t)b * capM], (size_t)m);
    }
}

// initialPoseEstimation (src/CameraPoseEstimator.cpp:264-376)
void BatchedPipeline::poseInitial(int frameIdx)
{
    const int B = streams(), cap = runner_->cap();
    MatchSet ms;
    matchAll(slotOf(frameIdx - 1), slotOf(frameIdx), cpe_.ratioInitial, ms);   // query = previous frame, train = current (:280)
    std::vector<const Frame*> fq((size_t)B), ft((size_t)B);
    std::vector<uint64_t> seeds((size_t)B, cpe_.seed + 1000ull * (uint64_t)frameIdx);
    for (int b = 0; b < B; b++) { fq[(size_t)b] = &dms_[(size_t)b]->frames[(size_t)frameIdx - 1]; ft[(size_t)b] = &dms_[(size_t)b]->frames[(size_t)frameIdx]; }
    std::vector<double> F;
    std::vector<std::vector<uint8_t>> status;
    std::vector<int> rc;
    fmatAll(fq, ft, ms, seeds, F, status, rc);
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
            if (status[(size_t)b][(size_t)i]) {
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
            if (status[(size_t)b][(size_t)i] == 1) {
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
    for (int i = frameIdx - 1; i >= 0 && i >= frameIdx - cpe_.numBackTraverse; i--) {
        MatchSet ms;
        matchAll(slotOf(frameIdx), slotOf(i), cpe_.ratioTracking, ms);   // query = current, train = previous frame i (:405-409)
        std::vector<const Frame*> fq((size_t)B), ft((size_t)B);
        std::vector<uint64_t> seeds((size_t)B, cpe_.seed + 1000ull * (uint64_t)frameIdx + (uint64_t)(frameIdx - i));
        for (int b = 0; b < B; b++) {
            fq[(size_t)b] = &dms_[(size_t)b]->frames[(size_t)frameIdx]; ft[(size_t)b] = &dms_[(size_t)b]->frames[(size_t)i];
            if (ps[(size_t)b].done) ms.count[(size_t)b] = 0;   // this stream left the back-traverse loop already (:456)
        }
        std::vector<double> F;
        std::vector<std::vector<uint8_t>> status;
        std::vector<int> rc;
        fmatAll(fq, ft, ms, seeds, F, status, rc);   // epipolar filter (FILTERING_WITH_F, :412-427): F itself is discarded
        for (int b = 0; b < B; b++) {
            PerStream& s = ps[(size_t)b];
            if (s.done) continue;
            DataManager& data = *dms_[(size_t)b];
            Frame& cur = data.frames[(size_t)frameIdx];
            Features& preF = data.frames[(size_t)i].features;
            if (rc[(size_t)b] != MSLAM_OK) lastStatus[(size_t)b] = rc[(size_t)b];
            std::vector<cv::DMatch> raw;
            for (int j = 0; j < ms.count[(size_t)b]; j++)
                if (status[(size_t)b][(size_t)j]) {
                    cv::DMatch m; m.queryIdx = ms.query[(size_t)b * cap + j]; m.trainIdx = ms.train[(size_t)b * cap + j]; m.distance = ms.dist[(size_t)b * cap + j];
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
    }
    // ---- solvePnPRansac for all streams (:472)
    const int H = cpe_.pnpHypotheses;
    int capP = 4;
    for (int b = 0; b < B; b++) capP = std::max(capP, (int)ps[(size_t)b].mapPoints.size());
    std::vector<double> X((size_t)B * capP * 3, 0.0), x((size_t)B * capP * 2, 0.0), K((size_t)B * 9), D((size_t)B * 8);
    std::vector<int32_t> P((size_t)B, 0), S((size_t)B * H * 4, 0);
    for (int b = 0; b < B; b++) {
        const PerStream& s = ps[(size_t)b];
        Frame& cur = dms_[(size_t)b]->frames[(size_t)frameIdx];
        cur.Rt = dms_[(size_t)b]->frames[(size_t)frameIdx - 1].Rt.clone(); // kept if PnP cannot be solved
        matTo(cur.K, &K[(size_t)b * 9], 9);
        std::memcpy(&D[(size_t)b * 8], cpe_.distCoeff, sizeof(double) * 8);
        const int p = (int)s.mapPoints.size();
        if (p < 4) { lastStatus[(size_t)b] = MSLAM_ERR_TOO_FEW; continue; }
        P[(size_t)b] = p;
        for (int i = 0; i < p; i++) {
            X[((size_t)b * capP + i) * 3] = s.mapPoints[(size_t)i].x; X[((size_t)b * capP + i) * 3 + 1] = s.mapPoints[(size_t)i].y; X[((size_t)b * capP + i) * 3 + 2] = s.mapPoints[(size_t)i].z;
            x[((size_t)b * capP + i) * 2] = s.imagePoints[(size_t)i].x; x[((size_t)b * capP + i) * 2 + 1] = s.imagePoints[(size_t)i].y;
        }
        const std::vector<int32_t> t = geom::sampleTable(p, H, 4, cpe_.seed + 1000ull * (uint64_t)frameIdx + 99);
        std::memcpy(&S[(size_t)b * H * 4], t.data(), sizeof(int32_t) * t.size());
    }
    runner_->setPnpProblems(capP, H, X.data(), x.data(), P.data(), K.data(), D.data(), S.data(), cpe_.pnp);
    runner_->pnp();
    std::vector<double> Rt;
    std::vector<int32_t> best, ninl;
    runner_->downloadPnp(Rt, best, ninl);
    for (int b = 0; b < B; b++) {
        if (P[(size_t)b] < 4) continue;
        if (best[(size_t)b] >= 0 && ninl[(size_t)b] >= cpe_.pnp.min_inliers) dms_[(size_t)b]->frames[(size_t)frameIdx].Rt = matFrom(&Rt[(size_t)b * 12], 3, 4);
        else lastStatus[(size_t)b] = MSLAM_ERR_DEGENERATE;
    }
    // ---- triangulate the matches not yet tied to a map point, against each previous frame (:488-512): per back frame one
    // batched call over the streams (the association state of a stream advances from one back frame to the next, so the
    // back frames stay sequential)
    for (int i = frameIdx - 1; i >= 0 && i >= frameIdx - cpe_.numBackTraverse; i--) {
        const size_t mi = (size_t)(frameIdx - 1 - i);
        std::vector<double> gx1, gx2, gRt1, gRt2, gK1, gK2;
        std::vector<int32_t> ofs(1, 0), gstream;
        std::vector<std::vector<std::pair<int, int>>> gpairs;
        for (int b = 0; b < B; b++) {
            if (mi >= ps[(size_t)b].cachedMatches.size()) continue;   // this stream left the loop earlier (:456)
            DataManager& data = *dms_[(size_t)b];
            Frame& cur = data.frames[(size_t)frameIdx];
            Frame& pre = data.frames[(size_t)i];
            Features& curF = cur.features;
            Features& preF = pre.features;
            std::vector<std::pair<int, int>> pairs;
            const size_t x0 = gx1.size();
            for (const cv::DMatch& m : ps[(size_t)b].cachedMatches[mi]) {
                const int cf = m.queryIdx, pf = m.trainIdx;
                // both ends unassociated; a feature associated earlier in THIS loop is skipped like in the reference,
                // where the test runs after each registerNewMapPoint
                if (preF.mapPointsIndices[(size_t)pf] == -1 && curF.mapPointsIndices[(size_t)cf] == -1) {
                    bool dup = false;
                    for (const auto& pr : pairs) dup |= (pr.first == pf || pr.second == cf);
                    if (dup) continue;
                    pairs.push_back({ pf, cf });
                    gx1.push_back(preF.positions[(size_t)pf].x); gx1.push_back(preF.positions[(size_t)pf].y);
                    gx2.push_back(curF.positions[(size_t)cf].x); gx2.push_back(curF.positions[(size_t)cf].y);
                }
            }
            if (pairs.empty()) { gx1.resize(x0); gx2.resize(x0); continue; }
            double m12[12], k9[9];
            matTo(pre.Rt, m12, 12); gRt1.insert(gRt1.end(), m12, m12 + 12);
            matTo(cur.Rt, m12, 12); gRt2.insert(gRt2.end(), m12, m12 + 12);
            matTo(pre.K, k9, 9); gK1.insert(gK1.end(), k9, k9 + 9);
            matTo(cur.K, k9, 9); gK2.insert(gK2.end(), k9, k9 + 9);
            ofs.push_back(ofs.back() + (int32_t)pairs.size());
            gstream.push_back(b);
            gpairs.push_back(std::move(pairs));
        }
        const int G = (int)gstream.size();
        if (G == 0) continue;
        std::vector<double> gX((size_t)ofs.back() * 3);
        chk(mslam_triangulate_batch(runner_->ctx(), gx1.data(), gx2.data(), ofs.data(), G, gRt1.data(), gRt2.data(), gK1.data(), gK2.data(), gX.data(), nullptr, nullptr),
            "mslam_triangulate_batch");
        for (int g = 0; g < G; g++) {
            DataManager& data = *dms_[(size_t)gstream[(size_t)g]];
            const double* Xn = gX.data() + (size_t)ofs[(size_t)g] * 3;
            for (size_t k = 0; k < gpairs[(size_t)g].size(); k++)
                registerNewMapPoint(data, Point3d(Xn[3 * k], Xn[3 * k + 1], Xn[3 * k + 2]), data.frames[(size_t)i], gpairs[(size_t)g][k].first,
                                    data.frames[(size_t)frameIdx], gpairs[(size_t)g][k].second);
        }
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
