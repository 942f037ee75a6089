#include "CameraPoseEstimator.h"
#include <cstring>
#include "Geometry.h"

namespace {

void matTo(const cv::Mat& m, double* out, int n) { for (int i = 0; i < n; i++) out[i] = m.at<double>(i / m.cols, i % m.cols); }
cv::Mat matFrom(const double* v, int r, int c) { cv::Mat m(r, c, cv::CV_64F); for (int i = 0; i < r * c; i++) m.at<double>(i / c, i % c) = v[i]; return m; }

void flatten(const std::vector<Point2d>& p, std::vector<double>& out)
{
    out.resize(p.size() * 2);
    for (size_t i = 0; i < p.size(); i++) { out[2 * i] = p[i].x; out[2 * i + 1] = p[i].y; }
}

// associateFeatureWithMapPoint / registerNewMapPoint (:218-243)
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

CameraPoseEstimator::CameraPoseEstimator() : ProcessingNode("CameraPoseEstimator")
{
    mslam_fmat_default_params(&fmat);
    fmat.threshold = MAX_DISTANCE; fmat.confidence = CONFIDENCE;
    mslam_pnp_default_params(&pnp);
    pnp.stop_inliers = 100;   // solvePnPRansac's default minInliersCount: the call at :472 passes none, so its iterations end at the first pose with 100 inliers
    const double d[8] = { 0.2624, -0.9531, -0.0054, 0.0026, 1.1633, 0, 0, 0 };
    std::memcpy(distCoeff, d, sizeof(d));
}

void CameraPoseEstimator::matchFeatures(const cv::Mat& d1, const cv::Mat& d2, std::vector<cv::DMatch>& matches, float ratio)
{
    matches.clear();
    const int nq = d1.rows, nt = d2.rows;
    if (nq == 0 || nt < 2) return; // the reference indexes raw_matches[i][1] unguarded (:209); here: no matches
    std::vector<int32_t> qi((size_t)nq), ti((size_t)nq);
    std::vector<float> dd((size_t)nq);
    MslamRuntime& rt = MslamRuntime::instance();
    int n = 0;
    DeviceFeatures* a = rt.deviceFeatures(d1.device);
    DeviceFeatures* b = rt.deviceFeatures(d2.device);
    // a twin counts only while the host rows it was attached to are unchanged (cv::Mat semantics allow in-place edits)
    if (a && (a->n != nq || d1.cols != 32 || a->content != contentHash(d1.data, (size_t)nq * 32))) a = nullptr;
    if (b && (b->n != nt || d2.cols != 32 || b->content != contentHash(d2.data, (size_t)nt * 32))) b = nullptr;
    if (a && b && a->cap == b->cap) {
        // both descriptor sets are still on the device (left there by FeatureExtractor::extract): kNN-2, ratio test and the
        // ordered compaction run there, only the surviving (query, train, distance) triples come back
        const size_t cap = (size_t)a->cap;
        uint8_t* out = (uint8_t*)rt.scratch(2, cap * (2 * 4 * 2 + 3 * 4) + 256);
        int32_t* idx = (int32_t*)out; int32_t* dist = idx + 2 * cap;
        int32_t* mq = dist + 2 * cap; int32_t* mt = mq + cap; float* md = (float*)(mt + cap); int32_t* mc = (int32_t*)(md + cap);
        MslamRuntime::check(mslam_match_knn2_batch_dev(rt.ctx(), a->desc, a->n_dev, b->desc, b->n_dev, (int)cap, 1, idx, dist, ratio, mq, mt, md, mc),
                            "mslam_match_knn2_batch_dev");
        int32_t hc = 0;
        MslamRuntime::check(mslam_memcpy_d2h(rt.ctx(), &hc, mc, sizeof(hc)), "mslam_memcpy_d2h");
        n = hc;
        if (n > 0) {
            MslamRuntime::check(mslam_memcpy_d2h_async(rt.ctx(), qi.data(), mq, sizeof(int32_t) * (size_t)n), "mslam_memcpy_d2h_async");
            MslamRuntime::check(mslam_memcpy_d2h_async(rt.ctx(), ti.data(), mt, sizeof(int32_t) * (size_t)n), "mslam_memcpy_d2h_async");
            MslamRuntime::check(mslam_memcpy_d2h(rt.ctx(), dd.data(), md, sizeof(float) * (size_t)n), "mslam_memcpy_d2h");
        }
    } else {
        // host-pointer form (descriptors that did not come from this process's extract, e.g. loaded fixtures)
        std::vector<int32_t> idx((size_t)nq * 2), dist((size_t)nq * 2);
        MslamRuntime::check(mslam_match_knn2(rt.ctx(), d1.data, nq, d2.data, nt, idx.data(), dist.data()), "mslam_match_knn2");
        MslamRuntime::check(mslam_ratio_filter(idx.data(), dist.data(), nq, ratio, qi.data(), ti.data(), dd.data(), &n), "mslam_ratio_filter");
    }
    matches.resize((size_t)n);
    for (int i = 0; i < n; i++) { matches[(size_t)i].queryIdx = qi[(size_t)i]; matches[(size_t)i].trainIdx = ti[(size_t)i]; matches[(size_t)i].distance = dd[(size_t)i]; }
}

// computeFundamentalMatrix (:545-586): RANSAC, inlier lists, 8-point refit on the inliers (refit 2 does both in one call)
bool CameraPoseEstimator::computeFundamentalMatrix(const std::vector<Point2d>& p1, const std::vector<Point2d>& p2,
                                                   const std::vector<cv::DMatch>& matches, std::vector<Point2d>& in1,
                                                   std::vector<Point2d>& in2, double F[9], std::vector<unsigned char>& status,
                                                   uint64_t tableSeed)
{
    const int M = (int)matches.size();
    std::vector<Point2d> a((size_t)M), b((size_t)M);
    for (int i = 0; i < M; i++) { a[(size_t)i] = p1[(size_t)matches[(size_t)i].queryIdx]; b[(size_t)i] = p2[(size_t)matches[(size_t)i].trainIdx]; }
    status.assign((size_t)M, 0);
    in1.clear(); in2.clear();
    const int k = fmatSamplePoints == 7 ? 7 : 8;
    if (M < k) { lastStatus = MSLAM_ERR_TOO_FEW; return false; }
    std::vector<double> x1, x2;
    flatten(a, x1); flatten(b, x2);
    std::vector<int32_t> table = geom::sampleTable(M, fmatHypotheses, k, tableSeed);
    int best = -1, ninl = 0;
    const int rc = k == 7 ? mslam_fmat_ransac7(MslamRuntime::instance().ctx(), x1.data(), x2.data(), M, table.data(), fmatHypotheses, &fmat, 0, F,
                                               status.data(), nullptr, &best, &ninl)
                          : mslam_fmat_ransac8(MslamRuntime::instance().ctx(), x1.data(), x2.data(), M, table.data(), fmatHypotheses, &fmat, F,
                                               status.data(), nullptr, &best, &ninl);
    if (rc != MSLAM_OK) { lastStatus = rc; std::fill(status.begin(), status.end(), 0); return false; }
    for (int i = 0; i < M; i++) if (status[(size_t)i]) { in1.push_back(a[(size_t)i]); in2.push_back(b[(size_t)i]); }
    return true;
}

void CameraPoseEstimator::setReferenceFramePose(DataManager& data, int frameIdx) { data.frames[(size_t)frameIdx].Rt = cv::Mat::eye(3, 4, cv::CV_64F); }

void CameraPoseEstimator::initialPoseEstimation(DataManager& data, int frameIdx)
{
    Frame& cur = data.frames[(size_t)frameIdx];
    Frame& pre = data.frames[(size_t)frameIdx - 1];
    std::vector<cv::DMatch> matches;
    matchFeatures(pre.features.descriptors, cur.features.descriptors, matches, ratioInitial);
    double F[9];
    std::vector<Point2d> good1, good2;
    std::vector<unsigned char> status;
    cur.Rt = pre.Rt.clone(); // kept if the geometry below cannot be estimated
    if (!computeFundamentalMatrix(pre.features.positions, cur.features.positions, matches, good1, good2, F, status, seed + 1000ull * (uint64_t)frameIdx))
        return;
    double K1[9], K2[9], E[9], R[2][9], t[2][3];
    matTo(pre.K, K1, 9); matTo(cur.K, K2, 9);
    geom::essentialFromF(F, K1, K2, E);
    geom::extractRTfromE(E, R, t);
    // four candidates in the reference's order (R1|t1, R1|t2, R2|t1, R2|t2); keep the one with most points in front (:337-354)
    const double I34[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    std::vector<double> x1, x2, X(good1.size() * 3), bestX;
    flatten(good1, x1); flatten(good2, x2);
    int bestIdx = -1, maxCount = -1;
    double bestRt[12];
    for (int c = 0; c < 4; c++) {
        double cand[12];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) cand[i * 4 + j] = R[c / 2][i * 3 + j]; cand[i * 4 + 3] = t[c % 2][i]; }
        int count = 0;
        MslamRuntime::check(mslam_triangulate(MslamRuntime::instance().ctx(), x1.data(), x2.data(), (int)good1.size(), I34, cand, K1, K2,
                                              X.data(), nullptr, &count), "mslam_triangulate");
        if (maxCount < count) { maxCount = count; bestIdx = c; std::memcpy(bestRt, cand, sizeof(cand)); bestX = X; }
    }
    if (bestIdx < 0) { lastStatus = MSLAM_ERR_DEGENERATE; return; } // the reference indexes Rts[-1] here (:355)
    double preRt[12], out[12];
    matTo(pre.Rt, preRt, 12);
    geom::concatenateRts(preRt, bestRt, out);
    cur.Rt = matFrom(out, 3, 4);
    int count = 0;
    for (size_t i = 0; i < matches.size(); i++)
        if (status[i] == 1) {
            registerNewMapPoint(data, Point3d(bestX[3 * (size_t)count], bestX[3 * (size_t)count + 1], bestX[3 * (size_t)count + 2]), pre,
                                matches[i].queryIdx, cur, matches[i].trainIdx);
            count++;
        }
}

void CameraPoseEstimator::pnpPoseEstimation(DataManager& data, int frameIdx)
{
    std::vector<Frame>& frames = data.frames;
    Frame& cur = frames[(size_t)frameIdx];
    Features& curF = cur.features;
    const int numCur = (int)curF.positions.size();
    std::vector<bool> matched((size_t)numCur, false);
    std::vector<Point3d> mapPoints;
    std::vector<Point2d> imagePoints;
    std::vector<std::vector<cv::DMatch>> cachedMatches;
    int count = 0;
    for (int i = frameIdx - 1; i >= 0 && i >= frameIdx - numBackTraverse; i--) {
        Features& preF = frames[(size_t)i].features;
        std::vector<cv::DMatch> raw;
        matchFeatures(curF.descriptors, preF.descriptors, raw, ratioTracking);
        // epipolar filter (FILTERING_WITH_F, :412-427): F itself is discarded
        std::vector<Point2d> t0, t1;
        double Ftmp[9];
        std::vector<unsigned char> status;
        computeFundamentalMatrix(curF.positions, preF.positions, raw, t0, t1, Ftmp, status, seed + 1000ull * (uint64_t)frameIdx + (uint64_t)(frameIdx - i));
        size_t keep = 0;
        for (size_t j = 0; j < status.size(); j++) if (status[j]) raw[keep++] = raw[j];
        raw.resize(keep);
        cachedMatches.push_back(raw);
        for (size_t j = 0; j < raw.size(); j++) {
            const int cf = raw[j].queryIdx, pf = raw[j].trainIdx;
            if (preF.mapPointsIndices[(size_t)pf] != -1 && !matched[(size_t)cf]) {
                count++;
                matched[(size_t)cf] = true;
                const int mp = preF.mapPointsIndices[(size_t)pf];
                associate(data, mp, cur, cf);
                mapPoints.push_back(data.mapPoints[(size_t)mp].worldPosition);
                imagePoints.push_back(curF.positions[(size_t)cf]);
                if (count == numCur) break;
            }
        }
        if (count == numCur) break;
    }
    cur.Rt = frames[(size_t)frameIdx - 1].Rt.clone(); // kept if PnP cannot be solved
    const int P = (int)mapPoints.size();
    if (P >= 4) {
        std::vector<double> X((size_t)P * 3), x;
        for (int i = 0; i < P; i++) { X[3 * (size_t)i] = mapPoints[(size_t)i].x; X[3 * (size_t)i + 1] = mapPoints[(size_t)i].y; X[3 * (size_t)i + 2] = mapPoints[(size_t)i].z; }
        flatten(imagePoints, x);
        double K[9], rvec[3], tvec[3], Rt[12];
        matTo(cur.K, K, 9);
        std::vector<int32_t> table = geom::sampleTable(P, pnpHypotheses, 4, seed + 1000ull * (uint64_t)frameIdx + 99);
        std::vector<uint8_t> inl((size_t)P);
        int best = -1, ninl = 0;
        const int rc = mslam_pnp_ransac(MslamRuntime::instance().ctx(), X.data(), x.data(), P, K, distCoeff, table.data(), pnpHypotheses, &pnp,
                                        rvec, tvec, Rt, inl.data(), nullptr, &best, &ninl);
        if (rc == MSLAM_OK) cur.Rt = matFrom(Rt, 3, 4); // Rodrigues(rvec) + constructRt (:473-474)
        else lastStatus = rc;
    } else lastStatus = MSLAM_ERR_TOO_FEW;
    // triangulate the matches not yet tied to a map point, against each previous frame (:488-512)
    double curRt[12], curK[9];
    matTo(cur.Rt, curRt, 12); matTo(cur.K, curK, 9);
    for (int i = frameIdx - 1; i >= 0 && i >= frameIdx - numBackTraverse; i--) {
        Frame& pre = frames[(size_t)i];
        Features& preF = pre.features;
        const size_t mi = (size_t)(frameIdx - 1 - i);
        if (mi >= cachedMatches.size()) break;
        const std::vector<cv::DMatch>& matches = cachedMatches[mi];
        std::vector<double> x1, x2;
        std::vector<std::pair<int, int>> pairs;
        for (const cv::DMatch& m : matches) {
            const int cf = m.queryIdx, pf = m.trainIdx;
            // both ends unassociated; a feature associated earlier in THIS loop is skipped like in the reference,
            // where the test runs after each registerNewMapPoint
            if (preF.mapPointsIndices[(size_t)pf] == -1 && curF.mapPointsIndices[(size_t)cf] == -1) {
                bool dup = false;
                for (const auto& pr : pairs) dup |= (pr.first == pf || pr.second == cf);
                if (dup) continue;
                pairs.push_back({ pf, cf });
                x1.push_back(preF.positions[(size_t)pf].x); x1.push_back(preF.positions[(size_t)pf].y);
                x2.push_back(curF.positions[(size_t)cf].x); x2.push_back(curF.positions[(size_t)cf].y);
            }
        }
        if (pairs.empty()) continue;
        double preRt[12], preK[9];
        matTo(pre.Rt, preRt, 12); matTo(pre.K, preK, 9);
        std::vector<double> X(pairs.size() * 3);
        MslamRuntime::check(mslam_triangulate(MslamRuntime::instance().ctx(), x1.data(), x2.data(), (int)pairs.size(), preRt, curRt, preK, curK,
                                              X.data(), nullptr, nullptr), "mslam_triangulate");
        for (size_t k = 0; k < pairs.size(); k++)
            registerNewMapPoint(data, Point3d(X[3 * k], X[3 * k + 1], X[3 * k + 2]), pre, pairs[k].first, cur, pairs[k].second);
    }
}

void CameraPoseEstimator::estimate(DataManager& data, int frameIdx)
{
    lastStatus = MSLAM_OK;
    if (frameIdx == 0) setReferenceFramePose(data, frameIdx);
    else if (frameIdx == 1) initialPoseEstimation(data, frameIdx);
    else pnpPoseEstimation(data, frameIdx);
}

void CameraPoseEstimator::process(DataManager& data, int frameIdx)
{
    estimate(data, frameIdx);
    data.relative_frame_poses.push_back(data.frames[(size_t)frameIdx].Rt);
}

bool CameraPoseEstimator::validationCheck(DataManager&, int) { return true; }
