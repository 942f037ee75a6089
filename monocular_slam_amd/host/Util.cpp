#include "Util.h"
#include <cfloat>
#include <cstring>
#include <unordered_map>
#include "Geometry.h"
#include "Runtime.h"

std::vector<double> Util::lastChi2History;

Frame* Util::findFrameById(std::vector<Frame>& frames, int target_id)
{
    for (Frame& f : frames) if (f.meta.frameID == target_id) return &f;
    return nullptr;
}

MapPoint* Util::findMapPointById(std::vector<MapPoint>& map_points, int target_id)
{
    for (MapPoint& m : map_points) if ((int)m.id == target_id) return &m;
    return nullptr;
}

void Util::GlobalBundleAdjustemnt(DataManager& data, int n_iterations, bool pb_stop_flag, const unsigned long n_loop_kf, const bool b_robust)
{
    Util::BundleAdjustment(data, data.frames, data.mapPoints, n_iterations, pb_stop_flag, n_loop_kf, b_robust);
}

namespace {

struct System {
    int nf = 0, nnzb = 0, npts = 0;
    std::vector<double> err, chi2, rho1, Hpp, Hll, Hpl, b;
    std::vector<int32_t> rowptr, col;
    double robust_chi2 = 0;
    mslam_ba_system view() {
        mslam_ba_system s{};
        s.err = err.data(); s.chi2 = chi2.data(); s.rho1 = rho1.data(); s.Hpp = Hpp.data(); s.Hll = Hll.data();
        s.Hpl_rowptr = rowptr.data(); s.Hpl_col = col.data(); s.Hpl_val = Hpl.data(); s.b = b.data();
        return s;
    }
};

// BlockSolver::solve with the Schur complement (block_solver.hpp:354-486); lambda is added to both diagonals (setLambda)
bool schurSolve(const System& s, double lambda, std::vector<double>& x)
{
    const int nf = s.nf, np6 = 6 * nf, npts = s.npts;
    std::vector<double> S((size_t)np6 * np6, 0.0), coeff((size_t)np6, 0.0), Dinv((size_t)npts * 9);
    for (int f = 0; f < nf; f++)
        for (int a = 0; a < 6; a++)
            for (int c = 0; c < 6; c++) S[(size_t)(6 * f + a) * np6 + 6 * f + c] = s.Hpp[(size_t)36 * f + a * 6 + c] + (a == c ? lambda : 0.0);
    // landmark-major view of the block-CSR
    std::vector<int> cptr((size_t)npts + 1, 0), crow((size_t)s.nnzb), cblk((size_t)s.nnzb), fill((size_t)npts, 0);
    for (int k = 0; k < s.nnzb; k++) cptr[(size_t)s.col[(size_t)k] + 1]++;
    for (int i = 0; i < npts; i++) cptr[(size_t)i + 1] += cptr[(size_t)i];
    for (int f = 0; f < nf; f++)
        for (int k = s.rowptr[(size_t)f]; k < s.rowptr[(size_t)f + 1]; k++) {
            const int i = s.col[(size_t)k];
            const int at = cptr[(size_t)i] + fill[(size_t)i]++;
            crow[(size_t)at] = f; cblk[(size_t)at] = k;
        }
    for (int i = 0; i < npts; i++) {
        double D[9];
        std::memcpy(D, &s.Hll[(size_t)9 * i], sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = &Dinv[(size_t)9 * i];
        geom::inv3(D, Di);
        const double* bl = &s.b[(size_t)np6 + 3 * (size_t)i];
        double db[3];
        for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * bl[0] + Di[a * 3 + 1] * bl[1] + Di[a * 3 + 2] * bl[2];
        for (int o = cptr[(size_t)i]; o < cptr[(size_t)i + 1]; o++) {
            const int i1 = crow[(size_t)o];
            const double* Bi = &s.Hpl[(size_t)18 * cblk[(size_t)o]];
            double BD[18];
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 3; c++) BD[a * 3 + c] = Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c] + Bi[a * 3 + 2] * Di[6 + c];
            for (int a = 0; a < 6; a++) coeff[(size_t)6 * i1 + a] += Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
            for (int o2 = o; o2 < cptr[(size_t)i + 1]; o2++) {
                const int i2 = crow[(size_t)o2];
                const double* Bj = &s.Hpl[(size_t)18 * cblk[(size_t)o2]];
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++)
                        S[(size_t)(6 * i1 + a) * np6 + 6 * i2 + c] -= BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1] + BD[a * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
    std::vector<double> bs((size_t)np6);
    for (int a = 0; a < np6; a++) bs[(size_t)a] = s.b[(size_t)a] - coeff[(size_t)a];
    if (np6 > 0 && !geom::ldltSolve(S.data(), np6, bs.data(), x.data())) return false;
    for (int i = 0; i < npts; i++) { // x_l = Dinv (b_l - Hpl^T x_p)
        double cl[3] = { s.b[(size_t)np6 + 3 * (size_t)i], s.b[(size_t)np6 + 3 * (size_t)i + 1], s.b[(size_t)np6 + 3 * (size_t)i + 2] };
        for (int o = cptr[(size_t)i]; o < cptr[(size_t)i + 1]; o++) {
            const double* Bi = &s.Hpl[(size_t)18 * cblk[(size_t)o]];
            const double* xp = &x[(size_t)6 * crow[(size_t)o]];
            for (int c = 0; c < 3; c++) for (int a = 0; a < 6; a++) cl[c] -= Bi[a * 3 + c] * xp[a];
        }
        const double* Di = &Dinv[(size_t)9 * i];
        for (int a = 0; a < 3; a++) x[(size_t)np6 + 3 * (size_t)i + a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
    }
    return true;
}

} // namespace

void Util::BundleAdjustment(DataManager& data, std::vector<Frame>& frames, std::vector<MapPoint>& map_points, int n_iterations,
                            bool /*pb_stop_flag*/, const unsigned long n_loop_kf, const bool b_robust)
{
    runBA(data, frames, map_points, n_iterations, b_robust, 1, nullptr, (double)THRESH_HUBER_FULL_BA, n_loop_kf == 0);
}

void Util::LocalBundleAdjustment(DataManager& data, int first_free_frame_id, int n_iterations, const bool b_robust)
{
    runBA(data, data.frames, data.mapPoints, n_iterations, b_robust, first_free_frame_id < 1 ? 1 : first_free_frame_id, nullptr,
          (double)THRESH_HUBER_FULL_BA, true);
}

cv::Mat Util::ComputeF(Frame f1, Frame f2)
{
    if (f1.Rt.empty() || f2.Rt.empty() || f1.K.empty() || f2.K.empty()) return cv::Mat();
    double R1[9], R2[9], t1[3], t2[3], K1[9], K2[9];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { R1[i * 3 + j] = f1.Rt.at<double>(i, j); R2[i * 3 + j] = f2.Rt.at<double>(i, j);
                                      K1[i * 3 + j] = f1.K.at<double>(i, j); K2[i * 3 + j] = f2.K.at<double>(i, j); }
        t1[i] = f1.Rt.at<double>(i, 3); t2[i] = f2.Rt.at<double>(i, 3);
    }
    // X2 = R X1 + t with R = R2 R1^T, t = t2 - R t1
    double R[9], t[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R2[i * 3 + k] * R1[j * 3 + k]; R[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) t[i] = t2[i] - (R[i * 3] * t1[0] + R[i * 3 + 1] * t1[1] + R[i * 3 + 2] * t1[2]);
    const double tx[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 };
    double E[9], K1i[9], K2i[9], T[9], F[9];
    geom::matmul3(tx, R, E);
    if (!geom::inv3(K1, K1i) || !geom::inv3(K2, K2i)) return cv::Mat();
    geom::matmul3(E, K1i, T);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += K2i[k * 3 + i] * T[k * 3 + j]; F[i * 3 + j] = s; } // K2^-T * T
    cv::Mat out(3, 3, cv::CV_64F);
    for (int i = 0; i < 9; i++) out.at<double>(i / 3, i % 3) = F[i];
    return out;
}

bool Util::poseBARoundsEnabled = false;
std::vector<uint8_t> Util::lastPoseBAOutliers;

void Util::PoseBundleAdjustment(Frame& frame, DataManager& data, int n_round)
{
    // one round of POSE_BA_ITER iterations (the reference leaves its round loop after the first optimize, :312),
    // or the full n_round loop of :305-339 when poseBARoundsEnabled
    runBA(data, data.frames, data.mapPoints, POSE_BA_ITER, true, 0, &frame, (double)THRESH_HUBER, true,
          poseBARoundsEnabled ? std::max(1, n_round) : 1);
}

// frames with frameID < first_free_frame_id are fixed (1 == the reference's "fix frame 0", src/Util.cpp:72).
// only_pose_frame != nullptr: pose-only BA of that frame -- its pose is the single free vertex, every map point it
// observes is a fixed parameter (EdgeSE3ProjectXYZOnlyPose): no point blocks, a 6x6 system per LM trial.
void Util::runBA(DataManager& data, std::vector<Frame>& frames, std::vector<MapPoint>& map_points, int n_iterations, bool b_robust,
                 int first_free_frame_id, Frame* only_pose_frame, double huber_delta, bool write_back, int pose_rounds)
{
    const bool poseOnly = only_pose_frame != nullptr;
    lastChi2History.clear();
    // ---- graph construction (src/Util.cpp:61-169): pose vertices for frames with an Rt, fixed iff frameID == 0
    std::vector<int> poseOfFrame;             // index into `frames`
    std::unordered_map<int, int> poseIndexOfId; // frameID -> pose index
    std::vector<double> poses, intr;
    std::vector<uint8_t> fixed;
    for (size_t i = 0; i < frames.size(); i++) {
        Frame& f = frames[i];
        if (f.Rt.empty()) continue;
        if (poseOnly && &f != only_pose_frame) continue;
        double Rt[12], p7[7];
        for (int k = 0; k < 12; k++) Rt[k] = f.Rt.at<double>(k / 4, k % 4);
        geom::RtToSE3Quat(Rt, p7);            // Converter::cvMatToSE3Quat
        poseIndexOfId[f.meta.frameID] = (int)poseOfFrame.size();
        poseOfFrame.push_back((int)i);
        poses.insert(poses.end(), p7, p7 + 7);
        fixed.push_back(poseOnly ? 0 : (f.meta.frameID < first_free_frame_id));
        intr.push_back(f.K.at<double>(0, 0)); intr.push_back(f.K.at<double>(1, 1)); intr.push_back(f.K.at<double>(0, 2)); intr.push_back(f.K.at<double>(1, 2));
    }
    // point vertices for map points with at least one observer among those frames; one edge per such observation
    std::vector<int> pointOfMap((size_t)map_points.size(), -1), mapOfPoint;
    std::vector<double> points, obs, info;
    std::vector<int32_t> edge_pose, edge_point;
    for (size_t i = 0; i < map_points.size(); i++) {
        MapPoint& mp = map_points[i];
        bool any = false;
        for (const auto& ob : mp.observers()) {
            Frame* o = findFrameById(data.frames, ob.first);
            if (o && poseIndexOfId.count(o->meta.frameID)) { any = true; break; }
        }
        if (!any) continue;
        const int pi = (int)mapOfPoint.size();
        pointOfMap[i] = pi; mapOfPoint.push_back((int)i);
        points.push_back(mp.worldPosition.x); points.push_back(mp.worldPosition.y); points.push_back(mp.worldPosition.z);
        for (const auto& ob : mp.observers()) {
            Frame* o = findFrameById(data.frames, ob.first);
            if (!o || !poseIndexOfId.count(o->meta.frameID)) continue;
            if (ob.second < 0 || (size_t)ob.second >= o->features.positions.size() || (size_t)ob.second >= o->features.scales.size())
                continue; // inconsistent fixture: the reference would read out of bounds here (src/Util.cpp:136,151)
            const cv::Point2f pos(o->features.positions[(size_t)ob.second]);   // truncated to float (:136)
            const float scale = (float)o->features.scales[(size_t)ob.second];  // (:151)
            edge_pose.push_back(poseIndexOfId[o->meta.frameID]); edge_point.push_back(pi);
            obs.push_back((double)pos.x); obs.push_back((double)pos.y);
            info.push_back((double)(1.0f / scale));                            // Identity * (1.0f/scale) (:153)
        }
    }
    if (edge_pose.empty() || poseOfFrame.empty()) return;
    mslam_ba_problem prob{};
    prob.n_poses = (int)poseOfFrame.size(); prob.n_points = (int)mapOfPoint.size(); prob.n_edges = (int)edge_pose.size();
    prob.poses = poses.data(); prob.pose_fixed = fixed.data(); prob.points = points.data();
    prob.edge_pose = edge_pose.data(); prob.edge_point = edge_point.data(); prob.obs = obs.data(); prob.info = info.data();
    prob.intr = intr.data(); prob.huber_delta = b_robust ? huber_delta : 0.0;

    mslam_ctx* ctx = MslamRuntime::instance().ctx();
    if (!ctx) { MslamRuntime::instance().ensure(640, 480, 1024); ctx = MslamRuntime::instance().ctx(); }
    mslam_ba_plan* plan = nullptr;
    MslamRuntime::check(mslam_ba_plan_create(ctx, &prob, &plan), "mslam_ba_plan_create");
    System s;
    MslamRuntime::check(mslam_ba_sizes(&prob, &s.nf, &s.nnzb), "mslam_ba_sizes");
    s.npts = prob.n_points;
    s.err.resize((size_t)prob.n_edges * 2); s.chi2.resize((size_t)prob.n_edges); s.rho1.resize((size_t)prob.n_edges);
    s.Hpp.resize((size_t)36 * std::max(1, s.nf)); s.Hll.resize((size_t)9 * s.npts); s.Hpl.resize((size_t)18 * std::max(1, s.nnzb));
    s.b.resize((size_t)6 * s.nf + 3 * (size_t)s.npts); s.rowptr.resize((size_t)s.nf + 1); s.col.resize((size_t)std::max(1, s.nnzb));
    auto linearise = [&](bool download_blocks) {
        MslamRuntime::check(mslam_ba_plan_set_state(plan, poses.data(), points.data()), "mslam_ba_plan_set_state");
        MslamRuntime::check(mslam_ba_plan_build_dev(plan), "mslam_ba_plan_build_dev");
        mslam_ba_system v = s.view();
        if (!download_blocks) { v.Hpp = nullptr; v.Hll = nullptr; v.Hpl_val = nullptr; v.b = nullptr; v.err = nullptr; v.chi2 = nullptr; v.rho1 = nullptr; }
        MslamRuntime::check(mslam_ba_plan_download(plan, &v), "mslam_ba_plan_download");
        s.robust_chi2 = v.robust_chi2;
    };

    // ---- OptimizationAlgorithmLevenberg::solve per outer iteration (optimization_algorithm_levenberg.cpp:61-164).
    // Full BA: the Schur complement and the landmark back-substitution run on the device (mslam_ba_plan_schur_dev /
    // _backsub_dev); per LM trial the host only sees the reduced pose system (6 fp x 6 fp), solves it by LDL^T, updates
    // the poses (SE3 exp) and reads back one chi2.  Pose-only BA and plans without Schur scratch use the host solve.
    const int nx = 6 * s.nf + 3 * s.npts, n6 = 6 * s.nf;
    std::vector<double> x((size_t)nx, 0.0), bkPoses, bkPoints, S((size_t)n6 * n6), bsv((size_t)n6);
    bool gpuSchur = !poseOnly && mslam_ba_plan_schur_dev(plan, 0.0) == MSLAM_OK; // probe (MSLAM_ERR_CAPACITY: not available)
    double lambda = -1, ni = 2;
    int nBad = 0;
    auto chi2_only = [&]() {
        MslamRuntime::check(mslam_ba_plan_build_dev(plan), "mslam_ba_plan_build_dev");
        mslam_ba_system v{};
        MslamRuntime::check(mslam_ba_plan_download(plan, &v), "mslam_ba_plan_download");
        return v.robust_chi2;
    };
    // ---- PoseBundleAdjustment's rounds (src/Util.cpp:305-339); every other entry point runs the loop body once
    const int rounds = poseOnly ? std::max(1, pose_rounds) : 1;
    const std::vector<double> poses0 = poses;
    std::vector<double> info_cur = info;
    std::vector<uint8_t> is_outlier((size_t)prob.n_edges, 1);             // "mark everyone as outlier first" (:296)
    bool robust_on = b_robust;
    for (int rd = 0; rd < rounds; rd++) {
    if (rounds > 1) {
        poses = poses0;                                                   // "reset pose every time" (:308)
        MslamRuntime::check(mslam_ba_plan_set_robust(plan, info_cur.data(), robust_on ? huber_delta : 0.0), "mslam_ba_plan_set_robust");
        lambda = -1; ni = 2; nBad = 0;
        lastChi2History.clear();                                          // diagnostics of the last round
    }
    for (int it = 0; it < n_iterations; it++) {
        linearise(true);
        double currentChi = s.robust_chi2, tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) { // computeLambdaInit: tau * max diagonal
            double maxd = 0;
            for (int f = 0; f < s.nf; f++) for (int a = 0; a < 6; a++) maxd = std::max(maxd, std::fabs(s.Hpp[(size_t)36 * f + 7 * a]));
            if (!poseOnly)
                for (int i = 0; i < s.npts; i++) for (int a = 0; a < 3; a++) maxd = std::max(maxd, std::fabs(s.Hll[(size_t)9 * i + 4 * a]));
            lambda = 1e-5 * maxd; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            bkPoses = poses;                                            // push
            if (gpuSchur) MslamRuntime::check(mslam_ba_plan_points_snapshot_dev(plan, 0), "points_snapshot");
            else bkPoints = points;
            bool ok2;
            double scale = 0;
            if (poseOnly) { // BlockSolver without Schur: (Hpp + lambda I) x = b_p
                std::vector<double> Hd((size_t)n6 * n6, 0.0);
                for (int f = 0; f < s.nf; f++)
                    for (int a = 0; a < 6; a++)
                        for (int c = 0; c < 6; c++) Hd[(size_t)(6 * f + a) * n6 + 6 * f + c] = s.Hpp[(size_t)36 * f + a * 6 + c] + (a == c ? lambda : 0.0);
                std::fill(x.begin(), x.end(), 0.0);
                ok2 = geom::ldltSolve(Hd.data(), n6, s.b.data(), x.data());
            } else if (gpuSchur) {
                MslamRuntime::check(mslam_ba_plan_schur_dev(plan, lambda), "mslam_ba_plan_schur_dev");
                MslamRuntime::check(mslam_ba_plan_download_schur(plan, S.data(), bsv.data()), "mslam_ba_plan_download_schur");
                ok2 = n6 == 0 || geom::ldltSolve(S.data(), n6, bsv.data(), x.data());
                if (!ok2) std::fill(x.begin(), x.begin() + n6, 0.0);
                MslamRuntime::check(mslam_ba_plan_backsub_dev(plan, x.data(), 1), "mslam_ba_plan_backsub_dev"); // X += x_l on the device
                MslamRuntime::check(mslam_ba_plan_download_x(plan, nullptr, &scale), "mslam_ba_plan_download_x");
            } else ok2 = schurSolve(s, lambda, x);
            for (int j = 0, f = 0; j < prob.n_poses; j++)             // oplus on the poses
                if (!fixed[(size_t)j]) { geom::se3ExpLeft(&x[(size_t)6 * f], &poses[(size_t)7 * j]); f++; }
            if (gpuSchur) {
                MslamRuntime::check(mslam_ba_plan_set_state(plan, poses.data(), nullptr), "mslam_ba_plan_set_state");
                tempChi = chi2_only();                                  // computeActiveErrors + activeRobustChi2
                for (int j = 0; j < n6; j++) scale += x[(size_t)j] * (lambda * x[(size_t)j] + s.b[(size_t)j]);
            } else {
                for (int i = 0; i < 3 * s.npts; i++) points[(size_t)i] += x[(size_t)6 * s.nf + i];
                const double keepChi = s.robust_chi2;
                linearise(false);
                tempChi = s.robust_chi2;
                s.robust_chi2 = keepChi;
                for (int j = 0; j < (poseOnly ? n6 : nx); j++) scale += x[(size_t)j] * (lambda * x[(size_t)j] + s.b[(size_t)j]);
            }
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                poses = bkPoses;                                        // pop
                if (gpuSchur) {
                    MslamRuntime::check(mslam_ba_plan_points_snapshot_dev(plan, 1), "points_restore");
                    MslamRuntime::check(mslam_ba_plan_set_state(plan, poses.data(), nullptr), "mslam_ba_plan_set_state");
                    if (rho < 0 && qmax + 1 < 10) (void)chi2_only();    // re-linearise at the restored state for the next trial
                } else points = bkPoints;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        lastChi2History.push_back(currentChi);
        if (gpuSchur) MslamRuntime::check(mslam_ba_plan_get_state(plan, nullptr, points.data()), "mslam_ba_plan_get_state");
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    if (rounds > 1 && rd != 0) {
        // classify at the optimised pose: e->computeError(); (float)e->chi2() <= CHI2_THRESH (:316-333).  Only edges
        // still marked outlier are looked at; an inlier stays an inlier.
        linearise(true);
        for (int j = 0; j < prob.n_edges; j++) {
            if (is_outlier[(size_t)j]) {
                const double ex = s.err[(size_t)2 * j], ey = s.err[(size_t)2 * j + 1];
                const float chi2_error = (float)(info[(size_t)j] * (ex * ex + ey * ey));
                if (chi2_error <= CHI2_THRESH) { is_outlier[(size_t)j] = 0; info_cur[(size_t)j] = info[(size_t)j]; }   // setLevel(0)
                else { is_outlier[(size_t)j] = 1; info_cur[(size_t)j] = 0.0; }                                          // setLevel(1)
            }
        }
        if (rd == rounds - 2) robust_on = false;                          // setRobustKernel(NULL) (:335)
    }
    } // rounds
    if (poseOnly) lastPoseBAOutliers = is_outlier;
    mslam_ba_plan_destroy(plan);

    // ---- write back (src/Util.cpp:184-219)
    if (write_back) {
        for (size_t k = 0; k < poseOfFrame.size(); k++) {
            // fixed poses did not move: leave their Rt alone instead of sending it through the quaternion round trip (which
            // may flip last bits; the reference writes every vertex back, but its only fixed vertex is frame 0 = identity,
            // for which the round trip is exact)
            if (fixed[k]) continue;
            double Rt[12];
            geom::SE3QuatToRt(&poses[7 * k], Rt);                      // Converter::SE3QuatToCvMat
            cv::Mat m(3, 4, cv::CV_64F);
            for (int q = 0; q < 12; q++) m.at<double>(q / 4, q % 4) = Rt[q];
            frames[(size_t)poseOfFrame[k]].Rt = m;
        }
        for (size_t k = 0; k < mapOfPoint.size() && !poseOnly; k++)
            map_points[(size_t)mapOfPoint[k]].worldPosition = Point3d(points[3 * k], points[3 * k + 1], points[3 * k + 2]);
    }
}
