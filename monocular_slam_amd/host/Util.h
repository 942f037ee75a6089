// Util -- the optimisation entry points of the reference (src/Util.h:14-35): same names (including the reference's
// spelling "GlobalBundleAdjustemnt"), same defaults.  The g2o graph + Levenberg-Marquardt of src/Util.cpp:34-220 is
// replaced by: linearisation on the GPU (mslam_ba_plan_*), LM control + Schur complement + LDL^T on the host.
#pragma once
#include <map>
#include <string>
#include "DataManager.h"

class Util {
public:
    static void BundleAdjustment(DataManager& data, std::vector<Frame>& frames, std::vector<MapPoint>& map_points, int n_iterations = 5,
                                 bool pb_stop_flag = false, const unsigned long n_loop_kf = 0, const bool b_robust = true);
    static void GlobalBundleAdjustemnt(DataManager& data, int n_iterations = 10, bool pb_stop_flag = false,
                                       const unsigned long n_loop_kf = 0, const bool b_robust = true);
    // Sliding-window variant the reference only declares (src/Util.h:27; LocalMapper is a stub): poses of frames with
    // frameID < first_free_frame_id are held fixed, later ones and every point they observe are optimised.
    static void LocalBundleAdjustment(DataManager& data, int first_free_frame_id, int n_iterations = 5, const bool b_robust = true);
    // Single-pose refinement against fixed map points (src/Util.cpp:222-358: unary EdgeSE3ProjectXYZOnlyPose edges,
    // Huber delta = sqrt(5.991), POSE_BA_ITER iterations, effectively one round because of the `break` at :312).
    static void PoseBundleAdjustment(Frame& frame, DataManager& data, int n_round = 5);
    // false (default): as the reference executes today.  true: the n_round inlier/outlier rounds that the `break` at
    // src/Util.cpp:312 disables run as written (:305-339): every round restarts from the frame's pose and optimises the
    // level-0 edges; from the second round on, edges still marked outlier are re-classified by chi2 <= CHI2_THRESH
    // (outliers leave the optimisation), and after round n_round-2 the robust kernel is dropped.
    static bool poseBARoundsEnabled;
    // per edge of the last PoseBundleAdjustment call: 1 = classified outlier (all 1 if the rounds did not run)
    static std::vector<uint8_t> lastPoseBAOutliers;
    // Fundamental matrix implied by two posed frames, x2^T F x1 = 0 with F = K2^-T [t]x R K1^-1, (R, t) = pose 2 relative to
    // pose 1; 3x3 CV_64F, empty if a pose or K is missing.  (The reference declares it and returns an empty Mat, src/Util.cpp:24-26.)
    static cv::Mat ComputeF(Frame f1, Frame f2);
    static Frame* findFrameById(std::vector<Frame>& frames, int target_id);
    static MapPoint* findMapPointById(std::vector<MapPoint>& map_points, int target_id);
    // CSV fixtures in the reference's formats (src/Util.cpp:384-588) and the matching writers (the reference has none)
    static std::vector<Frame> loadFrames(std::string frame_filename);
    static std::vector<MapPoint> loadMapPoints(std::string map_points_filename);
    static std::vector<Frame> loadFramesOnlyIdRt(std::string frame_filename);
    static std::vector<MapPoint> loadMapPointsOnlyIdXYZ(std::string map_points_filename);
    // "<frame id> <image file name>" per line (src/Util.cpp:590-612; UnitTest/compareORBSLAM reads out_info.txt with it)
    static std::map<int, std::string> loadFrameIdToImageFileName(std::string info_filename);
    static void saveFrames(const std::vector<Frame>& frames, std::string frame_filename);
    static void saveMapPoints(const std::vector<MapPoint>& map_points, std::string map_points_filename);
    static void saveFramesOnlyIdRt(const std::vector<Frame>& frames, std::string frame_filename);
    static void saveMapPointsOnlyIdXYZ(const std::vector<MapPoint>& map_points, std::string map_points_filename);
    // diagnostics of the last BundleAdjustment call: robust chi2 after each outer iteration
    static std::vector<double> lastChi2History;
private:
    static void runBA(DataManager& data, std::vector<Frame>& frames, std::vector<MapPoint>& map_points, int n_iterations, bool b_robust,
                      int first_free_frame_id, Frame* only_pose_frame, double huber_delta, bool write_back, int pose_rounds = 1);
};
