// Util -- the optimisation entry points of the reference (src/Util.h:14-35): same names (including the reference's
// spelling "GlobalBundleAdjustemnt"), same defaults.  The g2o graph + Levenberg-Marquardt of src/Util.cpp:34-220 is
// replaced by: linearisation on the GPU (mslam_ba_plan_*), LM control + Schur complement + LDL^T on the host.
#pragma once
#include "DataManager.h"

class Util {
public:
    static void BundleAdjustment(DataManager& data, std::vector<Frame>& frames, std::vector<MapPoint>& map_points, int n_iterations = 5,
                                 bool pb_stop_flag = false, const unsigned long n_loop_kf = 0, const bool b_robust = true);
    static void GlobalBundleAdjustemnt(DataManager& data, int n_iterations = 10, bool pb_stop_flag = false,
                                       const unsigned long n_loop_kf = 0, const bool b_robust = true);
    static Frame* findFrameById(std::vector<Frame>& frames, int target_id);
    static MapPoint* findMapPointById(std::vector<MapPoint>& map_points, int target_id);
    // diagnostics of the last BundleAdjustment call: robust chi2 after each outer iteration
    static std::vector<double> lastChi2History;
};
