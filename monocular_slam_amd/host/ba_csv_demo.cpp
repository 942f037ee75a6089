// ba_csv_demo -- the reference's only result-checking test, unitTestFullBA (UnitTest/g2o/UnitTestg2o.cpp:31-82), as a
// tool: load Frames.csv + MapPoints.csv (reference formats), set the intrinsics on every frame (:33-43), run
// Util::GlobalBundleAdjustemnt(dm, 15) [or local / pose-only BA], write FramesOptimised.csv + MapPointsOptimised.csv.
//   ba_csv_demo Frames.csv MapPoints.csv fx fy cx cy outFrames.csv outMapPoints.csv [--iters N] [--local FIRST_FREE_ID] [--pose FRAME_ID [--pose-rounds N]]
//   --pose-rounds N: run PoseBundleAdjustment's inlier/outlier rounds (src/Util.cpp:305-339, disabled in the reference by a `break`)
#include <cstdio>
#include <cstring>
#include "Util.h"

int main(int argc, char** argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s Frames.csv MapPoints.csv fx fy cx cy outFrames.csv outMapPoints.csv [--iters N] [--local ID] [--pose ID]\n", argv[0]); return 2; }
    int iters = FULL_BA_ITER, local = -1, pose = -1, pose_rounds = 0;
    for (int i = 9; i < argc; i++) {
        if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) iters = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--local") && i + 1 < argc) local = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pose") && i + 1 < argc) pose = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pose-rounds") && i + 1 < argc) pose_rounds = std::atoi(argv[++i]);
    }
    DataManager dm;
    dm.frames = Util::loadFrames(argv[1]);
    dm.mapPoints = Util::loadMapPoints(argv[2]);
    cv::Mat K = cv::Mat::eye(3, 3, cv::CV_64F);
    K.at<double>(0, 0) = std::atof(argv[3]); K.at<double>(1, 1) = std::atof(argv[4]);
    K.at<double>(0, 2) = std::atof(argv[5]); K.at<double>(1, 2) = std::atof(argv[6]);
    for (Frame& f : dm.frames) f.K = K;
    try {
        if (pose >= 0) {
            Frame* f = Util::findFrameById(dm.frames, pose);
            if (!f) { std::fprintf(stderr, "no frame %d\n", pose); return 2; }
            if (pose_rounds > 0) { Util::poseBARoundsEnabled = true; Util::PoseBundleAdjustment(*f, dm, pose_rounds); }
            else Util::PoseBundleAdjustment(*f, dm);
        } else if (local >= 0) Util::LocalBundleAdjustment(dm, local, iters);
        else Util::GlobalBundleAdjustemnt(dm, iters);
    } catch (const std::exception& e) { std::fprintf(stderr, "BA failed: %s\n", e.what()); return 1; }
    Util::saveFramesOnlyIdRt(dm.frames, argv[7]);
    Util::saveMapPointsOnlyIdXYZ(dm.mapPoints, argv[8]);
    std::printf("chi2");
    for (double v : Util::lastChi2History) std::printf(" %.17g", v);
    std::printf("\n");
    if (pose >= 0 && pose_rounds > 0) {
        std::printf("outliers");
        for (uint8_t v : Util::lastPoseBAOutliers) std::printf(" %d", (int)v);
        std::printf("\n");
    }
    return 0;
}
