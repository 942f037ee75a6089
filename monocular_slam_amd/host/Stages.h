// Stages.h -- the three ProcessingNode stages of the hot path, declared together: FeatureExtractor, CameraPoseEstimator,
// Optimiser.  Each keeps the reference's class name, constructor and process()/validationCheck() overrides
// (src/FeatureExtractor.h, src/CameraPoseEstimator.h, src/Optimiser.h) and adds the entry point north_star names
// (extract / estimate / localBA).  The per-stage headers of the reference's layout forward to this file.
#pragma once
#include "ProcessingNode.h"
#include "Runtime.h"

// FeatureExtractor -- the reference's stage (src/FeatureExtractor.h:16-25, src/FeatureExtractor.cpp:13-31) with its two
// OpenCV ORB members replaced by the C ABI.  `extract` is the new entry point north_star names; process() calls it and
// fills frame.features exactly as the reference does (:20-25).

class FeatureExtractor : public ProcessingNode {
public:
    FeatureExtractor();
    // detector.detect + extractor.compute (:17,:19): keypoints and N x 32 descriptors of one frame
    void extract(const cv::Mat& frameBuffer, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors);
    virtual void process(DataManager& data, int frameIdx) override;
    virtual bool validationCheck(DataManager& data, int frameIdx) override;
    mslam_orb_params params;   // defaults = default-constructed cv::ORB (500 features, 1.2, 8 levels, 31, HARRIS, 31)
    int capacitySlack = 1024;  // room for response ties (retainBest keeps every tie)
};

// CameraPoseEstimator -- the reference's stage (src/CameraPoseEstimator.h:19-29, src/CameraPoseEstimator.cpp:517-530):
// frame 0 gets the identity pose, frame 1 the two-view initialisation (:264-376), later frames PnP against the map
// (:384-515).  Every OpenCV call of the reference is replaced by a C ABI call (match, F RANSAC, PnP RANSAC,
// triangulation); essential-matrix decomposition, candidate vote and map bookkeeping stay host code, as in the reference.
// `estimate` is the new entry point north_star names; process() calls it and appends to relative_frame_poses (:528).
// The GUI call of the reference (drawEpipolarLine, :308-312, blocks on waitKey) is deliberately not reproduced.

class CameraPoseEstimator : public ProcessingNode {
public:
    CameraPoseEstimator();
    void estimate(DataManager& data, int frameIdx);
    // matchFeatures (:200-213) for callers outside the stage (tools, tests): kNN-2 + ratio test, query = d1, train = d2
    void match(const cv::Mat& d1, const cv::Mat& d2, std::vector<cv::DMatch>& matches, float ratio = 0.8f) { matchFeatures(d1, d2, matches, ratio); }
    virtual void process(DataManager& data, int frameIdx) override;
    virtual bool validationCheck(DataManager& data, int frameIdx) override;

    // parameters; defaults are the reference's values
    float ratioInitial = FEATURE_MATCH_RATIO_TEST;  // 0.85f (:280)
    float ratioTracking = 0.8f;                     // matchFeatures default (:200, :409)
    mslam_fmat_params fmat;                         // metric 1, MAX_DISTANCE 3.0, CONFIDENCE 0.85, CV_FM_8POINT refit (:563,:585)
    int fmatHypotheses = 2000;                      // OpenCV's RANSAC iteration cap
    // 8: hypotheses from the in-tree eight-point solver (computeFundamentalMatrix2, :596-786 -- north_star's "RANSAC eight-point");
    // 7: OpenCV's own estimator behind findFundamentalMat(CV_FM_RANSAC) at :563 (mslam_fmat_ransac7: 7-point samples with up to
    //    three matrices each, runLMeDS below 15 matches, the direct solution for exactly 7)
    int fmatSamplePoints = 8;
    mslam_pnp_params pnp;                           // 8 px, stop_inliers = 100 (OpenCV 2.4 solvePnPRansac defaults: reprojectionError, minInliersCount)
    int pnpHypotheses = 100;                        // OpenCV 2.4 default iterationsCount
    double distCoeff[8];                            // TUM freiburg1 coefficients hard-coded by the reference (:462-469)
    int numBackTraverse = 5;                        // (:388)
    // sample tables: geom::sampleTable(n, H, k, seed + frame index * 1000 + use).  At the reference's confidence (0.85) the F
    // RANSAC stops after a handful of hypotheses, so the two-view initialisation depends visibly on the table: on the rendered
    // test sequence the translation direction is off by 1 to 27 degrees over seeds 1..15 (tests/test_pipeline_gpu.py states
    // its truth thresholds for this default).
    uint64_t seed = 5;
    int lastStatus = 0;                             // MSLAM_OK or the status that made the stage keep the previous pose

private:
    void setReferenceFramePose(DataManager& data, int frameIdx);
    void initialPoseEstimation(DataManager& data, int frameIdx);
    void pnpPoseEstimation(DataManager& data, int frameIdx);
    void matchFeatures(const cv::Mat& d1, const cv::Mat& d2, std::vector<cv::DMatch>& matches, float ratio);
    bool computeFundamentalMatrix(const std::vector<Point2d>& p1, const std::vector<Point2d>& p2, const std::vector<cv::DMatch>& matches,
                                  std::vector<Point2d>& in1, std::vector<Point2d>& in2, double F[9], std::vector<unsigned char>& status,
                                  uint64_t tableSeed);
};

// Optimiser -- the reference's stage (src/Optimiser.h:5-18, src/Optimiser.cpp:6-18): for frameIdx != 0 it runs the
// full bundle adjustment over everything in the DataManager.  `localBA` is the new entry point north_star names
// (the reference only declares Util::LocalBundleAdjustment, src/Util.h:27, and never defines it); process() calls it.

class Optimiser : public ProcessingNode {
public:
    Optimiser() : ProcessingNode("Optimiser") {}
    // window <= 0: the reference's behaviour (global BA, frame 0 fixed); window = k: the last k posed frames free, older
    // ones fixed (Util::LocalBundleAdjustment -- declared, never defined upstream: src/Util.h:27, src/LocalMapper.h:30-42)
    void localBA(DataManager& data, int n_iterations = FULL_BA_ITER, int window = 0);
    virtual void process(DataManager& data, int frameIdx) override;
    virtual bool validationCheck(DataManager& data, int frameIdx) override;
    int n_iterations = FULL_BA_ITER;   // src/ParamConfig.h:18 (what Optimiser::process passes upstream, src/Optimiser.cpp:12)
    int window = 0;                    // process() default = the reference's global BA
};
