/*
 * oracle.h -- CPU restatement of the reference's tracking/mapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.  The
 * product (libmslam_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED.  The reference (eastgeneral2007/Monocular_SLAM) delegates every
 * arithmetic step of this path except bundle adjustment to OpenCV 2.4.13, which is not
 * vendored and not present here, and its g2o copy needs Eigen 3.2.9, also absent; the
 * reference holds no golden vectors for the path (its one result-checking test,
 * UnitTest/g2o/UnitTestg2o.cpp:31-82, reads CSV fixtures from a git-ignored data/
 * directory).  The functions below therefore restate
 *   - the reference's own code where it exists (src/CameraPoseEstimator.cpp,
 *     src/CommonMath.cpp, src/Converter.cpp, src/Util.cpp, ThirdParty/g2o/...), and
 *   - the published OpenCV 2.4.13 algorithms at the reference's call sites
 *     (cv::ORB, BFMatcher::knnMatch, findFundamentalMat, solvePnPRansac),
 * and are pinned only by mathematical invariants, by the two known-answer crumbs the
 * reference holds (UnitTest/g2o/UnitTestg2o.cpp:14-17,33-35) and by an independent numpy
 * statement of the same mathematics (tests/).  Each function cites what it follows.
 *
 * Arithmetic contract (what "bit-exact" means for the GPU path): IEEE-754 binary32 /
 * binary64, round-to-nearest-even, evaluated in source order with NO fused multiply-add
 * (built with -ffp-contract=off), cvRound = round-half-to-even.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- ORB (cv::ORB of OpenCV 2.4.13; reference src/FeatureExtractor.cpp:17,19) */
typedef struct {
    int nfeatures; float scale_factor; int nlevels, edge_threshold, first_level, wta_k,
        score_type, patch_size, fast_threshold;
} orc_orb_params;
typedef struct { float x, y, size, angle, response; int32_t octave; } orc_keypoint;

void orc_orb_default_params(orc_orb_params* p);
/* level geometry: sizes and the float scale table; returns number of levels */
int orc_orb_level_sizes(int w, int h, const orc_orb_params* p, int* lw, int* lh, float* sf, float* inv_sf, int* nfeat);
void orc_bgr2gray(const uint8_t* src, int w, int h, int stride, int channels, uint8_t* dst, int dstride);
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void orc_gaussian_kernel7_q8(int k[7]);
/* FAST-9/16 score map before non-max suppression: out[y*w+x] = score (>= thr) of every segment-test corner, else 0 */
void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* out);
/* FAST-9/16 score map after 3x3 non-max suppression: out[y*w+x] = score (>= thr) or 0 */
void orc_fast_nms_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* out);
float orc_harris_response(const uint8_t* img, int stride, int x, int y);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y);
float orc_fast_atan2(float y, float x);
void orc_orb_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t desc[32]);
/* full detect + describe; keypoints level-major / row-major; returns 0, or -3 if n > cap */
int orc_orb_extract(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* one pyramid level (optionally blurred) of the frame, for stage-wise parity */
int orc_orb_level(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p,
                  int level, int blurred, uint8_t* out, int* lw, int* lh);

/* ---------------- match (BFMatcher::knnMatch k=2; reference src/CameraPoseEstimator.cpp:200-213) */
int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
int orc_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, float ratio,
                     int32_t* query_idx, int32_t* train_idx, float* d, int* n_out);

/* ---------------- small dense linear algebra (cv::SVD stand-ins; reference src/CommonMath.cpp) */
/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, n <= 12):
 * eigenvalues ascending in w, eigenvectors in the COLUMNS of V (row-major n x n) */
void orc_jacobi_eig(const double* A, int n, double* w, double* V);
/* SVD of a 3x3 (A = U diag(s) Vt, s descending, via eig of A^T A) */
void orc_svd3(const double A[9], double U[9], double s[3], double Vt[9]);
double orc_det3(const double A[9]);
/* null vector of an 8x9 system by Gauss-Jordan with complete pivoting (A destroyed); 0 if rank-deficient */
int orc_null_8x9(double* A, double f[9]);

/* ---------------- fundamental matrix (reference src/CameraPoseEstimator.cpp:545-786) */
typedef struct { int metric; double threshold; double confidence; int refit; } orc_fmat_params;
void orc_fmat_default_params(orc_fmat_params* p);
/* eight-point solve from the rows `sel` (n >= 8) of the pairs, coordinates pre-divided by `maxt`
 * and rounded through float as the reference does (:615-633); F for the scaled coordinates, rank 2 */
void orc_fmat_from_rows(const double* x1, const double* x2, const int32_t* sel, int n, double maxt, double F[9]);
/* Hartley-normalised 8-point on rows sel (CV_FM_8POINT) in pixel coordinates */
int orc_fmat_8point_normalized(const double* x1, const double* x2, const int32_t* sel, int n, double F[9]);
double orc_fmat_error(const double F[9], const double* p1, const double* p2, int metric);
int orc_fmat_ransac8(const double* x1, const double* x2, int M, const int32_t* samples, int H,
                     const orc_fmat_params* prm, double F[9], uint8_t* status, int32_t* inlier_counts,
                     int* best_h, int* n_inliers);
/* findFundamentalMat(CV_FM_RANSAC) as the reference calls it (:563): 7-point minimal samples, up to 3 matrices each */
int orc_solve_cubic(double c3, double c2, double c1, double c0, double r[3]);
int orc_fmat_7point(const double* x1, const double* x2, const int32_t* sel, double F[27]);
int orc_fmat_ransac7(const double* x1, const double* x2, int M, const int32_t* samples, int H,
                     const orc_fmat_params* prm, int method, double F[9], uint8_t* status, int32_t* inlier_counts,
                     int* best_model, int* n_inliers);
/* adaptive-stop emulation shared by F and PnP: returns winning hypothesis over ordered counts */
int orc_ransac_select(const int32_t* counts, int H, int n_points, int sample_size, double confidence, int* best_count);
int orc_ransac_select_stop(const int32_t* counts, int H, int n_points, int sample_size, double confidence, int stop_count, int* best_count);

/* ---------------- pose from F, triangulation (reference :86-194, :248-255, :316-356) */
int orc_triangulate(const double* x1, const double* x2, int n, const double Rt1[12], const double Rt2[12],
                    const double K1[9], const double K2[9], double* X, uint8_t* front, int* n_front);
/* E = K2^T F K1; decompose; cheirality vote; returns best candidate index 0..3, Rt (3x4) */
int orc_pose_from_F(const double F[9], const double K1[9], const double K2[9], const double* x1, const double* x2,
                    int n, double Rt[12], double* X, int* counts4);

/* ---------------- PnP (solvePnPRansac contract; reference src/CameraPoseEstimator.cpp:462-474) */
typedef struct { double reproj_thr; double confidence; int refine_iters; int min_inliers; int stop_inliers; } orc_pnp_params;
void orc_pnp_default_params(orc_pnp_params* p);
/* P3P: up to 4 poses (R row-major 3x3, t) from 3 unit bearings f and 3 world points X; returns count */
int orc_p3p(const double f[9], const double X[9], double R[4][9], double t[4][3]);
void orc_undistort_normalize(const double* x, int n, const double K[9], const double dist[8], double* xn);
void orc_project(const double Rt[12], const double K[9], const double dist[8], const double X[3], double uv[2]);
void orc_rodrigues_to_R(const double r[3], double R[9]);
void orc_R_to_rodrigues(const double R[9], double r[3]);
int orc_pnp_hypothesis(const double* X, const double* xn, const double* x, const double K[9], const double dist[8],
                       const int32_t s[4], double Rt[12]);
int orc_pnp_refine(const double* X, const double* x, int P, const uint8_t* mask, const double K[9], const double dist[8],
                   int iters, double Rt[12]);
int orc_pnp_ransac(const double* X, const double* x, int P, const double K[9], const double dist[8],
                   const int32_t* samples, int H, const orc_pnp_params* prm, double rvec[3], double tvec[3],
                   double Rt[12], uint8_t* inliers, int32_t* inlier_counts, int* best_h, int* n_inliers);

/* ---------------- bundle adjustment (reference src/Util.cpp:34-220 + ThirdParty/g2o) */
typedef struct {
    int n_poses, n_points, n_edges;
    const double* poses; const uint8_t* pose_fixed; const double* points;
    const int32_t* edge_pose; const int32_t* edge_point; const double* obs; const double* info;
    const double* intr; double huber_delta;
} orc_ba_problem;
typedef struct {
    double* err; double* chi2; double* rho1; double* Hpp; double* Hll;
    int32_t* Hpl_rowptr; int32_t* Hpl_col; double* Hpl_val; double* b;
    double robust_chi2; int n_free_poses; int nnzb;
} orc_ba_system;
int orc_ba_sizes(const orc_ba_problem* p, int* n_free_poses, int* nnzb);
int orc_ba_build(const orc_ba_problem* p, orc_ba_system* s);
/* g2o SE3Quat pieces */
void orc_Rt_to_se3quat(const double Rt[12], double pose7[7]);   /* Converter::cvMatToSE3Quat */
void orc_se3quat_to_Rt(const double pose7[7], double Rt[12]);   /* Converter::SE3QuatToCvMat */
void orc_se3_exp_left(const double delta6[6], double pose7[7]); /* T <- exp(delta) * T */
/* full LM optimisation (OptimizationAlgorithmLevenberg + BlockSolver Schur), in place */
int orc_ba_optimize(orc_ba_problem* p, double* poses, double* points, int iterations, double* chi2_history, int* n_hist);
/* one linearisation + Schur complement (BlockSolver::solve): S (6nf x 6nf, full), b_s, x = (x_p, x_l) */
int orc_ba_schur(const orc_ba_problem* p, double lambda, double* S, double* bs, double* x);
/* points_fixed != 0: pose-only BA (Util::PoseBundleAdjustment, src/Util.cpp:222-358) */
int orc_ba_optimize_ex(orc_ba_problem* p, double* poses, double* points, int iterations, int points_fixed, double* chi2_history, int* n_hist);

#ifdef __cplusplus
}
#endif
#endif
