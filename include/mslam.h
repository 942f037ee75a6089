/*
 * mslam.h -- C ABI of libmslam_hip.so: the MI355X (gfx950) implementation of the
 * per-frame tracking/mapping hot path of a monocular ORB SLAM pipeline.
 *
 * Every entry point states which interface of the reference (eastgeneral2007/
 * Monocular_SLAM, paths relative to its root) it stands in for.  The reference has no
 * FFI: its only plug-in surface is the C++ virtual `ProcessingNode::process(DataManager&,
 * int)` (src/ProcessingNode.h:16-32), and all arithmetic of this path sits behind the
 * OpenCV 2.4.13 / g2o calls made from the three stage classes.  The entry points below
 * are exactly those call sites, flattened to plain pointers and sizes.
 *
 * Conventions
 *  - every function returns an int status: MSLAM_OK (0) or a negative mslam_status.
 *  - the caller owns every buffer; the library owns only the opaque context, its HIP
 *    stream and its device scratch.  One context per GPU; a context is not thread-safe;
 *    distinct contexts are independent.
 *  - functions without a suffix take HOST pointers, copy in, run the device path, copy
 *    out and synchronise.  `_dev` functions take DEVICE pointers (hipMalloc memory on the
 *    context's device), enqueue on the context's stream and return without synchronising
 *    (use mslam_sync).  There is no CPU fallback anywhere in this library: without a
 *    usable gfx950 device mslam_create fails with MSLAM_ERR_NO_DEVICE.
 *  - "batch" = independent frames/problems processed by one launch sequence.
 */
#ifndef MSLAM_H
#define MSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MSLAM_OK = 0,
    MSLAM_ERR_INVALID = -1,     /* bad argument (null pointer, size out of range) */
    MSLAM_ERR_HIP = -2,         /* a HIP runtime call failed (see mslam_last_hip_error) */
    MSLAM_ERR_CAPACITY = -3,    /* result does not fit the capacity given at create/call */
    MSLAM_ERR_TOO_FEW = -4,     /* too few rows/points (e.g. kNN-2 with < 2 train rows:
                                   the reference indexes raw_matches[i][1] unguarded,
                                   src/CameraPoseEstimator.cpp:209) */
    MSLAM_ERR_NO_DEVICE = -5,   /* no usable gfx950 device */
    MSLAM_ERR_DEGENERATE = -6   /* no valid model found (all hypotheses degenerate) */
} mslam_status;

typedef struct mslam_ctx mslam_ctx;

/* ---- context ---------------------------------------------------------------------- */

/* device: HIP ordinal.  max_w/max_h: largest frame.  max_kpts: capacity of a per-frame
 * keypoint list (>= nfeatures + room for response ties, see mslam_orb_extract).
 * batch: frames per launch sequence (>= 1). */
int mslam_create(mslam_ctx** out, int device, int max_w, int max_h, int max_kpts, int batch);
void mslam_destroy(mslam_ctx* ctx);
int mslam_sync(mslam_ctx* ctx);
/* the context's hipStream_t (as void*), for callers that time with HIP events */
void* mslam_stream(mslam_ctx* ctx);
const char* mslam_strerror(int status);
/* text of the last failing HIP call on this context ("" if none) */
const char* mslam_last_hip_error(mslam_ctx* ctx);
/* ABI version of this header: major*1000 + minor */
int mslam_abi_version(void);

/* device memory helpers so that FFI callers without a HIP binding can stage buffers */
int mslam_dev_alloc(mslam_ctx* ctx, void** dptr, uint64_t bytes);
int mslam_dev_free(mslam_ctx* ctx, void* dptr);
int mslam_memcpy_h2d(mslam_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int mslam_memcpy_d2h(mslam_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
/* Asynchronous forms: enqueued on the context's stream, no host sync (use pinned host memory from mslam_host_alloc for
 * real overlap).  Frame ingest uses them on a second context (= copy stream) with mslam_event_record on that context
 * and mslam_stream_wait_event on the compute context, so that the H2D of frame i+1 overlaps the kernels of frame i
 * (the reference loads frames from disk before the loop, src/main.cpp:35-37; a live camera would be this path). */
int mslam_memcpy_h2d_async(mslam_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int mslam_memcpy_d2h_async(mslam_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
int mslam_memcpy_d2d_async(mslam_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes);
/* `rows` rows of `width` bytes from a pitched device array into a pitched host array (the used left part of capacity-padded
 * per-stream tables, e.g. the first max-count columns of the B x cap match lists) */
int mslam_memcpy_d2h_2d_async(mslam_ctx* ctx, void* dst_host, uint64_t dst_pitch, const void* src_dev, uint64_t src_pitch, uint64_t width, uint64_t rows);
int mslam_host_alloc(mslam_ctx* ctx, void** hptr, uint64_t bytes);   /* pinned (page-locked) host memory */
int mslam_host_free(mslam_ctx* ctx, void* hptr);
/* make this context's stream wait for an event recorded (mslam_event_record) on another context of the same device */
int mslam_stream_wait_event(mslam_ctx* ctx, void* ev);

/* HIP-event timing on the context's stream (so FFI callers can bracket launches) */
int mslam_event_create(mslam_ctx* ctx, void** ev);
int mslam_event_destroy(mslam_ctx* ctx, void* ev);
int mslam_event_record(mslam_ctx* ctx, void* ev);
int mslam_event_elapsed_ms(mslam_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* syncs on stop */

/* Per-kernel timing: when enabled, every kernel launch on this context is bracketed by HIP events
 * on the context's stream (up to max_launches launches; 0 disables).  mslam_profile_read returns the
 * summed milliseconds and the launch count per kernel id (arrays of mslam_profile_kernel_count()). */
int mslam_profile_enable(mslam_ctx* ctx, int max_launches);
int mslam_profile_kernel_count(void);
const char* mslam_profile_kernel_name(int kid);
int mslam_profile_read(mslam_ctx* ctx, double* total_ms, int32_t* launches);

/* HIP graphs for launch-bound use (small batches, one stream): when enabled, the kernel chain of
 * mslam_orb_extract_batch_dev is captured once per distinct argument set on this context and replayed with
 * hipGraphLaunch afterwards (same kernels, same order, same results).  Off by default; MSLAM_GRAPHS=1 in the
 * environment enables it at mslam_create.  Ignored while per-kernel profiling is on. */
int mslam_set_graphs(mslam_ctx* ctx, int enable);

/* ---- R1-R6: ORB detect + describe --------------------------------------------------
 * Stands in for `detector.detect(frameBuffer, keypoints)` + `extractor.compute(
 * frameBuffer, keypoints, descriptor)` of FeatureExtractor::process
 * (src/FeatureExtractor.cpp:17,19), i.e. cv::ORB (OpenCV 2.4.13) with the parameters of
 * the default-constructed OrbFeatureDetector/OrbDescriptorExtractor
 * (src/FeatureExtractor.h:23-24). */
typedef struct {
    int nfeatures;       /* 500 */
    float scale_factor;  /* 1.2f */
    int nlevels;         /* 8  (1..MSLAM_MAX_LEVELS) */
    int edge_threshold;  /* 31 */
    int first_level;     /* 0 (only 0 supported) */
    int wta_k;           /* 2 (only 2 supported) */
    int score_type;      /* 0 = HARRIS_SCORE, 1 = FAST_SCORE */
    int patch_size;      /* 31 (only 31 supported: the learned pattern) */
    int fast_threshold;  /* 20 (hard-wired in OpenCV 2.4.13's ORB) */
} mslam_orb_params;
#define MSLAM_MAX_LEVELS 16
void mslam_orb_default_params(mslam_orb_params* p);

/* cv::KeyPoint fields the reference reads (pt, size) plus the ones ORB fills */
typedef struct {
    float x, y;      /* level-0 pixel coordinates */
    float size;      /* 31 * scale_factor^octave */
    float angle;     /* degrees, [0,360) */
    float response;  /* Harris (or FAST) score */
    int32_t octave;  /* pyramid level */
} mslam_keypoint;    /* 24 bytes */

/* One frame.  img: h rows of `stride` bytes, `channels` in {1,3,4} interleaved (BGR[A],
 * as loaded by imread(..., CV_LOAD_IMAGE_UNCHANGED), src/FrameLoader.cpp:62).
 * kps/desc: capacity `cap` keypoints / cap*32 bytes.  Keypoints come out level-major and
 * row-major (y, then x) inside a level; the kept set is OpenCV's retainBest set (all
 * keypoints whose response is >= the n-th best), so n_out may exceed nfeatures on ties;
 * MSLAM_ERR_CAPACITY if it exceeds cap or the context's max_kpts. */
int mslam_orb_extract(mslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                      const mslam_orb_params* params,
                      mslam_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batched, device-resident.  imgs: B frames, frame b at imgs + b*frame_stride.
 * kps: B*cap keypoints, desc: B*cap*32 bytes, n_out: B int32 (device).  B <= ctx batch.
 * status_dev (optional, B int32, device): per-frame MSLAM_OK / MSLAM_ERR_CAPACITY. */
int mslam_orb_extract_batch_dev(mslam_ctx* ctx, const uint8_t* imgs, int B, int w, int h, int stride,
                                int64_t frame_stride, int channels, const mslam_orb_params* params,
                                mslam_keypoint* kps, uint8_t* desc, int cap, int32_t* n_out,
                                int32_t* status_dev);

/* Test hooks: intermediate products of the last single-frame mslam_orb_extract on this
 * context, so that parity can be checked stage by stage (pyramid level and its 7x7
 * Gaussian-blurred twin, tightly packed level_w*level_h bytes). */
int mslam_orb_debug_level(mslam_ctx* ctx, int level, int blurred, uint8_t* out, int* level_w, int* level_h);

/* ---- R7: brute-force Hamming kNN-2 + ratio test ------------------------------------
 * Stands in for `BFMatcher(NORM_HAMMING,false).knnMatch(d1, d2, raw, 2)` and the ratio
 * loop of matchFeatures (src/CameraPoseEstimator.cpp:200-213).
 * q: nq x 32 bytes, t: nt x 32 bytes.  idx/dist: nq x 2 (best, second).  Ties go to the
 * lower train index.  nt < 2 -> MSLAM_ERR_TOO_FEW. */
int mslam_match_knn2(mslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                     int32_t* idx, int32_t* dist);
/* keep query i iff (float)d0 < (float)d1 * ratio; output in query order */
int mslam_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, float ratio,
                       int32_t* query_idx, int32_t* train_idx, float* d, int* n_out);

/* Batched, device-resident: problem b matches q + b*cap*32 (nq[b] rows) against
 * t + b*cap*32 (nt[b] rows); nq/nt are device int32[B].  idx/dist: B*cap*2.
 * If ratio > 0 the ratio test is applied on device and the surviving matches are
 * written compacted in query order to m_query/m_train (B*cap int32) and m_dist (B*cap
 * float) with the count in m_count[b]; pass ratio <= 0 and NULLs to skip.
 * A problem with nt[b] < 2 yields m_count[b] = 0 and idx = -1. */
int mslam_match_knn2_batch_dev(mslam_ctx* ctx, const uint8_t* q, const int32_t* nq,
                               const uint8_t* t, const int32_t* nt, int cap, int B,
                               int32_t* idx, int32_t* dist, float ratio,
                               int32_t* m_query, int32_t* m_train, float* m_dist, int32_t* m_count);

/* ---- R8/R9: eight-point fundamental-matrix RANSAC -----------------------------------
 * Stands in for computeFundamentalMatrix (src/CameraPoseEstimator.cpp:545-586:
 * findFundamentalMat(CV_FM_RANSAC, 3.0, 0.85) + CV_FM_8POINT refit on the inliers) and
 * is the literal restatement of the in-tree eight-point RANSAC computeFundamentalMatrix2
 * / fundamentalMatrix / getInlier / getRandomSample (:596-786) with its index-width and
 * second-image bugs fixed.  The sample table is an explicit input (H x 8 indices into
 * the M pairs) so that the scored hypotheses are reproducible. */
typedef struct {
    int metric;         /* 0: Sampson distance on coordinates divided by max coordinate,
                              as getInlier (:715-763), inlier iff d < threshold (0.1)
                           1: max of the two squared point-to-epipolar-line distances in
                              pixels, as OpenCV's RANSAC, inlier iff err <= threshold^2 (3.0) */
    double threshold;
    double confidence;  /* in (0,1): OpenCV-style adaptive stop emulated over the ordered
                           hypotheses; <= 0 or >= 1: every hypothesis counts */
    int refit;          /* 0: none (F of the best hypothesis)
                           1: un-normalised refit on all inliers (fundamentalMatrix, :665)
                           2: Hartley-normalised 8-point on the inliers (CV_FM_8POINT, :585) */
} mslam_fmat_params;
void mslam_fmat_default_params(mslam_fmat_params* p); /* metric 1, 3.0, 0.85, refit 2 */

/* x1,x2: M x 2 doubles (pixel coordinates in image 1 / image 2); samples: H x 8.
 * F: row-major 3x3 with x2^T F x1 = 0.  status: M bytes (1 = inlier of the best
 * hypothesis).  inlier_counts (optional): H.  best_h: winning hypothesis (max count,
 * lowest index on ties, within the confidence-limited prefix). */
int mslam_fmat_ransac8(mslam_ctx* ctx, const double* x1, const double* x2, int M,
                       const int32_t* samples, int H, const mslam_fmat_params* params,
                       double F[9], uint8_t* status, int32_t* inlier_counts, int* best_h,
                       int* n_inliers);
/* Batched, device-resident: problem b has M[b] (device int32) pairs at x1 + b*capM*2 ...;
 * every problem uses H hypotheses from samples + b*H*8.  Outputs per problem. */
int mslam_fmat_ransac8_batch_dev(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* M,
                                 int capM, const int32_t* samples, int H, int B,
                                 const mslam_fmat_params* params, double* F, uint8_t* status,
                                 int32_t* inlier_counts, int32_t* best_h, int32_t* n_inliers);

/* ---- R8 as the reference's live path runs it: findFundamentalMat(CV_FM_RANSAC, 3.0, 0.85) (src/CameraPoseEstimator.cpp:563)
 * OpenCV 2.4.13's estimator with 7-point minimal samples (CvFMEstimator::run7Point: the null space of the 7 x 9 system is two-
 * dimensional, det(lambda f1 + (1 - lambda) f2) = 0 gives up to three matrices per sample, each scaled to F[8] = 1), every
 * matrix scored by computeReprojError (metric 1 above), and cvFindFundamentalMat's choice of estimator by the number of pairs:
 * runRANSAC for M >= 15 (record + adaptive iteration bound, tested once per sample), runLMeDS for 8 <= M < 15 (least median
 * of the errors over cvRound(log(1 - confidence) / log(1 - 0.55^7)) samples, at least 3, at most H; inlier threshold
 * sigma = 2.5 * 1.4826 * (1 + 5 / (M - 7)) * sqrt(median), at least 0.001; fewer than 7 inliers = failure), and the direct
 * solution for M == 7 (first matrix of sample 0, every pair an inlier).  OpenCV is absent: restated from its published
 * algorithm, parity unpinned; the null space comes from Gauss-Jordan elimination instead of an SVD and the cubic from a Newton
 * iteration + deflation instead of acos / cos / cbrt (identical operation for operation in the CPU oracle), roots in
 * cvSolveCubic's order.
 * samples: H x 7 indices.  method: 0 = cvFindFundamentalMat's choice (above), 1 = runRANSAC, 2 = runLMeDS (M <= 32 only).
 * params: metric must be 1; threshold / confidence as above (LMedS reads a confidence outside (0,1) as 0.99, like OpenCV);
 * refit 0 = the winning matrix, 2 = CV_FM_8POINT on its inliers when there are at least 8 (the reference's second call, :585).
 * inlier_counts (optional): H x 3, zero where a sample yields fewer matrices and for problems RANSAC does not run on.
 * best_model = 3 * sample + root of the winner, -1 (and MSLAM_ERR_DEGENERATE from the host form) if there is none. */
int mslam_fmat_ransac7(mslam_ctx* ctx, const double* x1, const double* x2, int M,
                       const int32_t* samples, int H, const mslam_fmat_params* params, int method,
                       double F[9], uint8_t* status, int32_t* inlier_counts, int* best_model,
                       int* n_inliers);
int mslam_fmat_ransac7_batch_dev(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* M,
                                 int capM, const int32_t* samples, int H, int B,
                                 const mslam_fmat_params* params, int method, double* F, uint8_t* status,
                                 int32_t* inlier_counts, int32_t* best_model, int32_t* n_inliers);

/* ---- R12: PnP RANSAC ------------------------------------------------------------------
 * Stands in for `solvePnPRansac(mapPoints, imagePoints, curK, distCoeff, rvec, tvec)`
 * + Rodrigues (src/CameraPoseEstimator.cpp:472-473).  OpenCV 2.4's implementation is
 * neither reproducible (threaded, shared RNG) nor available offline; this library keeps
 * its contract (inputs, inlier rule ||project(X) - x|| < reproj_thr with the distortion
 * model k1,k2,p1,p2,k3, outputs rvec/tvec) with a deterministic core: hypothesis h is
 * the P3P pose of sample points 0..2 disambiguated by sample point 3; the best hypothesis
 * is refined by Gauss-Newton on its inliers. */
typedef struct {
    double reproj_thr;   /* 8.0 (OpenCV 2.4 default); BASELINE config C2 uses 2.0 */
    double confidence;   /* as mslam_fmat_params; <= 0 -> off */
    int refine_iters;    /* Gauss-Newton iterations on the inlier set (10); 0 = none */
    int min_inliers;     /* fewer inliers of the winner -> MSLAM_ERR_DEGENERATE (default 4; OpenCV keeps any winner with at least
                            its 4 sample points) */
    int stop_inliers;    /* OpenCV 2.4's minInliersCount (its default: 100): the scan over the ordered hypotheses ends at the
                            first one whose inlier set holds at least this many points (PnPSolver::operator()'s early exit), so
                            that one wins although a later one may count more; <= 0 (default): off, the best of all H wins */
} mslam_pnp_params;
void mslam_pnp_default_params(mslam_pnp_params* p);

/* X: P x 3 world points, x: P x 2 pixels, K: row-major 3x3, dist: 8 coefficients
 * (k1,k2,p1,p2,k3,k4,k5,k6; only the first five are used, as the reference sets,
 * :462-469), samples: H x 4 indices.  rvec/tvec: world->camera.  inliers: P bytes. */
int mslam_pnp_ransac(mslam_ctx* ctx, const double* X, const double* x, int P, const double K[9],
                     const double dist[8], const int32_t* samples, int H,
                     const mslam_pnp_params* params, double rvec[3], double tvec[3],
                     double Rt[12], uint8_t* inliers, int32_t* inlier_counts, int* best_h,
                     int* n_inliers);
/* Batched, device-resident; problem b: P[b] points at X + b*capP*3, x + b*capP*2,
 * K + b*9, dist + b*8, samples + b*H*4.  Outputs Rt (B x 12 row-major 3x4), rvec/tvec
 * (B x 3, optional), inliers (B x capP), inlier_counts (B x H, optional), best_h, n_inliers. */
int mslam_pnp_ransac_batch_dev(mslam_ctx* ctx, const double* X, const double* x, const int32_t* P,
                               int capP, const double* K, const double* dist,
                               const int32_t* samples, int H, int B, const mslam_pnp_params* params,
                               double* Rt, double* rvec, double* tvec, uint8_t* inliers,
                               int32_t* inlier_counts, int32_t* best_h, int32_t* n_inliers);

/* ---- R11: two-view DLT triangulation --------------------------------------------------
 * Stands in for TriangulateMultiplePointsFromTwoView / TriangulateSinglePointFromTwoView
 * (src/CameraPoseEstimator.cpp:86-152).  Rt1,Rt2: row-major 3x4; K1,K2: 3x3.
 * X: n x 3.  front (optional): n bytes, 1 iff z > 0 in both cameras.  n_front optional. */
int mslam_triangulate(mslam_ctx* ctx, const double* x1, const double* x2, int n,
                      const double Rt1[12], const double Rt2[12], const double K1[9], const double K2[9],
                      double* X, uint8_t* front, int* n_front);

/* ---- device-side preparation of batched RANSAC inputs (what the stage classes otherwise assemble on the host) ----------
 * RANSAC in the reference draws its samples inside OpenCV (cv::findFundamentalMat / cv::solvePnPRansac,
 * src/CameraPoseEstimator.cpp:563, :472); here the sample tables are explicit inputs, and this is the stated generator
 * the stages use for them (monocular_slam_amd/host/Geometry.h geom::sampleTable, restated in tests/ref_pipeline.py):
 * table b = H rows of k distinct indices in [0, n[b]); row h is drawn by xorshift64* started from
 * splitmix64(seeds[b] + (h + 1) * 0x9E3779B97F4A7C15) (a zero state is replaced by that constant), each index
 * (next() >> 32) % n[b], repeats within the row redrawn.  Rows are independent, so one thread draws one row.
 * n, seeds: device arrays of B; tables: device, B x H x k.  n[b] < k gives a zero table.  1 <= k <= 8. */
int mslam_sample_tables_dev(mslam_ctx* ctx, const int32_t* n, const uint64_t* seeds, int H, int k, int B, int32_t* tables);
/* Matched pixel coordinates for mslam_fmat_ransac8_batch_dev from device-resident keypoints and match lists
 * (the gather of computeFundamentalMatrix, src/CameraPoseEstimator.cpp:554-560): for i < count[b]
 *   x1[b][i] = (double) kps_q[b][query[b][i]].(x, y),   x2[b][i] = (double) kps_t[b][train[b][i]].(x, y)
 * (float -> double widening, as Point2f -> Point2d in the reference); kps_*: B x cap keypoints, query/train: B x cap,
 * x1/x2: B x cap x 2.  M_out[b] = count[b] if count[b] >= min_count else 0 (a problem the RANSAC kernels skip). */
int mslam_gather_matches_dev(mslam_ctx* ctx, const mslam_keypoint* kps_q, const mslam_keypoint* kps_t, int cap, const int32_t* query,
                             const int32_t* train, const int32_t* count, int B, int min_count, double* x1, double* x2, int32_t* M_out);

/* G independent groups in one launch sequence (host pointers, synchronous): group g holds points [ofs[g], ofs[g+1]) of
 * x1/x2/X/front (ofs[0] = 0) and its own cameras Rt1/Rt2 (G x 12), K1/K2 (G x 9); n_front: G counts (optional).  Point for
 * point the same arithmetic as mslam_triangulate.  What the batched stages use for the four-candidate cheirality vote
 * (:337-354) and the new-point triangulation (:488-512) of B streams at once. */
int mslam_triangulate_batch(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* ofs, int G,
                            const double* Rt1, const double* Rt2, const double* K1, const double* K2,
                            double* X, uint8_t* front, int32_t* n_front);

/* ---- R14-R17: bundle-adjustment linearisation ----------------------------------------
 * Stands in for one pass of g2o's BlockSolver<6,3>::buildSystem
 * (ThirdParty/g2o/g2o/core/block_solver.hpp:502-560) over the graph that
 * Util::BundleAdjustment builds (src/Util.cpp:34-170): per edge
 * EdgeSE3ProjectXYZ::computeError (types/types_six_dof_expmap.h:90-95), ::linearizeOplus
 * (types_six_dof_expmap.cpp:103-139), Huber weighting (core/robust_kernel_impl.cpp:78-91)
 * and BaseBinaryEdge::constructQuadraticForm (core/base_binary_edge.hpp:55-120). */
typedef struct {
    int n_poses, n_points, n_edges;
    const double* poses;        /* n_poses x 7: tx,ty,tz,qx,qy,qz,qw (world->camera, unit q, w >= 0) */
    const uint8_t* pose_fixed;  /* n_poses: 1 = fixed (frame id 0 in the reference, src/Util.cpp:72) */
    const double* points;       /* n_points x 3 */
    const int32_t* edge_pose;   /* n_edges */
    const int32_t* edge_point;  /* n_edges */
    const double* obs;          /* n_edges x 2 */
    const double* info;         /* n_edges: Omega = info * I2 (src/Util.cpp:151-153) */
    const double* intr;         /* n_poses x 4: fx,fy,cx,cy (src/Util.cpp:161-164) */
    double huber_delta;         /* <= 0: no robust kernel */
} mslam_ba_problem;

/* Layout of the linear system (what the host-side Schur/LDLt solve consumes):
 *  free poses are numbered in pose order skipping fixed ones (g2o index mapping,
 *  core/sparse_optimizer.cpp:166-190); fp = number of free poses.
 *  Hpp: fp x 36 (row-major 6x6, rotation (omega) first then translation (upsilon)).
 *  Hll: n_points x 9 (row-major 3x3).
 *  Hpl: block-CSR, rows = free poses, columns = points, sorted by column inside a row;
 *       Hpl_rowptr fp+1, Hpl_col nnzb, Hpl_val nnzb x 18 (row-major 6x3 = B^T w Omega A).
 *  b: 6*fp + 3*n_points (poses first, then points; b = -J^T w Omega e).
 *  err: n_edges x 2, chi2: n_edges (e^T Omega e), rho1: n_edges (Huber weight rho'). */
typedef struct {
    double* err;
    double* chi2;
    double* rho1;
    double* Hpp;
    double* Hll;
    int32_t* Hpl_rowptr;
    int32_t* Hpl_col;
    double* Hpl_val;
    double* b;
    double robust_chi2;   /* sum of rho(chi2) (core/sparse_optimizer.cpp:100-114) */
    int n_free_poses;
    int nnzb;
} mslam_ba_system;

/* Host-pointer, one-shot: analyses the structure and builds the system. The caller
 * sizes the outputs with mslam_ba_sizes (nnzb = edges whose pose is free). */
int mslam_ba_sizes(const mslam_ba_problem* p, int* n_free_poses, int* nnzb);
int mslam_ba_build(mslam_ctx* ctx, const mslam_ba_problem* p, mslam_ba_system* s);

/* Persistent form for the LM loop: the structure (index mapping, CSR, edge orders) is
 * analysed once (g2o: BlockSolver::buildStructure, block_solver.hpp:143-295) and kept on
 * the device; each relinearisation only uploads/reads state. */
typedef struct mslam_ba_plan mslam_ba_plan;
int mslam_ba_plan_create(mslam_ctx* ctx, const mslam_ba_problem* p, mslam_ba_plan** plan);
void mslam_ba_plan_destroy(mslam_ba_plan* plan);
/* device views of the plan's state and of the system it builds (all device pointers) */
typedef struct {
    double* poses; double* points;         /* state: n_poses x 7, n_points x 3 */
    double* err; double* chi2; double* rho1;  /* err, chi2: per edge, in the plan's ROW ORDER (edges sorted by pose row, then point:
                                              the order of the Hpl blocks); rho1: NULL -- the Huber weight is a function of chi2
                                              and is not streamed out; mslam_ba_plan_download returns all three in the caller's
                                              edge order */
    double* Hpp; double* Hll; double* Hpl_val; double* b;
    int32_t* Hpl_rowptr; int32_t* Hpl_col;
    double* robust_chi2;                   /* 1 double */
    int n_free_poses, nnzb;
} mslam_ba_dev_view;
int mslam_ba_plan_view(mslam_ba_plan* plan, mslam_ba_dev_view* view);
int mslam_ba_plan_set_state(mslam_ba_plan* plan, const double* poses_host, const double* points_host);
/* Edge weights and robust kernel of an existing plan (structure unchanged): info_host (n_edges, optional) replaces
 * the per-edge information scale -- 0 takes an edge out of the optimisation, which is what g2o's setLevel(1) +
 * initializeOptimization(0) does in Util::PoseBundleAdjustment's inlier/outlier rounds (src/Util.cpp:305-339);
 * huber_delta <= 0 removes the robust kernel (setRobustKernel(NULL), :335). */
int mslam_ba_plan_set_robust(mslam_ba_plan* plan, const double* info_host, double huber_delta);
/* enqueue one linearisation on the context's stream (no sync) */
int mslam_ba_plan_build_dev(mslam_ba_plan* plan);
/* copy the built system to host buffers (syncs) */
int mslam_ba_plan_download(mslam_ba_plan* plan, mslam_ba_system* s);

/* Schur complement of the built system on the device (g2o: BlockSolver::solve,
 * block_solver.hpp:372-481): with lambda added to every pose and landmark diagonal (setLambda, :564-589),
 *   S   = Hpp - sum_i Hpl_i Hll_i^-1 Hpl_i^T      (6 fp x 6 fp, dense, row-major, both triangles)
 *   b_s = b_p - sum_i Hpl_i Hll_i^-1 b_l,i
 * The caller solves S x_p = b_s (small, host), then mslam_ba_plan_backsub_dev computes x_l = Hll^-1 (b_l - Hpl^T x_p)
 * on the device, optionally applies X += x_l to the plan's points, and reduces sum x_l (lambda x_l + b_l)
 * (the landmark part of LM's computeScale).  mslam_ba_plan_points_snapshot_dev(restore=0/1) saves / restores the
 * points for rejected LM steps.  MSLAM_ERR_CAPACITY if the plan was built without Schur scratch (a tile of 64
 * consecutive points observed by more than 64 different free poses: solve that system on the host from the blocks).
 * A graph with two edges between the same pose and point is refused by mslam_ba_plan_create (MSLAM_ERR_INVALID): g2o
 * would accumulate them into one block, this layout gives every edge its own. */
int mslam_ba_plan_schur_dev(mslam_ba_plan* plan, double lambda);
int mslam_ba_plan_download_schur(mslam_ba_plan* plan, double* S, double* b_s);
int mslam_ba_plan_backsub_dev(mslam_ba_plan* plan, const double* x_pose_host, int apply_to_points);
int mslam_ba_plan_download_x(mslam_ba_plan* plan, double* x /* 6 fp + 3 n_points, optional */, double* scale_landmarks /* optional */);
int mslam_ba_plan_points_snapshot_dev(mslam_ba_plan* plan, int restore);
int mslam_ba_plan_get_state(mslam_ba_plan* plan, double* poses_host, double* points_host);

/* Windows: several independent local-BA problems concatenated block-diagonally into one plan (what a multi-stream host
 * does with the sliding windows of its streams; the reference optimises one graph at a time, src/Optimiser.cpp:6-18).
 * pose_window (n_poses) / point_window (n_points) name the window of every vertex (fixed poses: any valid id).  After
 * this call lambda is per window: mslam_ba_plan_schur_windows_dev takes n_windows values (mslam_ba_plan_schur_dev keeps
 * working and uses one value for all), every build additionally reduces the robust chi2 per window and every
 * back-substitution the landmark part of LM's computeScale per window (mslam_ba_plan_download_window_sums), so that one
 * Levenberg-Marquardt trial of every window runs per launch sequence. */
int mslam_ba_plan_set_windows(mslam_ba_plan* plan, const int32_t* pose_window, const int32_t* point_window, int n_windows);
int mslam_ba_plan_schur_windows_dev(mslam_ba_plan* plan, const double* lambda_per_window_host);
int mslam_ba_plan_download_window_sums(mslam_ba_plan* plan, double* robust_chi2 /* n_windows, optional */,
                                       double* scale_landmarks /* n_windows, optional */);

/* ---- multi-GPU -------------------------------------------------------------------------
 * The path shards by independent streams (one process per GPU, no data-path collective).
 * The only exchange is the optional sharding of ONE problem's hypotheses over ranks (the
 * solvePnPRansac / findFundamentalMat call sites, src/CameraPoseEstimator.cpp:472,563):
 * every rank scores its slice and the int32 inlier counts are all-gathered over RCCL
 * (xGMI inside a node; <= 16 KiB, latency-bound); the global winner is then
 * mslam_argmax_counts over the gathered vector (max count, lowest index on ties).
 * The reference has no counterpart (single process). */
int mslam_argmax_counts(const int32_t* counts, int n, int* best, int* best_count);

/* One process per GPU.  Rank 0 obtains the 128-byte id and the host side hands it to the other ranks (file, socket,
 * MPI, torch.distributed store ...); every rank then creates its communicator on its own context.  RCCL is bound
 * lazily (dlopen librccl.so.1): MSLAM_ERR_NO_DEVICE if it cannot be loaded, MSLAM_ERR_HIP (+ mslam_last_hip_error) if
 * an RCCL call fails. */
#define MSLAM_COMM_ID_BYTES 128
typedef struct mslam_comm mslam_comm;
/* Binds RCCL now (optional; the first communicator does it otherwise): the librccl.so.1 beside the libamdhip64 this library
 * is linked to, by absolute path -- a process that also holds another ROCm stack under the same SONAMEs (a PyTorch wheel)
 * should call this BEFORE loading that stack.  path_out (optional) receives the file that was bound.
 * The environment variable MSLAM_RCCL_PATH replaces the search by one explicit file. */
int mslam_comm_preload(char* path_out, int cap);
int mslam_comm_unique_id(uint8_t id[MSLAM_COMM_ID_BYTES]);
int mslam_comm_create(mslam_ctx* ctx, const uint8_t id[MSLAM_COMM_ID_BYTES], int rank, int nranks, mslam_comm** out);
int mslam_comm_info(mslam_comm* comm, int* rank, int* nranks, int* rccl_version);
/* recv[r*n .. r*n+n) = rank r's send.  _dev: device pointers, enqueued on the context's stream (ordered behind the
 * scoring kernels, no host sync); the un-suffixed form takes host pointers and synchronises. */
int mslam_comm_allgather_i32_dev(mslam_comm* comm, const int32_t* send_dev, int n, int32_t* recv_dev);
int mslam_comm_allgather_i32(mslam_comm* comm, const int32_t* send_host, int n, int32_t* recv_host);
int mslam_comm_destroy(mslam_comm* comm);
/* One process driving ngpu contexts on distinct devices: rank g contributes send + g*n (host);
 * recv (host, ngpu*n) receives the gathered vector.  Communicators are created on first use per device list and cached
 * for the life of the process (calls are serialised by a library lock; on an error every stream of the call is drained
 * before it returns).  mslam_comm_shutdown destroys the cached communicators: call it with the contexts' streams idle,
 * before the contexts are destroyed; optional. */
int mslam_allgather_i32(mslam_ctx* const* ctxs, int ngpu, const int32_t* send, int n, int32_t* recv);
int mslam_comm_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* MSLAM_H */
