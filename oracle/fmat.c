/*
 * fmat.c -- oracle (test infrastructure, see oracle.h): fundamental matrix RANSAC,
 * essential-matrix pose recovery and two-view triangulation.
 *
 * Follows reference src/CameraPoseEstimator.cpp:
 *   computeFundamentalMatrix2 / fundamentalMatrix / getInlier / getRandomSample (:596-786)
 *     -- the in-tree eight-point RANSAC, restated with two of its bugs fixed: indices are
 *        int32 (the reference stores them in vector<uchar>, :593-594,:641,:728,:769), and
 *        the second point set is read from positions2 (the reference reads positions1,
 *        :619-620).  The sample table is an input (the reference draws rand() % n, :779).
 *   computeFundamentalMatrix (:545-586) -- findFundamentalMat(CV_FM_RANSAC, 3.0, 0.85)
 *        then CV_FM_8POINT on the inliers; OpenCV 2.4.13 is absent, so its error measure
 *        (max of the two squared point-to-epipolar-line distances, cast to float, inlier
 *        iff <= thr^2), its adaptive iteration bound (cvRANSACUpdateNumIters) and its
 *        Hartley-normalised 8-point refit are restated from the published algorithm, while
 *        the hypotheses come from the eight-point solver above instead of OpenCV's
 *        internal-RNG 7-point solver.  PARITY UNPINNED (no golden vectors exist).
 *   computeEssentialMatrix (:180-182), ExtractRTfromE (:154-174), the det(R1) == -1 sign
 *        fix (:328-331), the four-candidate cheirality vote (:337-354)
 *   TriangulateSinglePointFromTwoView / ...MultiplePoints... (:86-152), solveHLS
 *        (src/CommonMath.cpp:17-22)
 */
#include "oracle.h"
#include <float.h>
#if defined(__x86_64__) && defined(__GNUC__)
#define ORC_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_FMA_CLONES
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_fmat_default_params(orc_fmat_params* p)
{
    p->metric = 1; p->threshold = 3.0; p->confidence = 0.85; p->refit = 2; /* src/ParamConfig.h:24-25 */
}

static void rank2(double F[9])
{
    /* F <- U diag(s1,s2,0) Vt == F - (F v3) v3^T, v3 = right singular vector of the smallest singular value */
    double FtF[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += F[k * 3 + i] * F[k * 3 + j];
            FtF[i * 3 + j] = acc;
        }
    orc_jacobi_eig(FtF, 3, w, V);
    double v3[3] = { V[0], V[3], V[6] };
    for (int r = 0; r < 3; r++) {
        double fv = F[r * 3] * v3[0] + F[r * 3 + 1] * v3[1] + F[r * 3 + 2] * v3[2];
        for (int c = 0; c < 3; c++) F[r * 3 + c] = F[r * 3 + c] - fv * v3[c];
    }
}

/* fundamentalMatrix (:672-713) on the rows `sel`: A rows (u2u1,u2v1,u2,v2u1,v2v1,v2,u1,v1,1)
 * with coordinates divided by maxt and stored as float (Point2f, :615-633; the products are
 * float products, :678-685); null vector of A; rank-2 enforcement. */
void orc_fmat_from_rows(const double* x1, const double* x2, const int32_t* sel, int n, double maxt, double F[9])
{
    double AtA[81], w[9], V[81], A8[72];
    memset(AtA, 0, sizeof(AtA));
    for (int k = 0; k < n; k++) {
        int i = sel ? sel[k] : k;
        float u1 = (float)(x1[2 * i] / maxt), v1 = (float)(x1[2 * i + 1] / maxt);
        float u2 = (float)(x2[2 * i] / maxt), v2 = (float)(x2[2 * i + 1] / maxt);
        double r[9] = { (double)(u2 * u1), (double)(u2 * v1), (double)u2, (double)(v2 * u1), (double)(v2 * v1),
                        (double)v2, (double)u1, (double)v1, 1.0 };
        if (n == 8) { memcpy(A8 + 9 * k, r, sizeof(r)); continue; }
        for (int a = 0; a < 9; a++)
            for (int b = a; b < 9; b++) AtA[a * 9 + b] += r[a] * r[b];
    }
    if (n == 8) {
        /* minimal sample: A is 8 x 9, its null vector comes from elimination (see orc_null_8x9) */
        orc_null_8x9(A8, F);
    } else {
        for (int a = 0; a < 9; a++)
            for (int b = 0; b < a; b++) AtA[a * 9 + b] = AtA[b * 9 + a];
        orc_jacobi_eig(AtA, 9, w, V);
        for (int a = 0; a < 9; a++) F[a] = V[a * 9 + 0];
    }
    rank2(F);
}

/* run8Point of OpenCV 2.4.13 (CV_FM_8POINT): Hartley normalisation, A^T A eigenvector,
 * rank 2, de-normalise, scale so that F[8] == 1 */
int orc_fmat_8point_normalized(const double* x1, const double* x2, const int32_t* sel, int n, double F[9])
{
    if (n < 8) return -4;
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0;
    for (int k = 0; k < n; k++) {
        int i = sel ? sel[k] : k;
        c0x += x1[2 * i]; c0y += x1[2 * i + 1]; c1x += x2[2 * i]; c1y += x2[2 * i + 1];
    }
    double t = 1. / n;
    c0x *= t; c0y *= t; c1x *= t; c1y *= t;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < n; k++) {
        int i = sel ? sel[k] : k;
        double dx = x1[2 * i] - c0x, dy = x1[2 * i + 1] - c0y;
        s0 += sqrt(dx * dx + dy * dy);
        dx = x2[2 * i] - c1x; dy = x2[2 * i + 1] - c1y;
        s1 += sqrt(dx * dx + dy * dy);
    }
    s0 *= t; s1 *= t;
    if (s0 < FLT_EPSILON || s1 < FLT_EPSILON) return -6;
    s0 = sqrt(2.) / s0; s1 = sqrt(2.) / s1;
    double AtA[81], w[9], V[81];
    memset(AtA, 0, sizeof(AtA));
    for (int k = 0; k < n; k++) {
        int i = sel ? sel[k] : k;
        double X0 = (x1[2 * i] - c0x) * s0, Y0 = (x1[2 * i + 1] - c0y) * s0;
        double X1 = (x2[2 * i] - c1x) * s1, Y1 = (x2[2 * i + 1] - c1y) * s1;
        double r[9] = { X1 * X0, X1 * Y0, X1, Y1 * X0, Y1 * Y0, Y1, X0, Y0, 1 };
        for (int a = 0; a < 9; a++)
            for (int b = 0; b < 9; b++) AtA[a * 9 + b] += r[a] * r[b];
    }
    orc_jacobi_eig(AtA, 9, w, V);
    /* at most one (numerically) zero eigenvalue allowed */
    if (fabs(w[1]) < DBL_EPSILON) return -6;
    double F0[9];
    for (int a = 0; a < 9; a++) F0[a] = V[a * 9 + 0];
    rank2(F0);
    double T0[9] = { s0, 0, -s0 * c0x, 0, s0, -s0 * c0y, 0, 0, 1 };
    double T1[9] = { s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1 };
    double tmp[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { /* T1^T * F0 */
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += T1[k * 3 + i] * F0[k * 3 + j];
            tmp[i * 3 + j] = acc;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += tmp[i * 3 + k] * T0[k * 3 + j];
            F[i * 3 + j] = acc;
        }
    if (fabs(F[8]) > FLT_EPSILON) {
        double sc = 1. / F[8];
        for (int a = 0; a < 9; a++) F[a] *= sc;
    }
    return 0;
}

/* metric 0: getInlier's Sampson distance (:715-763); metric 1: OpenCV computeReprojError, its line products and
 * norms evaluated as explicit fma chains (exact IEEE fma: identical on CPU and GPU) */
ORC_FMA_CLONES
double orc_fmat_error(const double F[9], const double* p1, const double* p2, int metric)
{
    double u1 = p1[0], v1 = p1[1], u2 = p2[0], v2 = p2[1];
    if (metric == 0) {
        double Fx1u = F[0] * u1 + F[1] * v1 + F[2];
        double Fx1v = F[3] * u1 + F[4] * v1 + F[5];
        double Fx1w = F[6] * u1 + F[7] * v1 + F[8];
        double Ftx2u = F[0] * u2 + F[3] * v2 + F[6];
        double Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
        double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
        double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
        return fabs(d);
    }
    double a = fma(F[0], u1, fma(F[1], v1, F[2]));
    double b = fma(F[3], u1, fma(F[4], v1, F[5]));
    double c = fma(F[6], u1, fma(F[7], v1, F[8]));
    double s2 = 1. / fma(a, a, b * b);
    double d2 = fma(u2, a, fma(v2, b, c));
    a = fma(F[0], u2, fma(F[3], v2, F[6]));
    b = fma(F[1], u2, fma(F[4], v2, F[7]));
    c = fma(F[2], u2, fma(F[5], v2, F[8]));
    double s1 = 1. / fma(a, a, b * b);
    double d1 = fma(u1, a, fma(v1, b, c));
    double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (double)(float)(e1 > e2 ? e1 : e2);
}

/* cvRANSACUpdateNumIters (OpenCV 2.4.13 modelest.cpp) */
static int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* CvModelEstimator2::runRANSAC's control flow over the ordered per-hypothesis inlier counts:
 * a hypothesis wins if its count beats the best so far (and sample_size-1); the iteration
 * bound shrinks with the inlier ratio.  confidence outside (0,1) disables the bound. */
int orc_ransac_select(const int32_t* counts, int H, int n_points, int sample_size, double confidence, int* best_count)
{
    int niters = H, best = -1, maxgood = 0;
    int adaptive = confidence > 0. && confidence < 1.;
    for (int it = 0; it < niters; it++) {
        int good = counts[it];
        int floor_ = maxgood > sample_size - 1 ? maxgood : sample_size - 1;
        if (good > floor_) {
            best = it; maxgood = good;
            if (adaptive) niters = ransac_update_iters(confidence, (double)(n_points - good) / n_points, sample_size, niters);
        }
    }
    if (best_count) *best_count = maxgood;
    return best;
}

static void scale_point(const double* x, int i, double maxt, int metric, double out[2])
{
    if (metric == 0) { /* through float, as p_match (:31-34) */
        out[0] = (double)(float)(x[2 * i] / maxt); out[1] = (double)(float)(x[2 * i + 1] / maxt);
    } else { out[0] = x[2 * i]; out[1] = x[2 * i + 1]; }
}

static void denormalize(const double Fs[9], double maxt, double F[9])
{
    /* times_F^T * F * times_F with times_F = diag(1/maxt, 1/maxt, 1) (:666-669) */
    double t[3] = { 1.0 / maxt, 1.0 / maxt, 1.0 };
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) F[i * 3 + j] = t[i] * Fs[i * 3 + j] * t[j];
}

int orc_fmat_ransac8(const double* x1, const double* x2, int M, const int32_t* samples, int H,
                     const orc_fmat_params* prm, double F[9], uint8_t* status, int32_t* inlier_counts,
                     int* best_h, int* n_inliers)
{
    if (M < 8 || H < 1) return -4;
    double maxt = 0;
    for (int i = 0; i < M; i++) { /* :609-613 */
        double m = x1[2 * i] > x2[2 * i] ? x1[2 * i] : x2[2 * i];
        if (m > maxt) maxt = m;
        m = x1[2 * i + 1] > x2[2 * i + 1] ? x1[2 * i + 1] : x2[2 * i + 1];
        if (m > maxt) maxt = m;
    }
    if (!(maxt > 0)) return -6;
    const double thr = prm->metric == 0 ? prm->threshold : prm->threshold * prm->threshold;
    int32_t* counts = inlier_counts ? inlier_counts : (int32_t*)malloc(sizeof(int32_t) * (size_t)H);
    for (int h = 0; h < H; h++) {
        double Fs[9], Fe[9];
        orc_fmat_from_rows(x1, x2, samples + 8 * h, 8, maxt, Fs);
        if (prm->metric == 0) memcpy(Fe, Fs, sizeof(Fe)); else denormalize(Fs, maxt, Fe);
        int c = 0;
        for (int i = 0; i < M; i++) {
            double p1[2], p2[2];
            scale_point(x1, i, maxt, prm->metric, p1);
            scale_point(x2, i, maxt, prm->metric, p2);
            double e = orc_fmat_error(Fe, p1, p2, prm->metric);
            c += prm->metric == 0 ? (e < thr) : (e <= thr);
        }
        counts[h] = c;
    }
    int bc = 0;
    int best = orc_ransac_select(counts, H, M, 8, prm->confidence, &bc);
    if (!inlier_counts) free(counts);
    if (best < 0) return -6;
    double Fs[9], Fe[9];
    orc_fmat_from_rows(x1, x2, samples + 8 * best, 8, maxt, Fs);
    if (prm->metric == 0) memcpy(Fe, Fs, sizeof(Fe)); else denormalize(Fs, maxt, Fe);
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)M);
    int n = 0;
    for (int i = 0; i < M; i++) {
        double p1[2], p2[2];
        scale_point(x1, i, maxt, prm->metric, p1);
        scale_point(x2, i, maxt, prm->metric, p2);
        double e = orc_fmat_error(Fe, p1, p2, prm->metric);
        int in = prm->metric == 0 ? (e < thr) : (e <= thr);
        status[i] = (uint8_t)in;
        if (in) sel[n++] = i;
    }
    int rc = 0;
    if (prm->refit == 0) denormalize(Fs, maxt, F);
    else if (prm->refit == 1) { orc_fmat_from_rows(x1, x2, sel, n, maxt, Fs); denormalize(Fs, maxt, F); }
    else rc = orc_fmat_8point_normalized(x1, x2, sel, n, F);
    free(sel);
    if (best_h) *best_h = best;
    if (n_inliers) *n_inliers = n;
    return rc;
}

/* TriangulateSinglePointFromTwoView (:86-128): DLT rows P(0)-P(2)x, P(1)-P(2)y for both views,
 * X = null vector of the 4x4 system (solveHLS), dehomogenised; front = z > 0 in both cameras */
int orc_triangulate(const double* x1, const double* x2, int n, const double Rt1[12], const double Rt2[12],
                    const double K1[9], const double K2[9], double* X, uint8_t* front, int* n_front)
{
    double P1[12], P2[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; k++) { a += K1[i * 3 + k] * Rt1[k * 4 + j]; b += K2[i * 3 + k] * Rt2[k * 4 + j]; }
            P1[i * 4 + j] = a; P2[i * 4 + j] = b;
        }
    int cnt = 0;
    for (int p = 0; p < n; p++) {
        double A[16];
        for (int j = 0; j < 4; j++) {
            A[0 * 4 + j] = P1[0 * 4 + j] - P1[2 * 4 + j] * x1[2 * p];
            A[1 * 4 + j] = P1[1 * 4 + j] - P1[2 * 4 + j] * x1[2 * p + 1];
            A[2 * 4 + j] = P2[0 * 4 + j] - P2[2 * 4 + j] * x2[2 * p];
            A[3 * 4 + j] = P2[1 * 4 + j] - P2[2 * 4 + j] * x2[2 * p + 1];
        }
        double AtA[16], w[4], V[16];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double acc = 0;
                for (int k = 0; k < 4; k++) acc += A[k * 4 + i] * A[k * 4 + j];
                AtA[i * 4 + j] = acc;
            }
        orc_jacobi_eig(AtA, 4, w, V);
        double Xh[4] = { V[0], V[4], V[8], V[12] };
        double Xp[3] = { Xh[0] / Xh[3], Xh[1] / Xh[3], Xh[2] / Xh[3] };
        X[3 * p] = Xp[0]; X[3 * p + 1] = Xp[1]; X[3 * p + 2] = Xp[2];
        double z1 = Rt1[8] * Xp[0] + Rt1[9] * Xp[1] + Rt1[10] * Xp[2] + Rt1[11];
        double z2 = Rt2[8] * Xp[0] + Rt2[9] * Xp[1] + Rt2[10] * Xp[2] + Rt2[11];
        int f = z1 > 0 && z2 > 0;
        if (front) front[p] = (uint8_t)f;
        cnt += f;
    }
    if (n_front) *n_front = cnt;
    return 0;
}

/* initialPoseEstimation (:316-356): E = K2^T F K1, ExtractRTfromE, sign fix, cheirality vote.
 * Returns the winning candidate 0..3 in the reference's order (R1|t1, R1|t2, R2|t1, R2|t2). */
int orc_pose_from_F(const double F[9], const double K1[9], const double K2[9], const double* x1, const double* x2,
                    int n, double Rt[12], double* X, int* counts4)
{
    double tmp[9], E[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += K2[k * 3 + i] * F[k * 3 + j];
            tmp[i * 3 + j] = acc;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += tmp[i * 3 + k] * K1[k * 3 + j];
            E[i * 3 + j] = acc;
        }
    double U[9], s[3], Vt[9];
    orc_svd3(E, U, s, Vt);
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 };
    double R[2][9];
    for (int c = 0; c < 2; c++) {
        const double* Wm = c == 0 ? W : Wt;
        double UW[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += U[i * 3 + k] * Wm[k * 3 + j];
                UW[i * 3 + j] = acc;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += UW[i * 3 + k] * Vt[k * 3 + j];
                R[c][i * 3 + j] = acc;
            }
    }
    /* `if (abs(determinant(R1) + 1.0) < EPSILON) { E = -E; ExtractRTfromE(...) }` (:328-331): an SVD of -E
     * that keeps U is U S (-V)^T, i.e. both rotations change sign and t = +-u3 is unchanged */
    if (fabs(orc_det3(R[0]) + 1.0) < 1e-3)
        for (int c = 0; c < 2; c++)
            for (int a = 0; a < 9; a++) R[c][a] = -R[c][a];
    double t[2][3] = { { U[2], U[5], U[8] }, { -U[2], -U[5], -U[8] } };
    const double I34[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    int best = -1, maxc = -1;
    double* Xt = (double*)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    for (int c = 0; c < 4; c++) {
        const double* Rc = R[c / 2];
        const double* tc = t[c % 2];
        double cand[12];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) cand[i * 4 + j] = Rc[i * 3 + j]; cand[i * 4 + 3] = tc[i]; }
        int cnt = 0;
        orc_triangulate(x1, x2, n, I34, cand, K1, K2, Xt, NULL, &cnt);
        if (counts4) counts4[c] = cnt;
        if (maxc < cnt) {
            maxc = cnt; best = c;
            memcpy(Rt, cand, sizeof(cand));
            if (X) memcpy(X, Xt, sizeof(double) * 3 * (size_t)n);
        }
    }
    free(Xt);
    return best;
}
