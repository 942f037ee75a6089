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
    return orc_ransac_select_stop(counts, H, n_points, sample_size, confidence, 0, best_count);
}

/* ... with solvePnPRansac's early exit (OpenCV 2.4.13 PnPSolver::operator(): the loop over the iterations ends as soon as the best
 * inlier set holds at least minInliersCount points): stop_count <= 0 switches it off */
int orc_ransac_select_stop(const int32_t* counts, int H, int n_points, int sample_size, double confidence, int stop_count, int* best_count)
{
    int niters = H, best = -1, maxgood = 0;
    int adaptive = confidence > 0. && confidence < 1.;
    for (int it = 0; it < niters; it++) {
        int good = counts[it];
        int floor_ = maxgood > sample_size - 1 ? maxgood : sample_size - 1;
        if (good > floor_) {
            best = it; maxgood = good;
            if (adaptive) niters = ransac_update_iters(confidence, (double)(n_points - good) / n_points, sample_size, niters);
            if (stop_count > 0 && maxgood >= stop_count) break;
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

/* ---- the live path of computeFundamentalMatrix (:563): findFundamentalMat(CV_FM_RANSAC, 3.0, 0.85) ----------------
 * OpenCV 2.4.13 (absent here; restated from its published algorithm, PARITY UNPINNED): 7-point minimal samples
 * (CvFMEstimator::run7Point: the two-dimensional null space of the 7 x 9 system, det(lambda f1 + (1 - lambda) f2) = 0,
 * up to three matrices per sample, each scaled to F[8] = 1), every matrix of a sample scored by computeReprojError,
 * runRANSAC's record/iteration-bound loop when there are at least 15 pairs and runLMeDS (least median of the errors,
 * inlier threshold from the median) below that; exactly 7 pairs are solved directly.
 * Stated differences, none observable in exact arithmetic: the null space comes from Gauss-Jordan elimination with
 * complete pivoting instead of an SVD (any basis of the null space gives the same matrices), the cubic is solved by a
 * monotone Newton iteration plus deflation instead of acos / cos / cbrt (so that CPU and GPU agree operation for
 * operation), with the roots emitted in cvSolveCubic's order for three real roots (smallest, largest, middle). */

/* cofactors: det(A + t B) = det A + t sum(cof(A) . B) + t^2 sum(cof(B) . A) + t^3 det B */
static void cof3(const double A[9], double C[9])
{
    C[0] = A[4] * A[8] - A[5] * A[7]; C[1] = A[5] * A[6] - A[3] * A[8]; C[2] = A[3] * A[7] - A[4] * A[6];
    C[3] = A[2] * A[7] - A[1] * A[8]; C[4] = A[0] * A[8] - A[2] * A[6]; C[5] = A[1] * A[6] - A[0] * A[7];
    C[6] = A[1] * A[5] - A[2] * A[4]; C[7] = A[2] * A[3] - A[0] * A[5]; C[8] = A[0] * A[4] - A[1] * A[3];
}

/* a real root of x^3 + a x^2 + b x + c (the same monotone Newton iteration as the P3P solver's) */
static double cubic_root_m(double a, double b, double c)
{
    double xi = -a / 3.0;
    double pi = ((xi + a) * xi + b) * xi + c;
    if (pi == 0.0) return xi;
    double m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
    m = m > fabs(c) ? m : fabs(c);
    double x = pi > 0.0 ? -(1.0 + m) : (1.0 + m);
    for (int it = 0; it < 100; it++) {
        double p = ((x + a) * x + b) * x + c;
        double dp = (3.0 * x + 2.0 * a) * x + b;
        if (dp == 0.0) break;
        double xn = x - p / dp;
        double dx = fabs(xn - x);
        x = xn;
        if (dx <= 1e-15 * fabs(xn) + 1e-300) break;
    }
    return x;
}

/* real roots of c3 x^3 + c2 x^2 + c1 x + c0, in cvSolveCubic's order; returns their number (0..3) */
int orc_solve_cubic(double c3, double c2, double c1, double c0, double r[3])
{
    if (c3 == 0.0) {
        if (c2 == 0.0) {
            if (c1 == 0.0) return 0;
            r[0] = -c0 / c1;
            return 1;
        }
        double d = c1 * c1 - 4.0 * c2 * c0;
        if (!(d >= 0.0)) return 0;
        d = sqrt(d);
        double q = 1.0 / (2.0 * c2);
        r[0] = (-c1 + d) * q;
        r[1] = (-c1 - d) * q;
        return 2;
    }
    const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
    const double x0 = cubic_root_m(a, b, c);
    /* the deflated quadratic x^2 + p x + q: forward division when x0 is the small root, backward when it is the large one
     * (each is the stable direction; the other loses the small coefficients to cancellation) */
    double p, q;
    if (fabs(x0) * x0 * x0 > fabs(c)) { q = -c / x0; p = (q - b) / x0; }
    else { p = a + x0; q = b + p * x0; }
    const double disc = p * p - 4.0 * q;
    if (!(disc >= 0.0)) { r[0] = x0; return 1; }
    const double sq = sqrt(disc);
    const double h = -0.5 * (p + (p >= 0.0 ? sq : -sq));
    double x12[2] = { h, h != 0.0 ? q / h : 0.0 };
    for (int k = 0; k < 2; k++)   /* three Newton steps on the cubic itself */
        for (int it = 0; it < 3; it++) {
            const double x = x12[k];
            const double pv = ((x + a) * x + b) * x + c;
            const double dp = (3.0 * x + 2.0 * a) * x + b;
            if (dp == 0.0) break;
            x12[k] = x - pv / dp;
        }
    double x1 = x12[0], x2 = x12[1];
    /* ascending, then (smallest, largest, middle) */
    double lo = x0, mid = x1, hi = x2, t;
    if (mid < lo) { t = lo; lo = mid; mid = t; }
    if (hi < lo) { t = lo; lo = hi; hi = t; }
    if (hi < mid) { t = mid; mid = hi; hi = t; }
    r[0] = lo; r[1] = hi; r[2] = mid;
    return 3;
}

/* run7Point on the pairs sel[0..6] (pixel coordinates): up to three matrices (row-major, 9 doubles each) */
int orc_fmat_7point(const double* x1, const double* x2, const int32_t* sel, double F[27])
{
    double A[63];
    for (int k = 0; k < 7; k++) {
        const int i = sel ? sel[k] : k;
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* r = A + 9 * k;
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2; r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2; r[6] = u1; r[7] = v1; r[8] = 1.0;
    }
    int used_r[7] = { 0 }, used_c[9] = { 0 }, prow[7], pcol[7];
    for (int k = 0; k < 7; k++) {
        double best = -1.0;
        int br = 0, bc = 0;
        for (int r = 0; r < 7; r++) {
            if (used_r[r]) continue;
            for (int c = 0; c < 9; c++) {
                if (used_c[c]) continue;
                const double v = fabs(A[r * 9 + c]);
                if (v > best) { best = v; br = r; bc = c; }
            }
        }
        used_r[br] = 1; used_c[bc] = 1; prow[k] = br; pcol[k] = bc;
        const double pv = A[br * 9 + bc];
        if (pv == 0.0) return 0;
        for (int r = 0; r < 7; r++) {
            if (r == br) continue;
            const double m = A[r * 9 + bc] / pv;
            for (int c = 0; c < 9; c++) A[r * 9 + c] = (c == bc) ? 0.0 : A[r * 9 + c] - m * A[br * 9 + c];
        }
    }
    int cf[2], nf = 0;
    for (int c = 0; c < 9; c++) if (!used_c[c]) cf[nf++] = c;
    double f[2][9];
    for (int v = 0; v < 2; v++) {
        for (int c = 0; c < 9; c++) f[v][c] = 0.0;
        f[v][cf[v]] = 1.0;
        for (int k = 0; k < 7; k++) f[v][pcol[k]] = -A[prow[k] * 9 + cf[v]] / A[prow[k] * 9 + pcol[k]];
        double nn = 0.0;
        for (int c = 0; c < 9; c++) nn += f[v][c] * f[v][c];
        nn = sqrt(nn);
        for (int c = 0; c < 9; c++) f[v][c] = f[v][c] / nn;
    }
    /* F(lambda) = lambda G + f2 with G = f1 - f2 */
    double G[9], CG[9], C2[9];
    const double* f2 = f[1];
    for (int c = 0; c < 9; c++) G[c] = f[0][c] - f2[c];
    cof3(G, CG); cof3(f2, C2);
    double c3 = G[0] * CG[0] + G[1] * CG[1] + G[2] * CG[2];
    double c0 = f2[0] * C2[0] + f2[1] * C2[1] + f2[2] * C2[2];
    double c2 = 0.0, c1 = 0.0;
    for (int k = 0; k < 9; k++) { c2 += CG[k] * f2[k]; c1 += C2[k] * G[k]; }
    double roots[3];
    const int n = orc_solve_cubic(c3, c2, c1, c0, roots);
    for (int k = 0; k < n; k++) {
        double lambda = roots[k], mu = 1.0;
        const double s = G[8] * roots[k] + f2[8];
        double* Fk = F + 9 * k;
        if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda = lambda * mu; Fk[8] = 1.0; }
        else Fk[8] = 0.0;
        for (int c = 0; c < 8; c++) Fk[c] = G[c] * lambda + f2[c] * mu;
    }
    return n;
}

static int cmp_float(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* method: 0 = as cvFindFundamentalMat chooses (7 pairs: direct; >= 15: RANSAC; else LMedS), 1 = RANSAC, 2 = LMedS.
 * samples: H x 7.  inlier_counts (optional): H x 3, RANSAC only (0 where a sample has fewer matrices).
 * best_model = 3 * sample + root of the winner. */
int orc_fmat_ransac7(const double* x1, const double* x2, int M, const int32_t* samples, int H,
                     const orc_fmat_params* prm, int method, double F[9], uint8_t* status, int32_t* inlier_counts,
                     int* best_model, int* n_inliers)
{
    if (M < 7 || H < 1) return -4;
    if (method == 0) method = M == 7 ? 3 : (M >= 15 ? 1 : 2);
    else if (M == 7) method = 3;
    double Fb[9] = { 0 };
    int best = -1;
    double thr = prm->threshold * prm->threshold;
    double* models = (double*)malloc(sizeof(double) * 27 * (size_t)H);
    int* nm = (int*)malloc(sizeof(int) * (size_t)H);
    for (int h = 0; h < H; h++) {
        int ok = 1;
        for (int k = 0; k < 7; k++) if (samples[7 * h + k] < 0 || samples[7 * h + k] >= M) ok = 0;
        nm[h] = ok ? orc_fmat_7point(x1, x2, samples + 7 * h, models + 27 * h) : 0;
    }
    if (method == 3) { /* exactly 7 pairs: the first matrix of sample 0, every pair is an inlier */
        if (nm[0] > 0) { best = 0; thr = INFINITY; }
    } else if (method == 1) {
        int32_t* counts = inlier_counts ? inlier_counts : (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)H);
        for (int h = 0; h < H; h++)
            for (int k = 0; k < 3; k++) {
                int c = 0;
                if (k < nm[h])
                    for (int i = 0; i < M; i++) c += orc_fmat_error(models + 27 * h + 9 * k, x1 + 2 * i, x2 + 2 * i, 1) <= thr;
                counts[3 * h + k] = c;
            }
        /* runRANSAC: the bound is tested once per sample, every matrix of the sample is a candidate */
        int niters = H, maxgood = 0;
        const int adaptive = prm->confidence > 0. && prm->confidence < 1.;
        for (int it = 0; it < niters; it++)
            for (int k = 0; k < 3; k++) {
                const int good = counts[3 * it + k];
                const int floor_ = maxgood > 6 ? maxgood : 6;
                if (good > floor_) {
                    best = 3 * it + k; maxgood = good;
                    if (adaptive) niters = ransac_update_iters(prm->confidence, (double)(M - good) / M, 7, niters);
                }
            }
        if (!inlier_counts) free(counts);
    } else { /* runLMeDS */
        double conf = prm->confidence;
        if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
        int niters = (int)lrint(log(1 - conf) / log(1 - pow(1 - 0.45, 7.0)));
        niters = niters > 3 ? niters : 3;
        niters = niters < H ? niters : H;
        double min_median = DBL_MAX;
        float* err = (float*)malloc(sizeof(float) * (size_t)M);
        for (int it = 0; it < niters; it++)
            for (int k = 0; k < nm[it]; k++) {
                for (int i = 0; i < M; i++) {   /* (a NaN error -- a degenerate matrix -- sorts as +inf: its sign differs between machines) */
                    const float e = (float)orc_fmat_error(models + 27 * it + 9 * k, x1 + 2 * i, x2 + 2 * i, 1);
                    err[i] = e == e ? e : INFINITY;
                }
                qsort(err, (size_t)M, sizeof(float), cmp_float);
                const double median = M % 2 != 0 ? (double)err[M / 2] : (double)(float)(err[M / 2 - 1] + err[M / 2]) * 0.5;
                if (median < min_median) { min_median = median; best = 3 * it + k; }
            }
        free(err);
        if (best >= 0) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (M - 7)) * sqrt(min_median);
            sigma = sigma > 0.001 ? sigma : 0.001;
            thr = sigma * sigma;
        }
    }
    int rc = 0, n = 0;
    if (best >= 0) memcpy(Fb, models + 27 * (best / 3) + 9 * (best % 3), sizeof(Fb));
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)M);
    for (int i = 0; i < M; i++) {
        const int in = best >= 0 && orc_fmat_error(Fb, x1 + 2 * i, x2 + 2 * i, 1) <= thr;
        status[i] = (uint8_t)in;
        if (in) sel[n++] = i;
    }
    if (best >= 0 && n < 7) { best = -1; for (int i = 0; i < M; i++) status[i] = 0; n = 0; } /* runLMeDS: result = count >= modelPoints */
    if (best < 0) { memset(F, 0, sizeof(double) * 9); rc = -6; }
    else if (prm->refit == 2 && n >= 8) rc = orc_fmat_8point_normalized(x1, x2, sel, n, F); /* the second call, CV_FM_8POINT (:585) */
    else memcpy(F, Fb, sizeof(Fb));
    free(sel); free(models); free(nm);
    if (best_model) *best_model = best;
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
