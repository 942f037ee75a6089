/*
 * pnp.c -- oracle (test infrastructure, see oracle.h): PnP RANSAC.
 *
 * Follows pnpPoseEstimation, reference src/CameraPoseEstimator.cpp:460-474:
 *   solvePnPRansac(mapPoints, imagePoints, curK, distCoeff, rvec, tvec); Rodrigues(rvec, R);
 * OpenCV 2.4.13's solvePnPRansac is not vendored, not available here and not even
 * run-to-run deterministic (parallel_for_ over hypotheses with a mutex-shared RNG), so
 * only its CONTRACT can be restated -- PARITY UNPINNED:
 *   inputs   object points, image points, K, distortion (k1,k2,p1,p2,k3[,k4,k5,k6])
 *   scoring  project every point with the full distortion model (cvProjectPoints2) and
 *            count ||projected - observed|| < reprojectionError
 *   outputs  rvec, tvec of the pose refitted on the best inlier set
 * with a deterministic core chosen by this build: the sample table is an input (H x 4);
 * hypothesis h is the P3P pose of sample points 0..2 (bearings from iteratively undistorted
 * normalised image points), disambiguated by the reprojection error of sample point 3; the
 * winner (max count, lowest index, optional OpenCV-style adaptive bound) is refined by
 * Gauss-Newton on its inliers.
 *
 * The P3P solver works on the three distance constraints
 *   l_i^2 + l_j^2 - 2 (f_i.f_j) l_i l_j = |X_i - X_j|^2
 * by picking a degenerate member of the conic pencil they span (a real root of a cubic,
 * found by monotone Newton iteration), splitting it into its two planes with a 3x3 Jacobi
 * eigen-decomposition and intersecting each plane with another member of the pencil
 * (a quadratic) -- only + - * / sqrt, so the GPU reproduces it operation for operation.
 */
#include "oracle.h"
#include <float.h>
/* hot loops with explicit fma(): build a hardware-FMA clone and a portable one, picked at load time */
#if defined(__x86_64__) && defined(__GNUC__)
#define ORC_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_FMA_CLONES
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_pnp_default_params(orc_pnp_params* p)
{
    p->reproj_thr = 8.0; p->confidence = 0.0; p->refine_iters = 10; p->min_inliers = 4; p->stop_inliers = 0;
}

/* iterative inverse of the distortion model (cvUndistortPoints' fixed-point iteration, 10 rounds) */
void orc_undistort_normalize(const double* x, int n, const double K[9], const double k[8], double* xn)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double x0 = (x[2 * i] - cx) / fx, y0 = (x[2 * i + 1] - cy) / fy;
        double xx = x0, yy = y0;
        for (int it = 0; it < 10; it++) {
            double r2 = xx * xx + yy * yy;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double dX = 2 * k[2] * xx * yy + k[3] * (r2 + 2 * xx * xx);
            double dY = k[2] * (r2 + 2 * yy * yy) + 2 * k[3] * xx * yy;
            xx = (x0 - dX) * icdist;
            yy = (y0 - dY) * icdist;
        }
        xn[2 * i] = xx; xn[2 * i + 1] = yy;
    }
}

/* cvProjectPoints2 for one point.  The three rows of R X + t and the final pixel mapping are evaluated as explicit
 * fused multiply-add chains (one of the evaluations a contracting compiler produces for OpenCV's expressions; exact
 * IEEE fma, so CPU and GPU agree bit for bit); everything else is plain source-order arithmetic. */
ORC_FMA_CLONES
void orc_project(const double Rt[12], const double K[9], const double k[8], const double X[3], double uv[2])
{
    double x = fma(Rt[0], X[0], fma(Rt[1], X[1], fma(Rt[2], X[2], Rt[3])));
    double y = fma(Rt[4], X[0], fma(Rt[5], X[1], fma(Rt[6], X[2], Rt[7])));
    double z = fma(Rt[8], X[0], fma(Rt[9], X[1], fma(Rt[10], X[2], Rt[11])));
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
    double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
    uv[0] = fma(xd, K[0], K[2]);
    uv[1] = fma(yd, K[4], K[5]);
}

void orc_rodrigues_to_R(const double r[3], double R[9])
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; memcpy(R, I, sizeof(I)); return; }
    double c = cos(th), s = sin(th), c1 = 1 - c, it = 1 / th;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

void orc_R_to_rodrigues(const double R[9], double r[3])
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double th = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        double nn = sqrt(rx * rx + ry * ry + rz * rz);
        th /= nn;
        r[0] = rx * th; r[1] = ry * th; r[2] = rz * th;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= th;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* ---- P3P ---------------------------------------------------------------------------- */

/* a real root of x^3 + a x^2 + b x + c by Newton iteration started beyond the extreme root on
 * the side of the inflection point where a root is guaranteed (monotone convergence) */
static double cubic_root(double a, double b, double c)
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

static void adj3s(const double A[9], double J[9])
{
    J[0] = A[4] * A[8] - A[5] * A[7]; J[1] = A[2] * A[7] - A[1] * A[8]; J[2] = A[1] * A[5] - A[2] * A[4];
    J[3] = A[5] * A[6] - A[3] * A[8]; J[4] = A[0] * A[8] - A[2] * A[6]; J[5] = A[2] * A[3] - A[0] * A[5];
    J[6] = A[3] * A[7] - A[4] * A[6]; J[7] = A[1] * A[6] - A[0] * A[7]; J[8] = A[0] * A[4] - A[1] * A[3];
}

static double quad_form(const double D[9], const double p[3], const double q[3])
{
    double Dq0 = D[0] * q[0] + D[1] * q[1] + D[2] * q[2];
    double Dq1 = D[3] * q[0] + D[4] * q[1] + D[5] * q[2];
    double Dq2 = D[6] * q[0] + D[7] * q[1] + D[8] * q[2];
    return p[0] * Dq0 + p[1] * Dq1 + p[2] * Dq2;
}

/* pair index: (0,1)->0, (0,2)->1, (1,2)->2 */
static int pair_of(int i, int j) { return i + j - 1; }

int orc_p3p(const double f[9], const double X[9], double Rout[4][9], double tout[4][3])
{
    const double* f0 = f; const double* f1 = f + 3; const double* f2 = f + 6;
    double b[3], a[3];
    b[0] = f0[0] * f1[0] + f0[1] * f1[1] + f0[2] * f1[2];
    b[1] = f0[0] * f2[0] + f0[1] * f2[1] + f0[2] * f2[2];
    b[2] = f1[0] * f2[0] + f1[1] * f2[1] + f1[2] * f2[2];
    double d01[3], d02[3], d12[3];
    for (int k = 0; k < 3; k++) { d01[k] = X[k] - X[3 + k]; d02[k] = X[k] - X[6 + k]; d12[k] = X[3 + k] - X[6 + k]; }
    a[0] = d01[0] * d01[0] + d01[1] * d01[1] + d01[2] * d01[2];
    a[1] = d02[0] * d02[0] + d02[1] * d02[1] + d02[2] * d02[2];
    a[2] = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    if (!(a[0] > 0 && a[1] > 0 && a[2] > 0)) return 0;
    /* D1 = M01*a12 - M12*a01, D2 = M02*a12 - M12*a02 (both Lambda^T D Lambda = 0) */
    double D1[9] = { a[2], -a[2] * b[0], 0, -a[2] * b[0], a[2] - a[0], a[0] * b[2], 0, a[0] * b[2], -a[0] };
    double D2[9] = { a[2], 0, -a[2] * b[1], 0, -a[1], a[1] * b[2], -a[2] * b[1], a[1] * b[2], a[2] - a[1] };
    double J1[9], J2[9];
    adj3s(D1, J1); adj3s(D2, J2);
    double c0 = orc_det3(D1), c3 = orc_det3(D2), c1 = 0, c2 = 0;
    for (int k = 0; k < 9; k++) { c1 += J1[k] * D2[k]; c2 += D1[k] * J2[k]; }
    double cs = fabs(c0) + fabs(c1) + fabs(c2) + fabs(c3);
    if (!(fabs(c3) > 1e-12 * cs)) return 0;
    double ca = c2 / c3, cb = c1 / c3, cc = c0 / c3;
    double roots[3];
    int nroots = 1;
    roots[0] = cubic_root(ca, cb, cc);
    { /* deflate: x^2 + (ca + r) x + (cb + (ca + r) r) */
        double r = roots[0], p = ca + r, q = cb + p * r;
        double disc = p * p - 4.0 * q;
        if (disc >= 0.0) {
            double sq = sqrt(disc);
            double h = -0.5 * (p + (p >= 0.0 ? sq : -sq));
            roots[1] = h;
            roots[2] = h != 0.0 ? q / h : 0.0;
            nroots = 3;
        }
    }
    int nsol = 0;
    for (int ri = 0; ri < nroots && nsol == 0; ri++) {
        double g = roots[ri];
        double D0[9], e[3], E[9];
        for (int k = 0; k < 9; k++) D0[k] = D1[k] + g * D2[k];
        orc_jacobi_eig(D0, 3, e, E);
        /* a real plane pair needs eigenvalues (-, ~0, +): the middle one is the null direction */
        if (!(e[0] < 0.0 && e[2] > 0.0) || !(fabs(e[1]) <= fabs(e[0]) && fabs(e[1]) <= fabs(e[2]))) continue;
        double s = sqrt(-e[0] / e[2]);
        const double* D = fabs(g) <= 1.0 ? D2 : D1; /* another member of the pencil */
        for (int sign = 0; sign < 2; sign++) {
            double w[3];
            for (int k = 0; k < 3; k++) w[k] = E[k * 3 + 2] + (sign ? -s : s) * E[k * 3 + 0];
            int kk = 0;
            if (fabs(w[1]) > fabs(w[kk])) kk = 1;
            if (fabs(w[2]) > fabs(w[kk])) kk = 2;
            if (w[kk] == 0.0) continue;
            int ii = kk == 0 ? 1 : 0, jj = kk == 2 ? 1 : 2;
            double al = -w[ii] / w[kk], be = -w[jj] / w[kk]; /* l_k = al l_i + be l_j */
            double p[3] = { 0, 0, 0 }, q[3] = { 0, 0, 0 };
            p[ii] = 1.0; p[kk] = al; q[jj] = 1.0; q[kk] = be;
            double A = quad_form(D, p, p), B = 2.0 * quad_form(D, p, q), C = quad_form(D, q, q);
            double taus[2];
            int nt = 0;
            if (A == 0.0) { if (B != 0.0) taus[nt++] = -C / B; }
            else {
                double disc = B * B - 4.0 * A * C;
                if (disc >= 0.0) {
                    double sq = sqrt(disc);
                    double h = -0.5 * (B + (B >= 0.0 ? sq : -sq));
                    taus[nt++] = h / A;
                    if (h != 0.0) taus[nt++] = C / h;
                }
            }
            for (int ti = 0; ti < nt && nsol < 4; ti++) {
                double tau = taus[ti];
                if (!(tau > 0.0)) continue;
                double bij = b[pair_of(ii, jj)], aij = a[pair_of(ii, jj)];
                double den = tau * tau + 1.0 - 2.0 * bij * tau;
                if (!(den > 0.0)) continue;
                double L[3];
                L[jj] = sqrt(aij / den);
                L[ii] = tau * L[jj];
                L[kk] = al * L[ii] + be * L[jj];
                if (!(L[kk] > 0.0)) continue;
                /* Gauss-Newton polish on the three distance constraints (5 fixed rounds) */
                for (int it = 0; it < 5; it++) {
                    double r0 = L[0] * L[0] + L[1] * L[1] - 2.0 * b[0] * L[0] * L[1] - a[0];
                    double r1 = L[0] * L[0] + L[2] * L[2] - 2.0 * b[1] * L[0] * L[2] - a[1];
                    double r2 = L[1] * L[1] + L[2] * L[2] - 2.0 * b[2] * L[1] * L[2] - a[2];
                    double Jm[9] = { 2.0 * L[0] - 2.0 * b[0] * L[1], 2.0 * L[1] - 2.0 * b[0] * L[0], 0.0,
                                     2.0 * L[0] - 2.0 * b[1] * L[2], 0.0, 2.0 * L[2] - 2.0 * b[1] * L[0],
                                     0.0, 2.0 * L[1] - 2.0 * b[2] * L[2], 2.0 * L[2] - 2.0 * b[2] * L[1] };
                    double Ja[9];
                    adj3s(Jm, Ja);
                    double det = orc_det3(Jm);
                    if (det == 0.0) break;
                    double id = 1.0 / det;
                    L[0] -= (Ja[0] * r0 + Ja[1] * r1 + Ja[2] * r2) * id;
                    L[1] -= (Ja[3] * r0 + Ja[4] * r1 + Ja[5] * r2) * id;
                    L[2] -= (Ja[6] * r0 + Ja[7] * r1 + Ja[8] * r2) * id;
                }
                if (!(L[0] > 0.0 && L[1] > 0.0 && L[2] > 0.0)) continue;
                /* pose: R maps the world triad onto the camera triad, t = Y0 - R X0 */
                double Y[9];
                for (int k = 0; k < 3; k++) { Y[k] = L[0] * f0[k]; Y[3 + k] = L[1] * f1[k]; Y[6 + k] = L[2] * f2[k]; }
                double y1[3], y2[3], y3[3], x3[3];
                for (int k = 0; k < 3; k++) { y1[k] = Y[k] - Y[3 + k]; y2[k] = Y[k] - Y[6 + k]; }
                y3[0] = y1[1] * y2[2] - y1[2] * y2[1]; y3[1] = y1[2] * y2[0] - y1[0] * y2[2]; y3[2] = y1[0] * y2[1] - y1[1] * y2[0];
                x3[0] = d01[1] * d02[2] - d01[2] * d02[1]; x3[1] = d01[2] * d02[0] - d01[0] * d02[2]; x3[2] = d01[0] * d02[1] - d01[1] * d02[0];
                double Xm[9] = { d01[0], d02[0], x3[0], d01[1], d02[1], x3[1], d01[2], d02[2], x3[2] };
                double Ym[9] = { y1[0], y2[0], y3[0], y1[1], y2[1], y3[1], y1[2], y2[2], y3[2] };
                double Xa[9];
                adj3s(Xm, Xa);
                double dx = orc_det3(Xm);
                if (dx == 0.0) continue;
                double idx_ = 1.0 / dx;
                double* R = Rout[nsol];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        R[i * 3 + j] = (Ym[i * 3] * Xa[j] + Ym[i * 3 + 1] * Xa[3 + j] + Ym[i * 3 + 2] * Xa[6 + j]) * idx_;
                for (int i = 0; i < 3; i++)
                    tout[nsol][i] = Y[i] - (R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2]);
                nsol++;
            }
        }
    }
    return nsol;
}

/* hypothesis: P3P on sample points 0..2, disambiguated by the squared reprojection error of point 3 */
int orc_pnp_hypothesis(const double* X, const double* xn, const double* x, const double K[9], const double dist[8],
                       const int32_t s[4], double Rt[12])
{
    double f[9], Xs[9];
    for (int k = 0; k < 3; k++) {
        double bx = xn[2 * s[k]], by = xn[2 * s[k] + 1];
        double inv = 1.0 / sqrt(bx * bx + by * by + 1.0);
        f[3 * k] = bx * inv; f[3 * k + 1] = by * inv; f[3 * k + 2] = inv;
        Xs[3 * k] = X[3 * s[k]]; Xs[3 * k + 1] = X[3 * s[k] + 1]; Xs[3 * k + 2] = X[3 * s[k] + 2];
    }
    double R[4][9], t[4][3];
    int n = orc_p3p(f, Xs, R, t);
    int best = -1;
    double beste = 0;
    for (int c = 0; c < n; c++) {
        double cand[12], uv[2];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) cand[i * 4 + j] = R[c][i * 3 + j]; cand[i * 4 + 3] = t[c][i]; }
        orc_project(cand, K, dist, X + 3 * s[3], uv);
        double dx = uv[0] - x[2 * s[3]], dy = uv[1] - x[2 * s[3] + 1];
        double e = fma(dx, dx, dy * dy);
        if (!(e == e)) continue; /* NaN */
        if (best < 0 || e < beste) { best = c; beste = e; memcpy(Rt, cand, sizeof(cand)); }
    }
    return best >= 0;
}

ORC_FMA_CLONES
static int pnp_count(const double* X, const double* x, int P, const double K[9], const double dist[8],
                     const double Rt[12], double thr, uint8_t* mask)
{
    int c = 0;
    for (int i = 0; i < P; i++) {
        double uv[2];
        orc_project(Rt, K, dist, X + 3 * i, uv);
        double dx = x[2 * i] - uv[0], dy = x[2 * i + 1] - uv[1];
        int in = sqrt(fma(dx, dx, dy * dy)) < thr; /* norm(points[i] - projected[i]) < reprojectionError */
        if (mask) mask[i] = (uint8_t)in;
        c += in;
    }
    return c;
}

/* Gauss-Newton on the reprojection error of the masked points; pose update T <- exp(delta) T with
 * delta = (omega, upsilon); Jacobians include the distortion model */
int orc_pnp_refine(const double* X, const double* x, int P, const uint8_t* mask, const double K[9], const double k[8],
                   int iters, double Rt[12])
{
    for (int it = 0; it < iters; it++) {
        double H[36], g[6];
        memset(H, 0, sizeof(H)); memset(g, 0, sizeof(g));
        int used = 0;
        for (int i = 0; i < P; i++) {
            if (mask && !mask[i]) continue;
            const double* Xi = X + 3 * i;
            double px = Rt[0] * Xi[0] + Rt[1] * Xi[1] + Rt[2] * Xi[2] + Rt[3];
            double py = Rt[4] * Xi[0] + Rt[5] * Xi[1] + Rt[6] * Xi[2] + Rt[7];
            double pz = Rt[8] * Xi[0] + Rt[9] * Xi[1] + Rt[10] * Xi[2] + Rt[11];
            if (pz == 0.0) continue;
            double iz = 1.0 / pz, xx = px * iz, yy = py * iz;
            double r2 = xx * xx + yy * yy, r4 = r2 * r2, r6 = r4 * r2;
            double cd = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
            double icd2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
            double xd = xx * cd * icd2 + k[2] * 2 * xx * yy + k[3] * (r2 + 2 * xx * xx);
            double yd = yy * cd * icd2 + k[2] * (r2 + 2 * yy * yy) + k[3] * 2 * xx * yy;
            double ru = x[2 * i] - (xd * K[0] + K[2]), rv = x[2 * i + 1] - (yd * K[4] + K[5]);
            /* d(xd,yd)/d(xx,yy) */
            double dcd = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4;          /* d cd / d r2 */
            double dicd = -(k[5] + 2 * k[6] * r2 + 3 * k[7] * r4) * icd2 * icd2; /* d icd2 / d r2 */
            double rad = cd * icd2, drad = dcd * icd2 + cd * dicd;       /* radial factor and d/d r2 */
            double dxx = rad + xx * drad * 2 * xx + 2 * k[2] * yy + k[3] * (2 * xx + 4 * xx);
            double dxy = xx * drad * 2 * yy + 2 * k[2] * xx + k[3] * 2 * yy;
            double dyx = yy * drad * 2 * xx + k[2] * 2 * xx + 2 * k[3] * yy;
            double dyy = rad + yy * drad * 2 * yy + k[2] * (2 * yy + 4 * yy) + 2 * k[3] * xx;
            /* d(xx,yy)/dPc */
            double a00 = iz, a02 = -xx * iz, a11 = iz, a12 = -yy * iz;
            /* d(u,v)/dPc (2x3) */
            double m[6];
            m[0] = K[0] * (dxx * a00); m[1] = K[0] * (dxy * a11); m[2] = K[0] * (dxx * a02 + dxy * a12);
            m[3] = K[4] * (dyx * a00); m[4] = K[4] * (dyy * a11); m[5] = K[4] * (dyx * a02 + dyy * a12);
            /* dPc/d(omega) = -[Pc]x, dPc/d(upsilon) = I */
            double Ju[6], Jv[6];
            Ju[0] = m[1] * (-pz) + m[2] * (py);  Ju[1] = m[0] * (pz) + m[2] * (-px); Ju[2] = m[0] * (-py) + m[1] * (px);
            Ju[3] = m[0]; Ju[4] = m[1]; Ju[5] = m[2];
            Jv[0] = m[4] * (-pz) + m[5] * (py);  Jv[1] = m[3] * (pz) + m[5] * (-px); Jv[2] = m[3] * (-py) + m[4] * (px);
            Jv[3] = m[3]; Jv[4] = m[4]; Jv[5] = m[5];
            for (int a = 0; a < 6; a++) {
                g[a] += Ju[a] * ru + Jv[a] * rv;
                for (int c = a; c < 6; c++) H[a * 6 + c] += Ju[a] * Ju[c] + Jv[a] * Jv[c];
            }
            used++;
        }
        if (used < 3) return -6;
        for (int a = 0; a < 6; a++) for (int c = 0; c < a; c++) H[a * 6 + c] = H[c * 6 + a];
        /* Cholesky solve H d = g */
        double Lc[36];
        memset(Lc, 0, sizeof(Lc));
        for (int i = 0; i < 6; i++)
            for (int j = 0; j <= i; j++) {
                double sum = H[i * 6 + j];
                for (int q = 0; q < j; q++) sum -= Lc[i * 6 + q] * Lc[j * 6 + q];
                if (i == j) { if (!(sum > 0.0)) return -6; Lc[i * 6 + i] = sqrt(sum); }
                else Lc[i * 6 + j] = sum / Lc[j * 6 + j];
            }
        double yv[6], d[6];
        for (int i = 0; i < 6; i++) { double sum = g[i]; for (int q = 0; q < i; q++) sum -= Lc[i * 6 + q] * yv[q]; yv[i] = sum / Lc[i * 6 + i]; }
        for (int i = 5; i >= 0; i--) { double sum = yv[i]; for (int q = i + 1; q < 6; q++) sum -= Lc[q * 6 + i] * d[q]; d[i] = sum / Lc[i * 6 + i]; }
        /* T <- exp(d) T: R' = dR R, t' = dR t + V upsilon */
        double dR[9], V[9];
        double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], th = sqrt(th2);
        double A_, B_, C_;
        if (th < 1e-8) { A_ = 1.0; B_ = 0.5; C_ = 1.0 / 6.0; }
        else { A_ = sin(th) / th; B_ = (1 - cos(th)) / th2; C_ = (th - sin(th)) / (th2 * th); }
        double Om[9] = { 0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0 };
        double Om2[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double acc = 0; for (int q = 0; q < 3; q++) acc += Om[i * 3 + q] * Om[q * 3 + j]; Om2[i * 3 + j] = acc; }
        for (int i = 0; i < 9; i++) { double I = (i % 4 == 0) ? 1.0 : 0.0; dR[i] = I + A_ * Om[i] + B_ * Om2[i]; V[i] = I + B_ * Om[i] + C_ * Om2[i]; }
        double nRt[12];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 4; j++) nRt[i * 4 + j] = dR[i * 3] * Rt[j] + dR[i * 3 + 1] * Rt[4 + j] + dR[i * 3 + 2] * Rt[8 + j];
            nRt[i * 4 + 3] += V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5];
        }
        memcpy(Rt, nRt, sizeof(nRt));
        /* converged: the step is below double resolution of the pose (quadratic convergence: usually 3-4 rounds) */
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5] < 1e-20) break;
    }
    return 0;
}

int orc_pnp_ransac(const double* X, const double* x, int P, const double K[9], const double dist[8],
                   const int32_t* samples, int H, const orc_pnp_params* prm, double rvec[3], double tvec[3],
                   double Rt[12], uint8_t* inliers, int32_t* inlier_counts, int* best_h, int* n_inliers)
{
    if (P < 4 || H < 1) return -4;
    double* xn = (double*)malloc(sizeof(double) * 2 * (size_t)P);
    orc_undistort_normalize(x, P, K, dist, xn);
    int32_t* counts = inlier_counts ? inlier_counts : (int32_t*)malloc(sizeof(int32_t) * (size_t)H);
    for (int h = 0; h < H; h++) {
        double hyp[12];
        counts[h] = orc_pnp_hypothesis(X, xn, x, K, dist, samples + 4 * h, hyp) ? pnp_count(X, x, P, K, dist, hyp, prm->reproj_thr, NULL) : 0;
    }
    int bc = 0;
    int best = orc_ransac_select_stop(counts, H, P, 4, prm->confidence, prm->stop_inliers, &bc);
    if (!inlier_counts) free(counts);
    int rc = 0;
    if (best < 0 || bc < prm->min_inliers) rc = -6;
    else {
        orc_pnp_hypothesis(X, xn, x, K, dist, samples + 4 * best, Rt);
        int n = pnp_count(X, x, P, K, dist, Rt, prm->reproj_thr, inliers);
        if (prm->refine_iters > 0) orc_pnp_refine(X, x, P, inliers, K, dist, prm->refine_iters, Rt);
        double R[9] = { Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10] };
        if (rvec) orc_R_to_rodrigues(R, rvec);
        if (tvec) { tvec[0] = Rt[3]; tvec[1] = Rt[7]; tvec[2] = Rt[11]; }
        if (best_h) *best_h = best;
        if (n_inliers) *n_inliers = n;
    }
    free(xn);
    return rc;
}
