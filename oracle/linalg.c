/*
 * linalg.c -- oracle (test infrastructure, see oracle.h): fixed-size dense linear algebra.
 *
 * Stands in for the cv::SVD calls the reference makes through src/CommonMath.cpp:8-22
 * (TakeSVD, solveHLS = last row of Vt) and cv::determinant (src/CameraPoseEstimator.cpp:328).
 * OpenCV and Eigen are absent here, so everything is built on one primitive: the cyclic
 * Jacobi eigen-decomposition of a small symmetric matrix, which uses only + - * / sqrt and
 * therefore evaluates identically on any IEEE-754 machine (the GPU path repeats the same
 * operation sequence).  Every consumer in the reference is invariant to the sign of a
 * singular vector (dehomogenize src/CommonMath.cpp:24-27; F up to scale; E sign handled at
 * src/CameraPoseEstimator.cpp:328-331), so any SVD with descending singular values is a
 * faithful stand-in.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define JMAX 12

/* eigenvalues ascending in w, eigenvectors in the columns of V (row-major n x n) */
void orc_jacobi_eig(const double* Ain, int n, double* w, double* V)
{
    double a[JMAX][JMAX], v[JMAX][JMAX], d[JMAX], bb[JMAX], z[JMAX];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { a[i][j] = Ain[i * n + j]; v[i][j] = (i == j) ? 1.0 : 0.0; }
        d[i] = bb[i] = a[i][i];
        z[i] = 0.0;
    }
    for (int sweep = 1; sweep <= 60; sweep++) {
        double sm = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) sm += fabs(a[p][q]);
        if (sm == 0.0) break;
        double tresh = sweep < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double g = 100.0 * fabs(a[p][q]);
                if (sweep > 4 && fabs(d[p]) + g == fabs(d[p]) && fabs(d[q]) + g == fabs(d[q])) {
                    a[p][q] = 0.0;
                } else if (fabs(a[p][q]) > tresh) {
                    double h = d[q] - d[p], t;
                    if (fabs(h) + g == fabs(h)) t = a[p][q] / h;
                    else {
                        double theta = 0.5 * h / a[p][q];
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
                    h = t * a[p][q];
                    z[p] -= h; z[q] += h; d[p] -= h; d[q] += h;
                    a[p][q] = 0.0;
                    for (int j = 0; j < p; j++) { double x = a[j][p], y = a[j][q]; a[j][p] = x - s * (y + x * tau); a[j][q] = y + s * (x - y * tau); }
                    for (int j = p + 1; j < q; j++) { double x = a[p][j], y = a[j][q]; a[p][j] = x - s * (y + x * tau); a[j][q] = y + s * (x - y * tau); }
                    for (int j = q + 1; j < n; j++) { double x = a[p][j], y = a[q][j]; a[p][j] = x - s * (y + x * tau); a[q][j] = y + s * (x - y * tau); }
                    for (int j = 0; j < n; j++) { double x = v[j][p], y = v[j][q]; v[j][p] = x - s * (y + x * tau); v[j][q] = y + s * (x - y * tau); }
                }
            }
        for (int i = 0; i < n; i++) { bb[i] += z[i]; d[i] = bb[i]; z[i] = 0.0; }
    }
    /* ascending selection sort (stable for equal values by original index) */
    int order[JMAX];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n - 1; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (d[order[j]] < d[order[k]]) k = j;
        int t = order[k];
        for (int j = k; j > i; j--) order[j] = order[j - 1];
        order[i] = t;
    }
    for (int i = 0; i < n; i++) {
        w[i] = d[order[i]];
        for (int j = 0; j < n; j++) V[j * n + i] = v[j][order[i]];
    }
}

double orc_det3(const double A[9])
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* A = U diag(s) Vt with s descending; built from eig(A^T A).  Columns of U for singular
 * values that are numerically zero are completed to a right-handed orthonormal basis. */
void orc_svd3(const double A[9], double U[9], double s[3], double Vt[9])
{
    double AtA[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = acc;
        }
    orc_jacobi_eig(AtA, 3, w, V);
    double u[3][3]; /* u[c] = column c */
    double vcol[3][3];
    for (int c = 0; c < 3; c++) {
        int src = 2 - c; /* descending */
        double lam = w[src] > 0 ? w[src] : 0;
        s[c] = sqrt(lam);
        for (int r = 0; r < 3; r++) vcol[c][r] = V[r * 3 + src];
    }
    double smax = s[0] > 0 ? s[0] : 1.0;
    int have[3] = { 0, 0, 0 };
    for (int c = 0; c < 3; c++) {
        /* eigenvalues of A^T A carry an absolute error of about eps*|A|^2, so singular values below
         * sqrt(eps)*smax are noise: treat them as zero and complete U by orthogonality */
        if (s[c] > 1e-6 * smax) {
            for (int r = 0; r < 3; r++)
                u[c][r] = (A[r * 3 + 0] * vcol[c][0] + A[r * 3 + 1] * vcol[c][1] + A[r * 3 + 2] * vcol[c][2]) / s[c];
            have[c] = 1;
        }
    }
    if (have[0] && have[1] && !have[2]) {
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    } else if (!(have[0] && have[1] && have[2])) {
        /* rank <= 1: complete with any orthonormal basis */
        if (!have[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
        double ax = fabs(u[0][0]), ay = fabs(u[0][1]), az = fabs(u[0][2]);
        double e[3] = { 0, 0, 0 };
        if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
        double dp = e[0] * u[0][0] + e[1] * u[0][1] + e[2] * u[0][2], nn = 0;
        for (int r = 0; r < 3; r++) { u[1][r] = e[r] - dp * u[0][r]; nn += u[1][r] * u[1][r]; }
        nn = sqrt(nn);
        for (int r = 0; r < 3; r++) u[1][r] /= nn;
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    }
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) { U[r * 3 + c] = u[c][r]; Vt[c * 3 + r] = vcol[c][r]; }
}

/* Null vector of an 8 x 9 matrix (the minimal eight-point system) by Gauss-Jordan elimination
 * with complete pivoting: the stand-in for "last column of V of SVD(A)" in fundamentalMatrix
 * (reference src/CameraPoseEstimator.cpp:690-705) when A has exactly 8 rows.  The pivot is the
 * first largest |a| in row-major scan order; eliminated entries are set to exactly 0; the one
 * column never chosen as a pivot is the free variable (= 1); the result has unit norm.
 * A (row-major, 72 doubles) is destroyed.  Returns 0 if a pivot is exactly zero (rank < 8). */
int orc_null_8x9(double* A, double f[9])
{
    int used_r[8] = { 0 }, used_c[9] = { 0 }, prow[8], pcol[8];
    for (int k = 0; k < 8; k++) {
        double best = -1.0;
        int br = 0, bc = 0;
        for (int r = 0; r < 8; r++) {
            if (used_r[r]) continue;
            for (int c = 0; c < 9; c++) {
                if (used_c[c]) continue;
                double v = fabs(A[r * 9 + c]);
                if (v > best) { best = v; br = r; bc = c; }
            }
        }
        used_r[br] = 1; used_c[bc] = 1; prow[k] = br; pcol[k] = bc;
        double pv = A[br * 9 + bc];
        if (pv == 0.0) { for (int c = 0; c < 9; c++) f[c] = 0.0; return 0; }
        for (int r = 0; r < 8; r++) {
            if (r == br) continue;
            double m = A[r * 9 + bc] / pv;
            for (int c = 0; c < 9; c++) A[r * 9 + c] = (c == bc) ? 0.0 : A[r * 9 + c] - m * A[br * 9 + c];
        }
    }
    int cf = 0;
    for (int c = 0; c < 9; c++) if (!used_c[c]) cf = c;
    f[cf] = 1.0;
    for (int k = 0; k < 8; k++) f[pcol[k]] = -A[prow[k] * 9 + cf] / A[prow[k] * 9 + pcol[k]];
    double nn = 0.0;
    for (int c = 0; c < 9; c++) nn += f[c] * f[c];
    nn = sqrt(nn);
    for (int c = 0; c < 9; c++) f[c] = f[c] / nn;
    return 1;
}
