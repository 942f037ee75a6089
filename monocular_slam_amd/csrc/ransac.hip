// ransac.hip -- K8: eight-point fundamental-matrix RANSAC, PnP (P3P) RANSAC, two-view
// triangulation (gfx950).  All f64, plain IEEE evaluation (built with -ffp-contract=off) so
// that every inlier decision equals the CPU restatement's.
//
// Stands in for computeFundamentalMatrix / computeFundamentalMatrix2 (reference
// src/CameraPoseEstimator.cpp:545-586, :596-786), solvePnPRansac + Rodrigues (:472-473) and
// TriangulateMultiplePointsFromTwoView (:136-152).
//
// Structure (per batch of independent problems):
//   *_prepare_k   per-point preprocessing (scaling through float / iterative undistortion)
//   *_solve_k     ONE HYPOTHESIS PER LANE: minimal solver (8x9 Gauss-Jordan null vector +
//                 rank-2, or P3P + 4th-point disambiguation) -> model table in HBM
//   *_score_k     ONE HYPOTHESIS PER WAVEFRONT-ITERATION: a workgroup owns HT hypotheses
//                 (models staged in LDS), its lanes stride over the points (each point is
//                 loaded once and reused HT times from registers), inlier votes are taken
//                 with __ballot + popcount and summed per hypothesis in LDS
//   select_k      argmax (max count, lowest index) with OpenCV's adaptive iteration bound
//                 emulated over the ordered counts
//   *_finalize_k  one workgroup per problem: inlier mask of the winner, refit (block
//                 reductions, fixed tree order => run-to-run deterministic), outputs
// No MFMA: the work is small fixed-size solves and a scored reduction, not a contraction.
#include "common.h"
#include <cfloat>
#include <cmath>

namespace {

constexpr int SCORE_THREADS = 256;

// ------------------------------------------------------------------ small linear algebra
__device__ __forceinline__ double det3(const double* A)
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

__device__ __forceinline__ void adj3s(const double* A, double* J)
{
    J[0] = A[4] * A[8] - A[5] * A[7]; J[1] = A[2] * A[7] - A[1] * A[8]; J[2] = A[1] * A[5] - A[2] * A[4];
    J[3] = A[5] * A[6] - A[3] * A[8]; J[4] = A[0] * A[8] - A[2] * A[6]; J[5] = A[2] * A[3] - A[0] * A[5];
    J[6] = A[3] * A[7] - A[4] * A[6]; J[7] = A[1] * A[6] - A[0] * A[7]; J[8] = A[0] * A[4] - A[1] * A[3];
}

// One Jacobi rotation on the pair (p,q) of a symmetric 3x3 held in scalars.  (x,y) are the two
// off-diagonal entries that couple p and q to the third index; vXp/vXq are eigenvector entries.
#define JAC3_ROT(apq, dp, dq, zp, zq, x, y, v0p, v0q, v1p, v1q, v2p, v2q)                               \
    {                                                                                                     \
        double g = 100.0 * fabs(apq);                                                                     \
        if (sweep > 4 && fabs(dp) + g == fabs(dp) && fabs(dq) + g == fabs(dq)) { apq = 0.0; }             \
        else if (fabs(apq) > tresh) {                                                                     \
            double h = dq - dp, t;                                                                        \
            if (fabs(h) + g == fabs(h)) t = apq / h;                                                      \
            else {                                                                                        \
                double theta = 0.5 * h / apq;                                                             \
                t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));                                      \
                if (theta < 0.0) t = -t;                                                                  \
            }                                                                                             \
            double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);                           \
            h = t * apq;                                                                                  \
            zp -= h; zq += h; dp -= h; dq += h; apq = 0.0;                                                \
            { double X_ = x, Y_ = y; x = X_ - s * (Y_ + X_ * tau); y = Y_ + s * (X_ - Y_ * tau); }        \
            { double X_ = v0p, Y_ = v0q; v0p = X_ - s * (Y_ + X_ * tau); v0q = Y_ + s * (X_ - Y_ * tau); } \
            { double X_ = v1p, Y_ = v1q; v1p = X_ - s * (Y_ + X_ * tau); v1q = Y_ + s * (X_ - Y_ * tau); } \
            { double X_ = v2p, Y_ = v2q; v2p = X_ - s * (Y_ + X_ * tau); v2q = Y_ + s * (X_ - Y_ * tau); } \
        }                                                                                                 \
    }

// cyclic Jacobi for symmetric 3x3 (same sweep/threshold rules as the generic n x n routine);
// eigenvalues ascending in w, eigenvectors in the columns of V (row-major)
__device__ void jacobi3(const double* A, double* w, double* V)
{
    double a01 = A[1], a02 = A[2], a12 = A[5];
    double d0 = A[0], d1 = A[4], d2 = A[8];
    double b0 = d0, b1 = d1, b2 = d2, z0 = 0.0, z1 = 0.0, z2 = 0.0;
    double v00 = 1.0, v01 = 0.0, v02 = 0.0, v10 = 0.0, v11 = 1.0, v12 = 0.0, v20 = 0.0, v21 = 0.0, v22 = 1.0;
    for (int sweep = 1; sweep <= 60; sweep++) {
        double sm = fabs(a01) + fabs(a02) + fabs(a12);
        if (sm == 0.0) break;
        double tresh = sweep < 4 ? 0.2 * sm / 9 : 0.0;
        JAC3_ROT(a01, d0, d1, z0, z1, a02, a12, v00, v01, v10, v11, v20, v21)
        JAC3_ROT(a02, d0, d2, z0, z2, a01, a12, v00, v02, v10, v12, v20, v22)
        JAC3_ROT(a12, d1, d2, z1, z2, a01, a02, v01, v02, v11, v12, v21, v22)
        b0 += z0; d0 = b0; z0 = 0.0;
        b1 += z1; d1 = b1; z1 = 0.0;
        b2 += z2; d2 = b2; z2 = 0.0;
    }
    // stable ascending order of (d0,d1,d2)
    int o0 = 0, o1 = 1, o2 = 2;
    double e0 = d0, e1 = d1, e2 = d2;
    // selection: smallest first (first index wins ties)
    if (e1 < e0) { double t = e0; e0 = e1; e1 = t; int ti = o0; o0 = o1; o1 = ti; }
    if (e2 < e0) { double t = e0; e0 = e2; e2 = e1; e1 = t; int ti = o0; o0 = o2; o2 = o1; o1 = ti; }
    else if (e2 < e1) { double t = e1; e1 = e2; e2 = t; int ti = o1; o1 = o2; o2 = ti; }
    w[0] = e0; w[1] = e1; w[2] = e2;
    const double vr[3][3] = { { v00, v01, v02 }, { v10, v11, v12 }, { v20, v21, v22 } };
#pragma unroll
    for (int r = 0; r < 3; r++) {
        V[r * 3 + 0] = o0 == 0 ? vr[r][0] : (o0 == 1 ? vr[r][1] : vr[r][2]);
        V[r * 3 + 1] = o1 == 0 ? vr[r][0] : (o1 == 1 ? vr[r][1] : vr[r][2]);
        V[r * 3 + 2] = o2 == 0 ? vr[r][0] : (o2 == 1 ? vr[r][1] : vr[r][2]);
    }
}

// generic cyclic Jacobi on an n x n symmetric matrix held in memory (LDS or scratch), n <= 9.
// Single-lane, used only by the once-per-problem refits.
__device__ void jacobi_mem(double* a, int n, double* d, double* v)
{
    double bb[9], z[9];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
        d[i] = bb[i] = a[i * n + i];
        z[i] = 0.0;
    }
    for (int sweep = 1; sweep <= 60; sweep++) {
        double sm = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) sm += fabs(a[p * n + q]);
        if (sm == 0.0) break;
        double tresh = sweep < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[p * n + q];
                double g = 100.0 * fabs(apq);
                if (sweep > 4 && fabs(d[p]) + g == fabs(d[p]) && fabs(d[q]) + g == fabs(d[q])) a[p * n + q] = 0.0;
                else if (fabs(apq) > tresh) {
                    double h = d[q] - d[p], t;
                    if (fabs(h) + g == fabs(h)) t = apq / h;
                    else {
                        double theta = 0.5 * h / apq;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
                    h = t * apq;
                    z[p] -= h; z[q] += h; d[p] -= h; d[q] += h;
                    a[p * n + q] = 0.0;
                    for (int j = 0; j < p; j++) { double x = a[j * n + p], y = a[j * n + q]; a[j * n + p] = x - s * (y + x * tau); a[j * n + q] = y + s * (x - y * tau); }
                    for (int j = p + 1; j < q; j++) { double x = a[p * n + j], y = a[j * n + q]; a[p * n + j] = x - s * (y + x * tau); a[j * n + q] = y + s * (x - y * tau); }
                    for (int j = q + 1; j < n; j++) { double x = a[p * n + j], y = a[q * n + j]; a[p * n + j] = x - s * (y + x * tau); a[q * n + j] = y + s * (x - y * tau); }
                    for (int j = 0; j < n; j++) { double x = v[j * n + p], y = v[j * n + q]; v[j * n + p] = x - s * (y + x * tau); v[j * n + q] = y + s * (x - y * tau); }
                }
            }
        for (int i = 0; i < n; i++) { bb[i] += z[i]; d[i] = bb[i]; z[i] = 0.0; }
    }
}

// The same cyclic Jacobi executed by a whole workgroup on matrices in LDS: every thread evaluates the (identical)
// rotation parameters, lanes 0..n-1 rotate the matrix entries and lanes n..2n-1 the eigenvector rows of one (p,q)
// step.  Element for element the arithmetic equals jacobi_mem / the oracle's routine.  All threads must call it.
__device__ void jacobi_block(double* a, int n, double* d, double* v, double* bb, double* z, int tid)
{
    for (int i = tid; i < n * n; i += blockDim.x) v[i] = (i / n == i % n) ? 1.0 : 0.0;
    if (tid < n) { d[tid] = bb[tid] = a[tid * n + tid]; z[tid] = 0.0; }
    __syncthreads();
    for (int sweep = 1; sweep <= 60; sweep++) {
        double sm = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) sm += fabs(a[p * n + q]);
        if (sm == 0.0) break; // uniform: every thread read the same values
        const double tresh = sweep < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = a[p * n + q], dp = d[p], dq = d[q];
                const double g = 100.0 * fabs(apq);
                const bool kill = sweep > 4 && fabs(dp) + g == fabs(dp) && fabs(dq) + g == fabs(dq);
                const bool rot = !kill && fabs(apq) > tresh;
                if (!kill && !rot) continue; // uniform
                double s_ = 0.0, tau = 0.0, h = 0.0;
                if (rot) {
                    double hh = dq - dp, t;
                    if (fabs(hh) + g == fabs(hh)) t = apq / hh;
                    else {
                        const double theta = 0.5 * hh / apq;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    s_ = t * c; tau = s_ / (1.0 + c); h = t * apq;
                }
                __syncthreads(); // everyone has read a[p][q], d[p], d[q]
                if (tid == 0) {
                    if (rot) { z[p] -= h; z[q] += h; d[p] -= h; d[q] += h; }
                    a[p * n + q] = 0.0;
                }
                if (rot && tid < n && tid != p && tid != q) {
                    const int j = tid;
                    double* px = j < p ? &a[j * n + p] : &a[p * n + j];
                    double* py = j < q ? &a[j * n + q] : &a[q * n + j];
                    const double x = *px, y = *py;
                    *px = x - s_ * (y + x * tau); *py = y + s_ * (x - y * tau);
                } else if (rot && tid >= n && tid < 2 * n) {
                    const int j = tid - n;
                    const double x = v[j * n + p], y = v[j * n + q];
                    v[j * n + p] = x - s_ * (y + x * tau); v[j * n + q] = y + s_ * (x - y * tau);
                }
                __syncthreads();
            }
        if (tid < n) { bb[tid] += z[tid]; d[tid] = bb[tid]; z[tid] = 0.0; }
        __syncthreads();
    }
}

// index of the smallest eigenvalue, first index on ties (== column 0 after the stable ascending sort)
__device__ __forceinline__ int argmin_first(const double* d, int n)
{
    int k = 0;
    for (int i = 1; i < n; i++) if (d[i] < d[k]) k = i;
    return k;
}

// F <- F - (F v3) v3^T with v3 the right singular vector of the smallest singular value
__device__ void rank2(double* F)
{
    double FtF[9], w[3], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) acc += F[k * 3 + i] * F[k * 3 + j];
            FtF[i * 3 + j] = acc;
        }
    jacobi3(FtF, w, V);
    const double v3[3] = { V[0], V[3], V[6] };
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double fv = F[r * 3] * v3[0] + F[r * 3 + 1] * v3[1] + F[r * 3 + 2] * v3[2];
#pragma unroll
        for (int c = 0; c < 3; c++) F[r * 3 + c] = F[r * 3 + c] - fv * v3[c];
    }
}

// Null vector of an 8x9 system: Gauss-Jordan with complete pivoting (same pivot order and arithmetic as the CPU
// restatement).  The 72 doubles of a lane live in LDS, element-major (s[(r*9+c)*64 + lane]): pivot positions differ
// from lane to lane, so the matrix is indexed dynamically -- in registers that costs a select chain per access
// (measured 6x slower); ds_read_b64 at [element][lane] is bank-conflict free whatever the element.
__device__ void null_8x9(double* __restrict__ s, int lane, double* f)
{
#define A_(r, c) s[((r) * 9 + (c)) * 64 + lane]
    unsigned used_r = 0, used_c = 0;
    int prow[8], pcol[8];
    bool ok = true;
    for (int k = 0; k < 8; k++) {
        double best = -1.0;
        int br = 0, bc = 0;
        for (int r = 0; r < 8; r++) {
            if ((used_r >> r) & 1u) continue;
            for (int c = 0; c < 9; c++) {
                if ((used_c >> c) & 1u) continue;
                const double v = fabs(A_(r, c));
                if (v > best) { best = v; br = r; bc = c; }
            }
        }
        used_r |= 1u << br; used_c |= 1u << bc; prow[k] = br; pcol[k] = bc;
        const double pv = A_(br, bc);
        if (pv == 0.0) { ok = false; break; }
        double piv[9];
#pragma unroll
        for (int c = 0; c < 9; c++) piv[c] = A_(br, c);
        for (int r = 0; r < 8; r++) {
            if (r == br) continue;
            const double m = A_(r, bc) / pv;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const double nv = A_(r, c) - m * piv[c];
                A_(r, c) = (c == bc) ? 0.0 : nv;
            }
        }
    }
    double fv[9];
#pragma unroll
    for (int c = 0; c < 9; c++) fv[c] = 0.0;
    if (ok) {
        int cf = 0;
        for (int c = 0; c < 9; c++) if (!((used_c >> c) & 1u)) cf = c;
#pragma unroll
        for (int c = 0; c < 9; c++) fv[c] = (c == cf) ? 1.0 : 0.0;
        for (int k = 0; k < 8; k++) {
            const double val = -A_(prow[k], cf) / A_(prow[k], pcol[k]);
#pragma unroll
            for (int c = 0; c < 9; c++) fv[c] = (c == pcol[k]) ? val : fv[c];
        }
        double nn = 0.0;
#pragma unroll
        for (int c = 0; c < 9; c++) nn += fv[c] * fv[c];
        nn = sqrt(nn);
#pragma unroll
        for (int c = 0; c < 9; c++) fv[c] = fv[c] / nn;
    }
#pragma unroll
    for (int c = 0; c < 9; c++) f[c] = fv[c];
#undef A_
}

// ------------------------------------------------------------------ fundamental matrix
struct FmatArgs {
    const double* x1; const double* x2; const int32_t* M; int capM;
    const int32_t* samples; int H; int B;
    int metric; double thr; double confidence; int refit;
    double* pts;      // [B][capM][4]: (u1,v1,u2,v2) in the scoring representation
    double* maxt;     // [B]
    double* models;   // [B][H][9] scoring F (scaled or pixel units by metric)
    double* models_s; // [B][H][9] F for the max-scaled coordinates (refit 0 de-normalises it)
    int32_t* counts;  // [B][H]
    int32_t* best;    // [B]
    double* F; uint8_t* status; int32_t* n_inliers;
    // hypotheses [h_lo, h_hi) of this launch; state (optional): {niters, best, maxgood} per problem, the adaptive bound
    // carried from one range to the next -- blocks past it have nothing to contribute and return
    int h_lo, h_hi; int32_t* state;
    // 7-point mode (mslam_fmat_ransac7): H counts matrices (3 per sample), models hold pixel-domain matrices, thr_b the inlier
    // threshold of each problem (RANSAC: thr; LMedS: from the least median; 7 pairs: +inf), method as the ABI's
    int mp7; int method; int H7; int lmeds_iters; double* thr_b; int32_t* best_cnt;
};

// which estimator cvFindFundamentalMat runs for a problem of M pairs: 1 RANSAC, 2 LMedS, 3 the 7 pairs solved directly
__device__ __forceinline__ int fm7_method(int method, int M)
{
    if (M == 7) return 3;
    if (method == 0) return M >= 15 ? 1 : 2;
    return method;
}

__global__ __launch_bounds__(256)
void fmat_prepare_k(FmatArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = a.M[b];
    const double* x1 = a.x1 + (size_t)b * a.capM * 2;
    const double* x2 = a.x2 + (size_t)b * a.capM * 2;
    __shared__ double s_max[256];
    double m = 0.0;
    for (int i = tid; i < M; i += 256) { // max is order independent (:609-613)
        m = fmax(m, fmax(x1[2 * i], x2[2 * i]));
        m = fmax(m, fmax(x1[2 * i + 1], x2[2 * i + 1]));
    }
    s_max[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_max[tid] = fmax(s_max[tid], s_max[tid + o]); __syncthreads(); }
    const double maxt = s_max[0];
    if (tid == 0) {
        a.maxt[b] = maxt;
        if (a.state) { a.state[3 * b] = a.mp7 ? (M >= 7 ? a.H7 : 0) : (M >= 8 ? a.H : 0); a.state[3 * b + 1] = -1; a.state[3 * b + 2] = 0; }
    }
    double* pts = a.pts + (size_t)b * a.capM * 4;
    for (int i = tid; i < M; i += 256) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        if (a.metric == 0) { // through float, as p_match (:31-34, :615-633)
            u1 = (double)(float)(u1 / maxt); v1 = (double)(float)(v1 / maxt);
            u2 = (double)(float)(u2 / maxt); v2 = (double)(float)(v2 / maxt);
        }
        pts[4 * i] = u1; pts[4 * i + 1] = v1; pts[4 * i + 2] = u2; pts[4 * i + 3] = v2;
    }
}

__global__ __launch_bounds__(64)
void fmat_solve_k(FmatArgs a)
{
    const int b = blockIdx.y;
    const int h = a.h_lo + blockIdx.x * 64 + threadIdx.x;
    if (a.state && a.h_lo + (int)blockIdx.x * 64 >= a.state[3 * b]) return; // past the adaptive bound (block-uniform)
    if (h >= a.h_hi) return;
    const int M = a.M[b];
    const double maxt = a.maxt[b];
    const double* x1 = a.x1 + (size_t)b * a.capM * 2;
    const double* x2 = a.x2 + (size_t)b * a.capM * 2;
    const int32_t* s = a.samples + ((size_t)b * a.H + h) * 8;
    __shared__ double s_A[72 * 64];
    const int lane = threadIdx.x;
    bool valid = M >= 8 && maxt > 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int i = s[k];
        if (i < 0 || i >= M) { valid = false; i = 0; }
        float u1 = (float)(x1[2 * i] / maxt), v1 = (float)(x1[2 * i + 1] / maxt);
        float u2 = (float)(x2[2 * i] / maxt), v2 = (float)(x2[2 * i + 1] / maxt);
        const double row[9] = { (double)(u2 * u1), (double)(u2 * v1), (double)u2, (double)(v2 * u1), (double)(v2 * v1), (double)v2,
                                (double)u1, (double)v1, 1.0 };
#pragma unroll
        for (int c = 0; c < 9; c++) s_A[(k * 9 + c) * 64 + lane] = row[c];
    }
    double F[9];
    null_8x9(s_A, lane, F);
    rank2(F);
    double* ms = a.models_s + ((size_t)b * a.H + h) * 9;
    double* mo = a.models + ((size_t)b * a.H + h) * 9;
    const double t[3] = { 1.0 / maxt, 1.0 / maxt, 1.0 };
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double v = valid ? F[i * 3 + j] : 0.0;
            ms[i * 3 + j] = v;
            mo[i * 3 + j] = a.metric == 0 ? v : t[i] * v * t[j]; // times_F^T F times_F (:666-669)
        }
}

// ---- 7-point minimal solver (OpenCV's run7Point restated; the CPU restatement evaluates the same operations in the same order)
__device__ __forceinline__ void cof3(const double* A, double* C)
{
    C[0] = A[4] * A[8] - A[5] * A[7]; C[1] = A[5] * A[6] - A[3] * A[8]; C[2] = A[3] * A[7] - A[4] * A[6];
    C[3] = A[2] * A[7] - A[1] * A[8]; C[4] = A[0] * A[8] - A[2] * A[6]; C[5] = A[1] * A[6] - A[0] * A[7];
    C[6] = A[1] * A[5] - A[2] * A[4]; C[7] = A[2] * A[3] - A[0] * A[5]; C[8] = A[0] * A[4] - A[1] * A[3];
}

__device__ double cubic_root(double a, double b, double c);

// real roots of c3 x^3 + c2 x^2 + c1 x + c0 in cvSolveCubic's order (three real roots: smallest, largest, middle): monotone
// Newton for one root, deflation in the stable direction, three Newton steps on the cubic for the other two.  + - * / sqrt only.
__device__ int solve_cubic(double c3, double c2, double c1, double c0, double* r)
{
    r[0] = r[1] = r[2] = 0.0;
    if (c3 == 0.0) {
        if (c2 == 0.0) {
            if (c1 == 0.0) return 0;
            r[0] = -c0 / c1;
            return 1;
        }
        double d = c1 * c1 - 4.0 * c2 * c0;
        if (!(d >= 0.0)) return 0;
        d = sqrt(d);
        const double q = 1.0 / (2.0 * c2);
        r[0] = (-c1 + d) * q;
        r[1] = (-c1 - d) * q;
        return 2;
    }
    const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
    const double x0 = cubic_root(a, b, c);
    double p, q;
    if (fabs(x0) * x0 * x0 > fabs(c)) { q = -c / x0; p = (q - b) / x0; }
    else { p = a + x0; q = b + p * x0; }
    const double disc = p * p - 4.0 * q;
    if (!(disc >= 0.0)) { r[0] = x0; return 1; }
    const double sq = sqrt(disc);
    const double h = -0.5 * (p + (p >= 0.0 ? sq : -sq));
    double x1 = h, x2 = h != 0.0 ? q / h : 0.0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        double x = k == 0 ? x1 : x2;
        for (int it = 0; it < 3; it++) {
            const double pv = ((x + a) * x + b) * x + c;
            const double dp = (3.0 * x + 2.0 * a) * x + b;
            if (dp == 0.0) break;
            x = x - pv / dp;
        }
        if (k == 0) x1 = x; else x2 = x;
    }
    double lo = x0, mid = x1, hi = x2, t;
    if (mid < lo) { t = lo; lo = mid; mid = t; }
    if (hi < lo) { t = lo; lo = hi; hi = t; }
    if (hi < mid) { t = mid; mid = hi; hi = t; }
    r[0] = lo; r[1] = hi; r[2] = mid;
    return 3;
}

// The 7 x 9 system of a lane lives in LDS, element-major (as null_8x9): Gauss-Jordan with complete pivoting leaves two free
// columns, i.e. a basis (f1, f2) of the null space; det(lambda (f1 - f2) + f2) = 0 gives up to three matrices, scaled to F[8] = 1.
__device__ int fmat_7point(double* __restrict__ s, int lane, double (*F)[9])
{
#define A_(r, c) s[((r) * 9 + (c)) * 64 + lane]
    unsigned used_r = 0, used_c = 0;
    int prow[7], pcol[7];
    for (int k = 0; k < 7; k++) {
        double best = -1.0;
        int br = 0, bc = 0;
        for (int r = 0; r < 7; r++) {
            if ((used_r >> r) & 1u) continue;
            for (int c = 0; c < 9; c++) {
                if ((used_c >> c) & 1u) continue;
                const double v = fabs(A_(r, c));
                if (v > best) { best = v; br = r; bc = c; }
            }
        }
        used_r |= 1u << br; used_c |= 1u << bc; prow[k] = br; pcol[k] = bc;
        const double pv = A_(br, bc);
        if (pv == 0.0) return 0;
        double piv[9];
#pragma unroll
        for (int c = 0; c < 9; c++) piv[c] = A_(br, c);
        for (int r = 0; r < 7; r++) {
            if (r == br) continue;
            const double m = A_(r, bc) / pv;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const double nv = A_(r, c) - m * piv[c];
                A_(r, c) = (c == bc) ? 0.0 : nv;
            }
        }
    }
    int cf[2] = { 0, 0 }, nf = 0;
    for (int c = 0; c < 9; c++)
        if (!((used_c >> c) & 1u)) { if (nf == 0) cf[0] = c; else cf[1] = c; nf++; }
    double f[2][9];
#pragma unroll
    for (int v = 0; v < 2; v++) {
        const int cfv = v == 0 ? cf[0] : cf[1];
#pragma unroll
        for (int c = 0; c < 9; c++) f[v][c] = (c == cfv) ? 1.0 : 0.0;
        for (int k = 0; k < 7; k++) {
            const double val = -A_(prow[k], cfv) / A_(prow[k], pcol[k]);
#pragma unroll
            for (int c = 0; c < 9; c++) f[v][c] = (c == pcol[k]) ? val : f[v][c];
        }
        double nn = 0.0;
#pragma unroll
        for (int c = 0; c < 9; c++) nn += f[v][c] * f[v][c];
        nn = sqrt(nn);
#pragma unroll
        for (int c = 0; c < 9; c++) f[v][c] = f[v][c] / nn;
    }
#undef A_
    double G[9], CG[9], C2[9];
#pragma unroll
    for (int c = 0; c < 9; c++) G[c] = f[0][c] - f[1][c];
    cof3(G, CG); cof3(f[1], C2);
    const double c3 = G[0] * CG[0] + G[1] * CG[1] + G[2] * CG[2];
    const double c0 = f[1][0] * C2[0] + f[1][1] * C2[1] + f[1][2] * C2[2];
    double c2 = 0.0, c1 = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) { c2 += CG[k] * f[1][k]; c1 += C2[k] * G[k]; }
    double roots[3];
    const int n = solve_cubic(c3, c2, c1, c0, roots);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double lambda = roots[k], mu = 1.0;
        const double sc = G[8] * roots[k] + f[1][8];
        double f8 = 0.0;
        if (fabs(sc) > DBL_EPSILON) { mu = 1.0 / sc; lambda = lambda * mu; f8 = 1.0; }
#pragma unroll
        for (int c = 0; c < 8; c++) F[k][c] = k < n ? G[c] * lambda + f[1][c] * mu : 0.0;
        F[k][8] = k < n ? f8 : 0.0;
    }
    return n;
}

__global__ __launch_bounds__(64)
void fmat7_solve_k(FmatArgs a)
{
    const int b = blockIdx.y;
    const int h = a.h_lo / 3 + blockIdx.x * 64 + threadIdx.x;   // (h_lo, h_hi count matrices: three per sample)
    if (a.state && a.h_lo / 3 + (int)blockIdx.x * 64 >= a.state[3 * b]) return; // past the adaptive bound (block-uniform)
    if (h >= a.h_hi / 3) return;
    const int M = a.M[b];
    const double* x1 = a.x1 + (size_t)b * a.capM * 2;
    const double* x2 = a.x2 + (size_t)b * a.capM * 2;
    const int32_t* s = a.samples + ((size_t)b * a.H7 + h) * 7;
    __shared__ double s_A[63 * 64];
    const int lane = threadIdx.x;
    bool valid = M >= 7;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        int i = s[k];
        if (i < 0 || i >= M) { valid = false; i = 0; }
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        const double row[9] = { u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, 1.0 };
#pragma unroll
        for (int c = 0; c < 9; c++) s_A[(k * 9 + c) * 64 + lane] = row[c];
    }
    double F[3][9];
    const int n = fmat_7point(s_A, lane, F);
    double* mo = a.models + ((size_t)b * a.H + 3 * h) * 9;
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int e = 0; e < 9; e++) mo[k * 9 + e] = (valid && k < n) ? F[k][e] : 0.0;   // all zero = no matrix
}

// OpenCV computeReprojError: max of the two squared point-to-epipolar-line distances, as float
__device__ __forceinline__ float fmat_err_f(const double* F, double u1, double v1, double u2, double v2)
{
    double aa = fma(F[0], u1, fma(F[1], v1, F[2]));
    double bb = fma(F[3], u1, fma(F[4], v1, F[5]));
    double cc = fma(F[6], u1, fma(F[7], v1, F[8]));
    double s2 = 1. / fma(aa, aa, bb * bb);
    double d2 = fma(u2, aa, fma(v2, bb, cc));
    aa = fma(F[0], u2, fma(F[3], v2, F[6]));
    bb = fma(F[1], u2, fma(F[4], v2, F[7]));
    cc = fma(F[2], u2, fma(F[5], v2, F[8]));
    double s1 = 1. / fma(aa, aa, bb * bb);
    double d1 = fma(u1, aa, fma(v1, bb, cc));
    double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// runLMeDS / the direct 7-pair case: one wavefront per problem, a lane takes a matrix, sorts its (at most FM7_LMEDS_MAXM) errors
// in LDS and keeps the least median; the winner is the first matrix with the least median, the inlier threshold follows from it.
// Problems that runRANSAC owns only get their threshold written.
constexpr int FM7_LMEDS_MAXM = 32;
__global__ __launch_bounds__(64)
void fmat7_lmeds_k(FmatArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int M = a.M[b];
    const int method = fm7_method(a.method, M);
    if (method == 1 || M < 7) {
        if (lane == 0) { a.thr_b[b] = a.thr; if (M < 7) a.best[b] = -1; }
        return;
    }
    const double* pts = a.pts + (size_t)b * a.capM * 4;
    if (method == 3) {   // exactly 7 pairs: the first matrix of sample 0, every pair is an inlier
        if (lane == 0) {
            bool any = false;
            for (int e = 0; e < 9; e++) any |= a.models[(size_t)b * a.H * 9 + e] != 0.0;
            a.best[b] = any ? 0 : -1;
            a.thr_b[b] = INFINITY;
        }
        return;
    }
    __shared__ float s_err[FM7_LMEDS_MAXM * 64];
    const int Mc = min(M, FM7_LMEDS_MAXM);   // (the host refuses LMedS for more pairs than this)
    double my_med = DBL_MAX;
    int my_best = -1;
    const int nmod = 3 * min(a.lmeds_iters, a.H7);
    for (int v0 = 0; v0 < nmod; v0 += 64) {
        const int v = v0 + lane;
        if (v < nmod) {
            double F[9];
            bool any = false;
#pragma unroll
            for (int e = 0; e < 9; e++) { F[e] = a.models[((size_t)b * a.H + v) * 9 + e]; any |= F[e] != 0.0; }
            if (any) {
                for (int i = 0; i < Mc; i++) {   // insertion sort, ascending; a NaN error sorts as +inf
                    float e = fmat_err_f(F, pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], pts[4 * i + 3]);
                    e = e == e ? e : INFINITY;
                    int j = i;
                    while (j > 0 && s_err[(j - 1) * 64 + lane] > e) { s_err[j * 64 + lane] = s_err[(j - 1) * 64 + lane]; j--; }
                    s_err[j * 64 + lane] = e;
                }
                const double median = (Mc & 1) ? (double)s_err[(Mc / 2) * 64 + lane]
                                               : (double)(s_err[(Mc / 2 - 1) * 64 + lane] + s_err[(Mc / 2) * 64 + lane]) * 0.5;
                if (median < my_med) { my_med = median; my_best = v; }
            }
        }
    }
    // first matrix with the least median: (median, index) lexicographic minimum over the lanes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(my_med, o);
        const int ob = __shfl_xor(my_best, o);
        const bool take = ob >= 0 && (my_best < 0 || om < my_med || (om == my_med && ob < my_best));
        if (take) { my_med = om; my_best = ob; }
    }
    if (lane == 0) {
        a.best[b] = my_best;
        double thr = a.thr;
        if (my_best >= 0) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (M - 7)) * sqrt(my_med);
            sigma = sigma > 0.001 ? sigma : 0.001;
            thr = sigma * sigma;
        }
        a.thr_b[b] = thr;
    }
}

__device__ __forceinline__ bool fmat_inlier(const double* F, double u1, double v1, double u2, double v2, int metric, double thr)
{
    if (metric == 0) { // getInlier (:715-763)
        double Fx1u = F[0] * u1 + F[1] * v1 + F[2];
        double Fx1v = F[3] * u1 + F[4] * v1 + F[5];
        double Fx1w = F[6] * u1 + F[7] * v1 + F[8];
        double Ftx2u = F[0] * u2 + F[3] * v2 + F[6];
        double Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
        double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
        double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
        return fabs(d) < thr;
    }
    double aa = fma(F[0], u1, fma(F[1], v1, F[2]));
    double bb = fma(F[3], u1, fma(F[4], v1, F[5]));
    double cc = fma(F[6], u1, fma(F[7], v1, F[8]));
    double s2 = 1. / fma(aa, aa, bb * bb);
    double d2 = fma(u2, aa, fma(v2, bb, cc));
    aa = fma(F[0], u2, fma(F[3], v2, F[6]));
    bb = fma(F[1], u2, fma(F[4], v2, F[7]));
    cc = fma(F[2], u2, fma(F[5], v2, F[8]));
    double s1 = 1. / fma(aa, aa, bb * bb);
    double d1 = fma(u1, aa, fma(v1, bb, cc));
    double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (double)(float)(e1 > e2 ? e1 : e2) <= thr; // err is stored as float by OpenCV
}

// Scoring (same structure as pnp_score_k): a workgroup owns FM_HPB hypotheses (F staged in LDS), each lane keeps
// FM_PT point pairs in registers.  For metric 1 a division-free band test (d^2 <= thr'*(a^2+b^2) on both epipolar
// lines, thr' = thr*(1+1e-5): a strict superset of the exact rule) skips the two f64 divisions of OpenCV's error
// measure unless some lane of the wavefront can be an inlier.
constexpr int FM_HPB = 32;
constexpr int FM_PT = 8;

__global__ __launch_bounds__(SCORE_THREADS)
void fmat_score_k(FmatArgs a)
{
    const int b = blockIdx.y, h0 = a.h_lo + blockIdx.x * FM_HPB, tid = threadIdx.x, lane = tid & 63;
    if (a.state && h0 >= a.state[3 * b] * (a.mp7 ? 3 : 1)) return; // past the adaptive bound (block-uniform; the bound counts samples)
    const int nh = min(FM_HPB, a.h_hi - h0);
    const int M = a.M[b];
    if (a.mp7 && fm7_method(a.method, M) != 1) return;   // LMedS / direct problems are not scored by counts
    __shared__ double s_F[FM_HPB][9];
    __shared__ int s_cnt[FM_HPB];
    for (int i = tid; i < FM_HPB * 9; i += SCORE_THREADS) {
        const int hh = i / 9, e = i % 9;
        s_F[hh][e] = (hh < nh) ? a.models[((size_t)b * a.H + h0 + hh) * 9 + e] : 0.0;
    }
    if (tid < FM_HPB) s_cnt[tid] = 0;
    __syncthreads();
    const double4* pts = reinterpret_cast<const double4*>(a.pts + (size_t)b * a.capM * 4);
    const double thr_lo = a.thr * (1.0 - 1e-5), thr_hi = a.thr * (1.0 + 1e-5);
    for (int c0 = 0; c0 < M; c0 += SCORE_THREADS * FM_PT) {
        double u1[FM_PT], v1[FM_PT], u2[FM_PT], v2[FM_PT];
        bool live[FM_PT];
        unsigned long long livem[FM_PT];
#pragma unroll
        for (int j = 0; j < FM_PT; j++) {
            const int i = c0 + j * SCORE_THREADS + tid;
            live[j] = i < M;
            livem[j] = __builtin_amdgcn_ballot_w64(live[j]);
            const double4 p = live[j] ? pts[i] : make_double4(0, 0, 0, 0);
            u1[j] = p.x; v1[j] = p.y; u2[j] = p.z; v2[j] = p.w;
        }
        for (int q = 0; q < nh; q++) {
            double F[9];
#pragma unroll
            for (int e = 0; e < 9; e++) F[e] = s_F[q][e];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < FM_PT; j++) {
                if (a.metric == 0) {
                    const bool in = live[j] && fmat_inlier(F, u1[j], v1[j], u2[j], v2[j], 0, a.thr);
                    cnt += __popcll(__ballot(in));
                } else {
                    // same operations as fmat_inlier (OpenCV computeReprojError), divisions deferred behind the band test
                    const double aa = fma(F[0], u1[j], fma(F[1], v1[j], F[2]));
                    const double bb = fma(F[3], u1[j], fma(F[4], v1[j], F[5]));
                    const double cc = fma(F[6], u1[j], fma(F[7], v1[j], F[8]));
                    const double n2 = fma(aa, aa, bb * bb);
                    const double d2 = fma(u2[j], aa, fma(v2[j], bb, cc));
                    const double a2 = fma(F[0], u2[j], fma(F[3], v2[j], F[6]));
                    const double b2 = fma(F[1], u2[j], fma(F[4], v2[j], F[7]));
                    const double c2 = fma(F[2], u2[j], fma(F[5], v2[j], F[8]));
                    const double n1 = fma(a2, a2, b2 * b2);
                    const double d1 = fma(u1[j], a2, fma(v1[j], b2, c2));
                    // d^2 / n against thr without the divisions: decided by the sign outside a 1e-5 relative shell (the
                    // exact rule differs from d^2 <= thr n only by the rounding of 1/n, two products and the float cast,
                    // < 1e-7 relative); the shell, NaNs and nothing else take the exact evaluation.
                    const double t1 = d1 * d1, t2 = d2 * d2;
                    const unsigned long long din = __builtin_amdgcn_ballot_w64(t1 < thr_lo * n1) & __builtin_amdgcn_ballot_w64(t2 < thr_lo * n2); // strict: n == 0 (no model) stays undecided
                    const unsigned long long dout = __builtin_amdgcn_ballot_w64(t1 > thr_hi * n1) | __builtin_amdgcn_ballot_w64(t2 > thr_hi * n2);
                    const unsigned long long un = livem[j] & ~(din | dout);
                    cnt += __popcll(din & livem[j]);
                    if (un) { // wave-uniform, rare
                        const double s2 = 1. / n2, s1 = 1. / n1;
                        const double e1 = t1 * s1, e2 = t2 * s2;
                        cnt += __popcll(__builtin_amdgcn_ballot_w64((double)(float)(e1 > e2 ? e1 : e2) <= a.thr) & un);
                    }
                }
            }
            if (lane == 0) atomicAdd(&s_cnt[q], cnt);
        }
    }
    __syncthreads();
    if (tid < nh) a.counts[(size_t)b * a.H + h0 + tid] = s_cnt[tid];
}

// cvRANSACUpdateNumIters
__device__ int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// one wavefront per problem: winner over the ordered counts (max count, lowest index) inside
// the adaptive iteration bound.  Chunks of 64 counts are skipped unless one beats the best.
__global__ __launch_bounds__(64)
void ransac_select_k(const int32_t* __restrict__ counts, const int32_t* __restrict__ npts, int H, int sample_size,
                     double confidence, int32_t* __restrict__ best_out, int32_t* __restrict__ best_cnt_out,
                     int h_lo, int h_hi, int32_t* __restrict__ state, int mpi, int stop_count)
{
    // stop_count > 0: solvePnPRansac's early exit -- the scan ends at the first record that reaches this many inliers
    // mpi: candidates per RANSAC iteration (1; 3 for the 7-point solver, whose samples yield up to three matrices: runRANSAC tests
    // its bound once per sample and takes every matrix of the sample as a candidate).  Indices and H count candidates.
    // state (optional): the scan continues over [h_lo, h_hi) from where the previous range left {niters, best, maxgood}
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t* c = counts + (size_t)b * H;
    const int n_points = npts[b];
    const bool adaptive = confidence > 0. && confidence < 1.;
    int niters = H / mpi, best = -1, maxgood = 0;
    if (state) { niters = state[3 * b]; best = state[3 * b + 1]; maxgood = state[3 * b + 2]; }
    // (eight chunks of 64 counts are fetched together: one memory latency per 512 counts instead of one per 64)
    constexpr int SEL_PF = 8;
    for (int g0 = h_lo; g0 < niters * mpi && g0 < h_hi; g0 += 64 * SEL_PF) {
      int vv[SEL_PF];
#pragma unroll
      for (int u = 0; u < SEL_PF; u++) { const int i = g0 + 64 * u + lane; vv[u] = i < h_hi ? c[i] : -1; }
#pragma unroll
      for (int u = 0; u < SEL_PF; u++) {
        const int i0 = g0 + 64 * u;
        if (!(i0 < niters * mpi && i0 < h_hi)) break;
        const int i = i0 + lane;
        const int v = vv[u];
        int floor_ = maxgood > sample_size - 1 ? maxgood : sample_size - 1;
        unsigned long long bal = __ballot(i < niters * mpi && v > floor_);
        while (bal) { // record breakers in index order
            const int j = __ffsll((long long)bal) - 1;
            bal &= bal - 1;
            const int it = i0 + j;
            if (it >= niters * mpi) break;
            const int good = __shfl(v, j);
            floor_ = maxgood > sample_size - 1 ? maxgood : sample_size - 1;
            if (good > floor_) {
                best = it; maxgood = good;
                if (adaptive && n_points > 0)
                    niters = ransac_update_iters(confidence, (double)(n_points - good) / n_points, sample_size, niters);
                if (stop_count > 0 && maxgood >= stop_count) niters = min(niters, it / mpi + 1);   // nothing after this iteration is looked at
            }
        }
      }
    }
    if (lane == 0) {
        best_out[b] = best;
        if (best_cnt_out) best_cnt_out[b] = maxgood;
        if (state) { state[3 * b] = niters; state[3 * b + 1] = best; state[3 * b + 2] = maxgood; }
    }
}

// block-wide (NT threads) sum of NV doubles per thread in a fixed order (deterministic, no atomics): values are
// parked in LDS (s_red: NV * NT doubles), value q is summed by wave q % NW (each lane adds NT/64 entries, then a
// butterfly over the wave).  Two barriers per reduction.
template <int NV, int NT>
__device__ void block_reduce(double (&v)[NV], double* s_red, int tid)
{
    constexpr int NW = NT / 64;
#pragma unroll
    for (int k = 0; k < NV; k++) s_red[k * NT + tid] = v[k];
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int q = wave; q < NV; q += NW) {
        const double* r = s_red + q * NT;
        double t = r[lane];
#pragma unroll
        for (int w = 1; w < NW; w++) t += r[lane + 64 * w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) s_red[q * NT] = t; // row q is only touched by this wave
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = s_red[k * NT];
    __syncthreads();
}

__global__ __launch_bounds__(256)
void fmat_finalize_k(FmatArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = a.M[b];
    const int best = a.best[b];
    uint8_t* status = a.status + (size_t)b * a.capM;
    double* Fout = a.F + (size_t)b * 9;
    __shared__ double s_red[9 * 256];
    __shared__ double s_A[81], s_V[81], s_d[9], s_F[9], s_bb[9], s_z[9];
    __shared__ int s_n;
    if (best < 0) {
        for (int i = tid; i < M; i += 256) status[i] = 0;
        if (tid < 9) Fout[tid] = 0.0;
        if (tid == 0) a.n_inliers[b] = 0;
        return;
    }
    if (tid < 9) s_F[tid] = a.models[((size_t)b * a.H + best) * 9 + tid];
    if (tid == 0) s_n = 0;
    __syncthreads();
    const double4* pts = reinterpret_cast<const double4*>(a.pts + (size_t)b * a.capM * 4);
    const double* x1 = a.x1 + (size_t)b * a.capM * 2;
    const double* x2 = a.x2 + (size_t)b * a.capM * 2;
    const double maxt = a.maxt[b];
    const double thr = a.thr_b ? a.thr_b[b] : a.thr;
    int my = 0;
    for (int i = tid; i < M; i += 256) {
        double4 p = pts[i];
        bool in = fmat_inlier(s_F, p.x, p.y, p.z, p.w, a.metric, thr);
        status[i] = (uint8_t)in;
        my += in;
    }
    atomicAdd(&s_n, my);
    __syncthreads();
    const int n = s_n;
    if (a.mp7 && n < 7) {   // runLMeDS: result = count >= modelPoints
        for (int i = tid; i < M; i += 256) status[i] = 0;
        if (tid < 9) Fout[tid] = 0.0;
        if (tid == 0) { a.n_inliers[b] = 0; a.best[b] = -1; }
        return;
    }
    if (tid == 0) a.n_inliers[b] = n;
    if (a.refit == 0 || n < 8) {
        if (tid < 9) {
            const double t[3] = { 1.0 / maxt, 1.0 / maxt, 1.0 };
            Fout[tid] = a.mp7 ? s_F[tid] : t[tid / 3] * a.models_s[((size_t)b * a.H + best) * 9 + tid] * t[tid % 3];
        }
        return;
    }
    // ---- refit on the inliers: accumulate A^T A (45 unique sums; upper triangle), Jacobi on one lane
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0, s0 = 1, s1 = 1;
    if (a.refit == 2) { // Hartley normalisation (run8Point)
        double acc[4] = { 0, 0, 0, 0 };
        for (int i = tid; i < M; i += 256)
            if (status[i]) { acc[0] += x1[2 * i]; acc[1] += x1[2 * i + 1]; acc[2] += x2[2 * i]; acc[3] += x2[2 * i + 1]; }
        block_reduce<4, 256>(acc, s_red, tid);
        const double t = 1. / n;
        c0x = acc[0] * t; c0y = acc[1] * t; c1x = acc[2] * t; c1y = acc[3] * t;
        double sc[2] = { 0, 0 };
        for (int i = tid; i < M; i += 256)
            if (status[i]) {
                double dx = x1[2 * i] - c0x, dy = x1[2 * i + 1] - c0y;
                sc[0] += sqrt(dx * dx + dy * dy);
                dx = x2[2 * i] - c1x; dy = x2[2 * i + 1] - c1y;
                sc[1] += sqrt(dx * dx + dy * dy);
            }
        block_reduce<2, 256>(sc, s_red, tid);
        s0 = sc[0] * t; s1 = sc[1] * t;
        if (s0 < (double)FLT_EPSILON || s1 < (double)FLT_EPSILON) { if (tid < 9) Fout[tid] = 0.0; return; }
        s0 = sqrt(2.) / s0; s1 = sqrt(2.) / s1;
    }
    for (int r0 = 0; r0 < 9; r0++) { // 9 passes, one row of the upper triangle each (keeps LDS small)
        double acc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k] = 0.0;
        for (int i = tid; i < M; i += 256) {
            if (!status[i]) continue;
            double r[9];
            if (a.refit == 2) {
                double X0 = (x1[2 * i] - c0x) * s0, Y0 = (x1[2 * i + 1] - c0y) * s0;
                double X1 = (x2[2 * i] - c1x) * s1, Y1 = (x2[2 * i + 1] - c1y) * s1;
                r[0] = X1 * X0; r[1] = X1 * Y0; r[2] = X1; r[3] = Y1 * X0; r[4] = Y1 * Y0; r[5] = Y1; r[6] = X0; r[7] = Y0; r[8] = 1;
            } else {
                float u1 = (float)(x1[2 * i] / maxt), v1 = (float)(x1[2 * i + 1] / maxt);
                float u2 = (float)(x2[2 * i] / maxt), v2 = (float)(x2[2 * i + 1] / maxt);
                r[0] = (double)(u2 * u1); r[1] = (double)(u2 * v1); r[2] = (double)u2; r[3] = (double)(v2 * u1);
                r[4] = (double)(v2 * v1); r[5] = (double)v2; r[6] = (double)u1; r[7] = (double)v1; r[8] = 1.0;
            }
            double rr = r[0];
#pragma unroll
            for (int k = 1; k < 9; k++) rr = (k == r0) ? r[k] : rr;
#pragma unroll
            for (int k = 0; k < 9; k++) acc[k] += rr * r[k];
        }
        block_reduce<9, 256>(acc, s_red, tid);
        if (tid == 0)
            for (int k = 0; k < 9; k++) s_A[r0 * 9 + k] = acc[k];
        __syncthreads();
    }
    jacobi_block(s_A, 9, s_d, s_V, s_bb, s_z, tid); // reads the upper triangle only
    if (tid == 0) {
        const int k0 = argmin_first(s_d, 9);
        double F0[9];
        for (int e = 0; e < 9; e++) F0[e] = s_V[e * 9 + k0];
        bool okf = true;
        if (a.refit == 2) { // at most one numerically zero eigenvalue
            int nz = 0;
            for (int e = 0; e < 9; e++) nz += fabs(s_d[e]) < DBL_EPSILON;
            okf = nz <= 1;
        }
        rank2(F0);
        double R[9];
        if (a.refit == 2) {
            const double T0[9] = { s0, 0, -s0 * c0x, 0, s0, -s0 * c0y, 0, 0, 1 };
            const double T1[9] = { s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1 };
            double tmp[9];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double acc = 0; for (int k = 0; k < 3; k++) acc += T1[k * 3 + i] * F0[k * 3 + j]; tmp[i * 3 + j] = acc; }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double acc = 0; for (int k = 0; k < 3; k++) acc += tmp[i * 3 + k] * T0[k * 3 + j]; R[i * 3 + j] = acc; }
            if (fabs(R[8]) > (double)FLT_EPSILON) { double sc = 1. / R[8]; for (int e = 0; e < 9; e++) R[e] *= sc; }
        } else {
            const double t[3] = { 1.0 / maxt, 1.0 / maxt, 1.0 };
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = t[i] * F0[i * 3 + j] * t[j];
        }
        for (int e = 0; e < 9; e++) Fout[e] = okf ? R[e] : 0.0;
    }
}

// ------------------------------------------------------------------ PnP
struct PnpArgs {
    const double* X; const double* x; const int32_t* P; int capP;
    const double* K; const double* dist; const int32_t* samples; int H; int B;
    double thr, thr2cut, confidence; int refine_iters, min_inliers;
    double* xn;       // [B][capP][2] undistorted normalised image points
    double* models;   // [B][H][12] hypothesis poses (row-major 3x4); invalid = all zero
    int32_t* counts;  // [B][H]
    int32_t* best; int32_t* best_cnt;
    double* Rt; double* rvec; double* tvec; uint8_t* inliers; int32_t* n_inliers; int32_t* status;
    int f32_tier;     // 1: pnp_score_k decides in single precision first (MSLAM_PNP_F32=0 keeps the double test alone)
};

__device__ __forceinline__ void project_pt(const double* Rt, double fx, double fy, double cx, double cy, const double* k,
                                           bool has_dist, double X, double Y, double Z, double& u, double& v)
{
    // cvProjectPoints2; R X + t and the pixel mapping as explicit fma chains (same chains in the CPU restatement)
    double x = fma(Rt[0], X, fma(Rt[1], Y, fma(Rt[2], Z, Rt[3])));
    double y = fma(Rt[4], X, fma(Rt[5], Y, fma(Rt[6], Z, Rt[7])));
    double z = fma(Rt[8], X, fma(Rt[9], Y, fma(Rt[10], Z, Rt[11])));
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    double xd = x, yd = y;
    if (has_dist) {
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
        yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
    }
    u = fma(xd, fx, cx);
    v = fma(yd, fy, cy);
}

__global__ __launch_bounds__(256)
void pnp_prepare_k(PnpArgs a)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P[b]) return;
    const double* K = a.K + b * 9;
    const double* k = a.dist + b * 8;
    const double* x = a.x + ((size_t)b * a.capP + i) * 2;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double x0 = (x[0] - cx) / fx, y0 = (x[1] - cy) / fy;
    double xx = x0, yy = y0;
    for (int it = 0; it < 10; it++) { // cvUndistortPoints' fixed-point iteration
        double r2 = xx * xx + yy * yy;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double dX = 2 * k[2] * xx * yy + k[3] * (r2 + 2 * xx * xx);
        double dY = k[2] * (r2 + 2 * yy * yy) + 2 * k[3] * xx * yy;
        xx = (x0 - dX) * icdist;
        yy = (y0 - dY) * icdist;
    }
    double* o = a.xn + ((size_t)b * a.capP + i) * 2;
    o[0] = xx; o[1] = yy;
}

// real root of x^3 + a x^2 + b x + c: monotone Newton from beyond the extreme root
__device__ double cubic_root(double a, double b, double c)
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

__device__ __forceinline__ double quad_form(const double* D, const double* p, const double* q)
{
    double Dq0 = D[0] * q[0] + D[1] * q[1] + D[2] * q[2];
    double Dq1 = D[3] * q[0] + D[4] * q[1] + D[5] * q[2];
    double Dq2 = D[6] * q[0] + D[7] * q[1] + D[8] * q[2];
    return p[0] * Dq0 + p[1] * Dq1 + p[2] * Dq2;
}

__device__ __forceinline__ double sel3(const double* v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }

// P3P, up to 4 poses.  The three distance constraints  l_i^2 + l_j^2 - 2 (f_i.f_j) l_i l_j = |X_i - X_j|^2  span a pencil
// of conics; a degenerate member (a real root of a cubic, by monotone Newton iteration) splits into two planes (3x3 Jacobi
// eigen-decomposition), each intersected with another member of the pencil (a quadratic); depths are polished by 5
// Gauss-Newton rounds and the pose follows from the two point triads.  Only + - * / sqrt: reproducible operation for operation.
__device__ int p3p(const double* f, const double* X, double (*Rout)[9], double (*tout)[3])
{
    const double* f0 = f; const double* f1 = f + 3; const double* f2 = f + 6;
    double b[3], a[3];
    b[0] = f0[0] * f1[0] + f0[1] * f1[1] + f0[2] * f1[2];
    b[1] = f0[0] * f2[0] + f0[1] * f2[1] + f0[2] * f2[2];
    b[2] = f1[0] * f2[0] + f1[1] * f2[1] + f1[2] * f2[2];
    double d01[3], d02[3], d12[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { d01[k] = X[k] - X[3 + k]; d02[k] = X[k] - X[6 + k]; d12[k] = X[3 + k] - X[6 + k]; }
    a[0] = d01[0] * d01[0] + d01[1] * d01[1] + d01[2] * d01[2];
    a[1] = d02[0] * d02[0] + d02[1] * d02[1] + d02[2] * d02[2];
    a[2] = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    if (!(a[0] > 0 && a[1] > 0 && a[2] > 0)) return 0;
    const double D1[9] = { a[2], -a[2] * b[0], 0, -a[2] * b[0], a[2] - a[0], a[0] * b[2], 0, a[0] * b[2], -a[0] };
    const double D2[9] = { a[2], 0, -a[2] * b[1], 0, -a[1], a[1] * b[2], -a[2] * b[1], a[1] * b[2], a[2] - a[1] };
    double J1[9], J2[9];
    adj3s(D1, J1); adj3s(D2, J2);
    double c0 = det3(D1), c3 = det3(D2), c1 = 0, c2 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) { c1 += J1[k] * D2[k]; c2 += D1[k] * J2[k]; }
    double cs = fabs(c0) + fabs(c1) + fabs(c2) + fabs(c3);
    if (!(fabs(c3) > 1e-12 * cs)) return 0;
    const double ca = c2 / c3, cb = c1 / c3, cc = c0 / c3;
    double roots[3];
    int nroots = 1;
    roots[0] = cubic_root(ca, cb, cc);
    {
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
        const double g = ri == 0 ? roots[0] : (ri == 1 ? roots[1] : roots[2]);
        double D0[9], e[3], E[9];
#pragma unroll
        for (int k = 0; k < 9; k++) D0[k] = D1[k] + g * D2[k];
        jacobi3(D0, e, E);
        if (!(e[0] < 0.0 && e[2] > 0.0) || !(fabs(e[1]) <= fabs(e[0]) && fabs(e[1]) <= fabs(e[2]))) continue;
        const double s = sqrt(-e[0] / e[2]);
        const bool useD2 = fabs(g) <= 1.0;
        double D[9];
#pragma unroll
        for (int k = 0; k < 9; k++) D[k] = useD2 ? D2[k] : D1[k];
        for (int sign = 0; sign < 2; sign++) {
            double w[3];
#pragma unroll
            for (int k = 0; k < 3; k++) w[k] = E[k * 3 + 2] + (sign ? -s : s) * E[k * 3 + 0];
            int kk = 0;
            if (fabs(w[1]) > fabs(sel3(w, kk))) kk = 1;
            if (fabs(w[2]) > fabs(sel3(w, kk))) kk = 2;
            const double wk = sel3(w, kk);
            if (wk == 0.0) continue;
            const int ii = kk == 0 ? 1 : 0, jj = kk == 2 ? 1 : 2;
            const double al = -sel3(w, ii) / wk, be = -sel3(w, jj) / wk;
            double p[3], q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { p[k] = (k == ii) ? 1.0 : ((k == kk) ? al : 0.0); q[k] = (k == jj) ? 1.0 : ((k == kk) ? be : 0.0); }
            const double A = quad_form(D, p, p), B = 2.0 * quad_form(D, p, q), C = quad_form(D, q, q);
            double taus[2] = { 0.0, 0.0 };
            int nt = 0;
            if (A == 0.0) { if (B != 0.0) { taus[0] = -C / B; nt = 1; } }
            else {
                double disc = B * B - 4.0 * A * C;
                if (disc >= 0.0) {
                    double sq = sqrt(disc);
                    double h = -0.5 * (B + (B >= 0.0 ? sq : -sq));
                    taus[0] = h / A; nt = 1;
                    if (h != 0.0) { taus[1] = C / h; nt = 2; }
                }
            }
            const int pr = ii + jj - 1;
            const double bij = sel3(b, pr), aij = sel3(a, pr);
            for (int ti = 0; ti < nt && nsol < 4; ti++) {
                const double tau = ti == 0 ? taus[0] : taus[1];
                if (!(tau > 0.0)) continue;
                const double den = tau * tau + 1.0 - 2.0 * bij * tau;
                if (!(den > 0.0)) continue;
                const double Lj = sqrt(aij / den), Li = tau * Lj, Lk = al * Li + be * Lj;
                if (!(Lk > 0.0)) continue;
                double L[3];
#pragma unroll
                for (int k = 0; k < 3; k++) L[k] = (k == ii) ? Li : ((k == jj) ? Lj : Lk);
                for (int it = 0; it < 5; it++) {
                    double r0 = L[0] * L[0] + L[1] * L[1] - 2.0 * b[0] * L[0] * L[1] - a[0];
                    double r1 = L[0] * L[0] + L[2] * L[2] - 2.0 * b[1] * L[0] * L[2] - a[1];
                    double r2 = L[1] * L[1] + L[2] * L[2] - 2.0 * b[2] * L[1] * L[2] - a[2];
                    const double Jm[9] = { 2.0 * L[0] - 2.0 * b[0] * L[1], 2.0 * L[1] - 2.0 * b[0] * L[0], 0.0,
                                           2.0 * L[0] - 2.0 * b[1] * L[2], 0.0, 2.0 * L[2] - 2.0 * b[1] * L[0],
                                           0.0, 2.0 * L[1] - 2.0 * b[2] * L[2], 2.0 * L[2] - 2.0 * b[2] * L[1] };
                    double Ja[9];
                    adj3s(Jm, Ja);
                    double det = det3(Jm);
                    if (det == 0.0) break;
                    double id = 1.0 / det;
                    L[0] -= (Ja[0] * r0 + Ja[1] * r1 + Ja[2] * r2) * id;
                    L[1] -= (Ja[3] * r0 + Ja[4] * r1 + Ja[5] * r2) * id;
                    L[2] -= (Ja[6] * r0 + Ja[7] * r1 + Ja[8] * r2) * id;
                }
                if (!(L[0] > 0.0 && L[1] > 0.0 && L[2] > 0.0)) continue;
                double Y[9];
#pragma unroll
                for (int k = 0; k < 3; k++) { Y[k] = L[0] * f0[k]; Y[3 + k] = L[1] * f1[k]; Y[6 + k] = L[2] * f2[k]; }
                double y1[3], y2[3], y3[3], x3[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { y1[k] = Y[k] - Y[3 + k]; y2[k] = Y[k] - Y[6 + k]; }
                y3[0] = y1[1] * y2[2] - y1[2] * y2[1]; y3[1] = y1[2] * y2[0] - y1[0] * y2[2]; y3[2] = y1[0] * y2[1] - y1[1] * y2[0];
                x3[0] = d01[1] * d02[2] - d01[2] * d02[1]; x3[1] = d01[2] * d02[0] - d01[0] * d02[2]; x3[2] = d01[0] * d02[1] - d01[1] * d02[0];
                const double Xm[9] = { d01[0], d02[0], x3[0], d01[1], d02[1], x3[1], d01[2], d02[2], x3[2] };
                const double Ym[9] = { y1[0], y2[0], y3[0], y1[1], y2[1], y3[1], y1[2], y2[2], y3[2] };
                double Xa[9];
                adj3s(Xm, Xa);
                double dx = det3(Xm);
                if (dx == 0.0) continue;
                double idx_ = 1.0 / dx;
                double R[9], tt[3];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        R[i * 3 + j] = (Ym[i * 3] * Xa[j] + Ym[i * 3 + 1] * Xa[3 + j] + Ym[i * 3 + 2] * Xa[6 + j]) * idx_;
#pragma unroll
                for (int i = 0; i < 3; i++) tt[i] = Y[i] - (R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2]);
                // nsol is a small runtime index: write through a switch-free select loop
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (c == nsol) {
#pragma unroll
                        for (int e = 0; e < 9; e++) Rout[c][e] = R[e];
#pragma unroll
                        for (int e = 0; e < 3; e++) tout[c][e] = tt[e];
                    }
                nsol++;
            }
        }
    }
    return nsol;
}

__global__ __launch_bounds__(64, 2)   // <= 256 VGPRs: two wavefronts per SIMD cover each other's f64 divide / sqrt chains (288 VGPRs = one)
void pnp_solve_k(PnpArgs a)
{
    const int b = blockIdx.y;
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= a.H) return;
    const int P = a.P[b];
    const double* Xw = a.X + (size_t)b * a.capP * 3;
    const double* x = a.x + (size_t)b * a.capP * 2;
    const double* xn = a.xn + (size_t)b * a.capP * 2;
    const double* K = a.K + b * 9;
    const double* k = a.dist + b * 8;
    const int32_t* s = a.samples + ((size_t)b * a.H + h) * 4;
    int si[4];
    bool valid = P >= 4;
#pragma unroll
    for (int j = 0; j < 4; j++) { si[j] = s[j]; if (si[j] < 0 || si[j] >= P) { valid = false; si[j] = 0; } }
    double f[9], Xs[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double bx = xn[2 * si[j]], by = xn[2 * si[j] + 1];
        double inv = 1.0 / sqrt(bx * bx + by * by + 1.0);
        f[3 * j] = bx * inv; f[3 * j + 1] = by * inv; f[3 * j + 2] = inv;
        Xs[3 * j] = Xw[3 * si[j]]; Xs[3 * j + 1] = Xw[3 * si[j] + 1]; Xs[3 * j + 2] = Xw[3 * si[j] + 2];
    }
    double R[4][9], t[4][3];
    const int n = valid ? p3p(f, Xs, R, t) : 0;
    const bool has_dist = (k[0] != 0.0) | (k[1] != 0.0) | (k[2] != 0.0) | (k[3] != 0.0) | (k[4] != 0.0) | (k[5] != 0.0) | (k[6] != 0.0) | (k[7] != 0.0);
    double bestRt[12];
#pragma unroll
    for (int e = 0; e < 12; e++) bestRt[e] = 0.0;
    bool have = false;
    double beste = 0.0;
    const double X4 = Xw[3 * si[3]], Y4 = Xw[3 * si[3] + 1], Z4 = Xw[3 * si[3] + 2];
    const double ox = x[2 * si[3]], oy = x[2 * si[3] + 1];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (c < n) {
            double cand[12];
#pragma unroll
            for (int i = 0; i < 3; i++) { cand[i * 4] = R[c][i * 3]; cand[i * 4 + 1] = R[c][i * 3 + 1]; cand[i * 4 + 2] = R[c][i * 3 + 2]; cand[i * 4 + 3] = t[c][i]; }
            double u, v;
            project_pt(cand, K[0], K[4], K[2], K[5], k, has_dist, X4, Y4, Z4, u, v);
            double dx = u - ox, dy = v - oy;
            double e = fma(dx, dx, dy * dy);
            if (e == e && (!have || e < beste)) {
                have = true; beste = e;
#pragma unroll
                for (int q = 0; q < 12; q++) bestRt[q] = cand[q];
            }
        }
    }
    double* mo = a.models + ((size_t)b * a.H + h) * 12;
#pragma unroll
    for (int e = 0; e < 12; e++) mo[e] = bestRt[e]; // all zero = no hypothesis
}

// Scoring: a workgroup owns PNP_HPB hypotheses (poses staged in LDS) and its lanes keep PNP_PT points each in
// registers (256 x 8 = 2048 points per pass), so a point is loaded once per workgroup and every hypothesis is
// evaluated against PNP_PT independent projections per lane (ILP hides the f64 divide latency).  Votes are
// __ballot + popcount per point slot, summed per hypothesis in LDS.
constexpr int PNP_HPB = 32;
constexpr int PNP_PT = 8;

template <bool HAS_DIST>
__device__ __forceinline__ void pnp_score_body(const PnpArgs& a, int b, int h0, int nh, int tid, int lane, const double (*s_Rt)[12],
                                               const double (*s_Rs)[8], const double* s_tn, const int* s_valid, int* s_cnt)
{
    const int P = a.P[b];
    const double* K = a.K + b * 9;
    const double* k = a.dist + b * 8;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double* Xw = a.X + (size_t)b * a.capP * 3;
    const double* x = a.x + (size_t)b * a.capP * 2;
    // Decision shell of the division-free test (see below).  A threshold too small for the error analysis to hold
    // (or not finite) makes eps infinite/NaN, which sends every point down the exact path.
    const double inv_thr = 1.0 / a.thr;
    const double eps = (a.thr >= 1e-3 && a.thr <= 1e6) ? 1e-6 : __longlong_as_double(0x7ff0000000000000ll);
    const double sf = (fabs(fx) > fabs(fy) ? fabs(fx) : fabs(fy)) * inv_thr;
    for (int c0 = 0; c0 < P; c0 += SCORE_THREADS * PNP_PT) {
        double X[PNP_PT], Y[PNP_PT], Z[PNP_PT], ox[PNP_PT], oy[PNP_PT], ocx[PNP_PT], ocy[PNP_PT];
        bool live[PNP_PT];
        unsigned long long livem[PNP_PT];
        double nrm = 0.0;
#pragma unroll
        for (int j = 0; j < PNP_PT; j++) {
            const int i = c0 + j * SCORE_THREADS + tid;
            live[j] = i < P;
            livem[j] = __builtin_amdgcn_ballot_w64(live[j]);
            const int ic = live[j] ? i : 0;
            X[j] = Xw[3 * ic]; Y[j] = Xw[3 * ic + 1]; Z[j] = Xw[3 * ic + 2];
            ox[j] = x[2 * ic]; oy[j] = x[2 * ic + 1];
            ocx[j] = (ox[j] - cx) * inv_thr; ocy[j] = (oy[j] - cy) * inv_thr; // threshold units
            nrm = fmax(nrm, fabs(X[j]) + fabs(Y[j]) + fabs(Z[j]));
        }
        for (int q = 0; q < nh; q++) {
            if (!s_valid[q]) continue; // wave-uniform
            int cnt = 0;
            if (HAS_DIST) {
                double Rt[12];
#pragma unroll
                for (int e = 0; e < 12; e++) Rt[e] = s_Rt[q][e];
#pragma unroll
                for (int j = 0; j < PNP_PT; j++) {
                    double u, v;
                    project_pt(Rt, fx, fy, cx, cy, k, true, X[j], Y[j], Z[j], u, v);
                    const double dx = ox[j] - u, dy = oy[j] - v;
                    // sqrt(d2) < thr  <=>  d2 < thr2cut  (thr2cut = smallest double whose sqrt is >= thr)
                    cnt += __popcll(__builtin_amdgcn_ballot_w64(fma(dx, dx, dy * dy) < a.thr2cut) & livem[j]);
                }
            } else {
                // Without distortion the reprojection error in threshold units is (A, B) / pz with
                //   A = (fx/thr) (R0.X + t0) - ((ox-cx)/thr) pz,   B likewise in y,
                // so  e = A^2 + B^2 - pz^2  is negative for an inlier.  Outside the shell |e| <= 1e-6 pz^2 the sign of e
                // decides (the exact rule's own rounding is ~1e-12 relative there, provided pz is not tiny against the
                // magnitudes it was formed from: |pz| >= 1e-7 (f/thr)(|X|_1 + |t|_1), which bounds the cancellation
                // error of A/pz by ~1e-7).  Points inside the shell, with a tiny pz, or producing NaN take the exact
                // cvProjectPoints2 evaluation (same operations as project_pt); they are ~1e-6 of all points.
                double Rs[8], Rz[4];
#pragma unroll
                for (int e = 0; e < 8; e++) Rs[e] = s_Rs[q][e];
#pragma unroll
                for (int e = 0; e < 4; e++) Rz[e] = s_Rt[q][8 + e];
                const double zg = 1e-7 * sf * (nrm + s_tn[q]) + 1e-300;
                unsigned long long um[PNP_PT], any = 0; // undecided lanes per point slot
#pragma unroll
                for (int j = 0; j < PNP_PT; j++) {
                    const double pxs = fma(Rs[0], X[j], fma(Rs[1], Y[j], fma(Rs[2], Z[j], Rs[3])));
                    const double pys = fma(Rs[4], X[j], fma(Rs[5], Y[j], fma(Rs[6], Z[j], Rs[7])));
                    const double pz = fma(Rz[0], X[j], fma(Rz[1], Y[j], fma(Rz[2], Z[j], Rz[3]))); // == project_pt's z
                    const double A = fma(-ocx[j], pz, pxs), Bv = fma(-ocy[j], pz, pys);
                    const double pz2 = pz * pz;
                    const double e = fma(A, A, fma(Bv, Bv, -pz2));
                    const double m = eps * pz2;
                    const unsigned long long un = (__builtin_amdgcn_ballot_w64(!(fabs(e) > m)) | __builtin_amdgcn_ballot_w64(fabs(pz) < zg)) & livem[j];
                    const unsigned long long in = __builtin_amdgcn_ballot_w64(e < -m) & livem[j] & ~un;
                    cnt += __popcll(in);
                    um[j] = un; any |= un;
                }
                if (any) { // wave-uniform, rare
                    const double* Rt = s_Rt[q];
#pragma unroll
                    for (int j = 0; j < PNP_PT; j++) {
                        if (!um[j]) continue;
                        const double px = fma(Rt[0], X[j], fma(Rt[1], Y[j], fma(Rt[2], Z[j], Rt[3])));
                        const double py = fma(Rt[4], X[j], fma(Rt[5], Y[j], fma(Rt[6], Z[j], Rt[7])));
                        const double pz = fma(Rt[8], X[j], fma(Rt[9], Y[j], fma(Rt[10], Z[j], Rt[11])));
                        const double z = pz ? 1. / pz : 1;
                        const double u = fma(px * z, fx, cx), v = fma(py * z, fy, cy);
                        const double dx = ox[j] - u, dy = oy[j] - v;
                        cnt += __popcll(__builtin_amdgcn_ballot_w64(fma(dx, dx, dy * dy) < a.thr2cut) & um[j]);
                    }
                }
            }
            if (lane == 0) atomicAdd(&s_cnt[q], cnt);
        }
    }
}

__device__ __forceinline__ bool pnp_has_dist(const double* k)
{
    return (k[0] != 0.0) | (k[1] != 0.0) | (k[2] != 0.0) | (k[3] != 0.0) | (k[4] != 0.0) | (k[5] != 0.0) | (k[6] != 0.0) | (k[7] != 0.0);
}

// ---- no-distortion scoring with a single-precision first tier -------------------------------------------------------
// The division-free test above costs 18 double-precision instructions per (hypothesis, point), each of which takes a SIMD's issue
// slot alone (4 cycles per wave instruction).  A packed single-precision instruction (v_pk_fma_f32) takes the same slot but
// carries two points: 8 packed instructions + 2 compares per pair here.  (Plain v_fma_f32 issues at twice the rate when two
// wavefronts of the SIMD have one ready -- tools/ubench, profiles/r03_valu_rate_*.txt -- and a first form of this kernel was
// built on that: 15 % faster alone, no gain beside the extract kernels of other streams, whose wavefronts take the pairing away.
// The packed form does not depend on it: 15 % faster alone, +1.8 % frames/s in the bench's schedule.)
// So the sign of  e = A^2 + B^2 - pz^2  is first formed in f32 from f32-rounded copies of the same inputs, together with a
// rigorous bound m on |e32 - e*| (e* = the value in exact arithmetic on the double inputs):
//   * a dot product fma(r0, X, fma(r1, Y, fma(r2, Z, r3))) of rounded inputs is within 5.01 u S of the exact one,
//     S = |r0 X| + |r1 Y| + |r2 Z| + |r3|,  u = 2^-24 (two input roundings and at most three fma roundings per term);
//   * A32 = fma(-oc32, pz32, px32) is within 7.1 u (Sx + |oc| Sz) of A*;  E = 8.5 u (Sxb + (ocm + 1) Szb) bounds the error of
//     A32, B32 and pz32 alike, with Sxb = rho_x nu + tau_x >= Sx, Sy and Szb = rho_z nu + tau_z >= Sz formed once per
//     (hypothesis, lane) from per-hypothesis row norms and the lane's largest |X|+|Y|+|Z| and |oc| (all rounded up);
//   * |e32 - e*| <= 2 E s + 3 E^2 + 3.01 u s^2  with  s = |A32| + |B32| + |pz32|;  with  2 E s <= E (s^2 / sigma + sigma)  for any
//     sigma > 0 and  s^2 <= 3 T,  T = A32^2 + B32^2 + pz32^2 = e32 + 2 pz32^2 (one fma, within 6 u T),  this is
//     <= E sigma + 3 E^2 + (3 E / sigma + 9.1 u) T:  a shell that is ONE fma of T.
// m = c0 + c1 t32,  c0 = E sigma + 3.1 E^2,  c1 = 3.05 E / sigma + 27 u  (sigma ~ s on the threshold circle at the scene's depth) leaves
// 1e-6 T >= 1e-6 pz^2 beyond that, so a pair decided here (|e32| > m, and
// |pz32| >= 1.01 zg + E, the double test's guard against a vanishing depth) lies outside the double test's own 1e-6 shell with
// the same sign: the count is the one the double test, and therefore the exact rule, gives.  Everything else -- about 0.15 % of
// the pairs at the bench sizes, one (hypothesis, point slot) in ten holds such a lane -- is appended to the wavefront's own queue in
// LDS (count in a scalar register) and decided after the hypothesis loop by the double-precision test, one pair per lane (its depth
// guard then uses the point's own norm, which the guard's derivation is about; the per-lane maximum above is an upper bound of it).
// NaN or infinite inputs fail every comparison and land there as well; a threshold outside [1e-3, 1e6] or magnitudes beyond 1e15
// make the guard infinite, i.e. every pair goes to the second tier.  The depth guard itself is checked once per hypothesis on the
// lane's smallest |pz| (v_min3, half an instruction per pair) and resolved per slot only when it fires.
__device__ __noinline__ unsigned long long pnp_slot_f64(const PnpArgs& a, const double* __restrict__ Xw, const double* __restrict__ x, int i,
                                                           unsigned long long mask, const double* Rs, const double* Rt, double zg,
                                                           double fx, double fy, double cx, double cy, double inv_thr, double eps)
{
    // the double-precision division-free test of pnp_score_body<false> for ONE point slot, lanes outside `mask` ignored
    const double X = Xw[3 * i], Y = Xw[3 * i + 1], Z = Xw[3 * i + 2], ox = x[2 * i], oy = x[2 * i + 1];
    const double ocx = (ox - cx) * inv_thr, ocy = (oy - cy) * inv_thr;
    const double pxs = fma(Rs[0], X, fma(Rs[1], Y, fma(Rs[2], Z, Rs[3])));
    const double pys = fma(Rs[4], X, fma(Rs[5], Y, fma(Rs[6], Z, Rs[7])));
    const double pz = fma(Rt[8], X, fma(Rt[9], Y, fma(Rt[10], Z, Rt[11])));
    const double A = fma(-ocx, pz, pxs), Bv = fma(-ocy, pz, pys);
    const double pz2 = pz * pz;
    const double e = fma(A, A, fma(Bv, Bv, -pz2));
    const double m = eps * pz2;
    const unsigned long long un = (__builtin_amdgcn_ballot_w64(!(fabs(e) > m)) | __builtin_amdgcn_ballot_w64(fabs(pz) < zg)) & mask;
    unsigned long long in = __builtin_amdgcn_ballot_w64(e < -m) & mask & ~un;
    if (un) { // wave-uniform, rare: the exact cvProjectPoints2 evaluation (same operations as project_pt)
        const double px = fma(Rt[0], X, fma(Rt[1], Y, fma(Rt[2], Z, Rt[3])));
        const double py = fma(Rt[4], X, fma(Rt[5], Y, fma(Rt[6], Z, Rt[7])));
        const double z = pz ? 1. / pz : 1;
        const double u = fma(px * z, fx, cx), v = fma(py * z, fy, cy);
        const double dx = ox - u, dy = oy - v;
        in |= __builtin_amdgcn_ballot_w64(fma(dx, dx, dy * dy) < a.thr2cut) & un;
    }
    return in;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_bc(float x) { return (f32x2){ x, x }; }
struct alignas(16) PnpModel32 { float r[12]; float c0, c1, zg, pad; };   // rows 0/1 scaled (threshold units), row 2; the shell m = c0 + c1 t and the depth guard (per pass over the points)
struct PnpBound32 { float e_ocnu, e_oc, e_nu, e_1, s_nu, s_1, tn; };             // per-hypothesis factors of the error bound
constexpr int PNP32_PT = 8;       // points per lane of the f32 kernel
constexpr int PNP_WQCAP = 1024;   // undecided pairs a wavefront can queue per pass over its points (a few dozen are, at the bench sizes)

__device__ __forceinline__ void pnp_score_body_f32(const PnpArgs& a, int b, int h0, int nh, int tid, int lane, const double (*s_Rt)[12],
                                                   const double (*s_Rs)[8], const double* s_tn, PnpModel32* s_m32, const PnpBound32* s_b32, float* s_red, const int* s_valid, int* s_cnt, uint32_t* wq)
{
    const int P = a.P[b];
    const double* K = a.K + b * 9;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double* Xw = a.X + (size_t)b * a.capP * 3;
    const double* x = a.x + (size_t)b * a.capP * 2;
    const double inv_thr = 1.0 / a.thr;
    const bool thr_ok = a.thr >= 1e-3 && a.thr <= 1e6;
    const double eps = thr_ok ? 1e-6 : __longlong_as_double(0x7ff0000000000000ll);
    const double sf = (fabs(fx) > fabs(fy) ? fabs(fx) : fabs(fy)) * inv_thr;
    const float UP = 1.0f + 0x1p-20f;                       // a rounded-to-nearest float times UP is >= the value it was rounded from
    const float sf32 = (float)sf * UP;
    constexpr float C1 = 27.0f * 0x1p-24f;
    for (int c0 = 0; c0 < P; c0 += SCORE_THREADS * PNP32_PT) {
        f32x2 X2[PNP32_PT / 2], Y2[PNP32_PT / 2], Z2[PNP32_PT / 2], nox[PNP32_PT / 2], noy[PNP32_PT / 2];   // points 2 jj, 2 jj + 1; (oc negated)
        unsigned long long livem[PNP32_PT];
        double nrm = 0.0;
        float ocm = 0.f;
        int wn = 0;   // entries in this wavefront's queue (wave-uniform)
#pragma unroll
        for (int j = 0; j < PNP32_PT; j++) {
            const int i = c0 + j * SCORE_THREADS + tid;
            const bool live = i < P;
            livem[j] = __builtin_amdgcn_ballot_w64(live);
            const int ic = live ? i : 0;
            const double Xd = Xw[3 * ic], Yd = Xw[3 * ic + 1], Zd = Xw[3 * ic + 2];
            const double ocxd = (x[2 * ic] - cx) * inv_thr, ocyd = (x[2 * ic + 1] - cy) * inv_thr;
            // A lane without a point gets one that every hypothesis rejects with certainty (an observation 1e15 thresholds away:
            // e ~ 1e30 pz^2 is far outside the shell, never an inlier), so the hot loop needs no liveness mask; if pz vanishes the
            // slot takes the double test, which masks by liveness.
            const float xf = live ? (float)Xd : 0.f, yf = live ? (float)Yd : 0.f, zf = live ? (float)Zd : 0.f;
            const float oxf = live ? (float)ocxd : 1e15f, oyf = live ? (float)ocyd : 1e15f;
            if (j & 1) { X2[j / 2].y = xf; Y2[j / 2].y = yf; Z2[j / 2].y = zf; nox[j / 2].y = -oxf; noy[j / 2].y = -oyf; }
            else { X2[j / 2].x = xf; Y2[j / 2].x = yf; Z2[j / 2].x = zf; nox[j / 2].x = -oxf; noy[j / 2].x = -oyf; }
            if (live) {
                nrm = fmax(nrm, fabs(Xd) + fabs(Yd) + fabs(Zd));   // the double test's own quantity (its zg uses it)
                ocm = fmaxf(ocm, fmaxf(fabsf(oxf), fabsf(oyf)));
            }
        }
        // The shell m = c0 + c1 t and the depth guard are formed once per (hypothesis, pass over the points) by the first 32 lanes, from
        // the workgroup's largest |X|_1 and |oc| (a bound for every lane; per-lane values would cost a dozen instructions per
        // (hypothesis, lane) for a few per cent of tightness) and the per-hypothesis factors of the prologue
        float nu = (float)nrm * UP;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { nu = fmaxf(nu, __shfl_xor(nu, o)); ocm = fmaxf(ocm, __shfl_xor(ocm, o)); }
        __syncthreads();   // the previous pass is done with s_red and the shell constants
        if (lane == 0) { s_red[2 * (tid >> 6)] = nu; s_red[2 * (tid >> 6) + 1] = ocm; }
        __syncthreads();
        if (tid < PNP_HPB) {
            float nuw = 0.f, ocw = 0.f;
#pragma unroll
            for (int w = 0; w < SCORE_THREADS / 64; w++) { nuw = fmaxf(nuw, s_red[2 * w]); ocw = fmaxf(ocw, s_red[2 * w + 1]); }
            ocw = ocw * UP + 1.0f;
            const bool tier_ok = thr_ok && sf32 < 1e15f && nuw < 1e15f && ocw < 1e15f;   // a sane dynamic range, else every pair goes to the second tier
            const PnpBound32 Bq = s_b32[tid];
            const float E = fmaf(Bq.e_ocnu, ocw * nuw * UP, fmaf(Bq.e_oc, ocw, fmaf(Bq.e_nu, nuw, Bq.e_1)));   // CE (Sxb + ocm Szb)
            const float sigma = fmaf(Bq.s_nu, nuw, Bq.s_1);     // about s on the threshold circle at the scene's depth (any positive value is valid)
            const float rs = __builtin_amdgcn_rcpf(sigma) * UP;  // >= 1 / sigma (v_rcp_f32 is within one ulp)
            s_m32[tid].c0 = E * fmaf(3.1f, E, sigma);
            s_m32[tid].c1 = fmaf(3.05f * E, rs, C1);
            s_m32[tid].zg = tier_ok ? fmaf(1.01e-7f * sf32, nuw + Bq.tn, E) + 1e-30f : __builtin_inff();
        }
        __syncthreads();
        // the hypotheses to score as a mask in a scalar register (no LDS round trip per hypothesis to learn that it has a pose)
        unsigned vmask = (unsigned)__builtin_amdgcn_ballot_w64(lane < nh && s_valid[lane & (PNP_HPB - 1)] != 0);
        while (vmask) {
            const int q = __builtin_ctz(vmask);
            vmask &= vmask - 1;
            const float4* mp = reinterpret_cast<const float4*>(&s_m32[q]);   // 64 bytes: four 16-byte LDS reads from one address
            const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3];
            const float r[12] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w };
            const float c0m = m3.x, c1m = m3.y, zg32 = m3.z;
            int cnt = 0;
            unsigned long long um[PNP32_PT], inm[PNP32_PT], any = 0;
            float pzmin = __builtin_inff();
#pragma unroll
            for (int jj = 0; jj < PNP32_PT / 2; jj++) {
                // two points per instruction (v_pk_fma_f32, the model coefficient broadcast by op_sel): a packed instruction takes the
                // issue slot alone like a double-precision one, but carries two pairs
                const f32x2 pz = pk_fma(pk_bc(r[8]), X2[jj], pk_fma(pk_bc(r[9]), Y2[jj], pk_fma(pk_bc(r[10]), Z2[jj], pk_bc(r[11]))));
                const f32x2 px = pk_fma(pk_bc(r[0]), X2[jj], pk_fma(pk_bc(r[1]), Y2[jj], pk_fma(pk_bc(r[2]), Z2[jj], pk_bc(r[3]))));
                const f32x2 py = pk_fma(pk_bc(r[4]), X2[jj], pk_fma(pk_bc(r[5]), Y2[jj], pk_fma(pk_bc(r[6]), Z2[jj], pk_bc(r[7]))));
                const f32x2 A = pk_fma(nox[jj], pz, px), Bv = pk_fma(noy[jj], pz, py);
                const f32x2 p2 = pz * pz;
                const f32x2 e = pk_fma(A, A, pk_fma(Bv, Bv, -p2));
                const f32x2 t = pk_fma(pk_bc(2.0f), p2, e);                 // A^2 + B^2 + pz^2
                const f32x2 m = pk_fma(pk_bc(c1m), t, pk_bc(c0m));
                // scalar-unit work per pair kept to a popcount, an add and an or: the scalar unit serves all four SIMDs of the CU
                um[2 * jj] = __builtin_amdgcn_ballot_w64(!(fabsf(e.x) > m.x));
                um[2 * jj + 1] = __builtin_amdgcn_ballot_w64(!(fabsf(e.y) > m.y));
                inm[2 * jj] = __builtin_amdgcn_ballot_w64(e.x < -m.x);
                inm[2 * jj + 1] = __builtin_amdgcn_ballot_w64(e.y < -m.y);
                cnt += __popcll(inm[2 * jj]) + __popcll(inm[2 * jj + 1]);
                any |= um[2 * jj] | um[2 * jj + 1];
                pzmin = fminf(pzmin, fminf(fabsf(pz.x), fabsf(pz.y)));   // the depth guard is checked once per hypothesis (one v_min3 per two points)
            }
            {   // depth guard: rare (pz is metres in front of the camera), so it is checked once per hypothesis on the lane's smallest
                // |pz| and the slots it concerns are found again only then (a lane without a point has pz = t_z, which may be small)
                if (__builtin_amdgcn_ballot_w64(!(pzmin >= zg32))) {
#pragma unroll
                    for (int j = 0; j < PNP32_PT; j++) {
                        const float xj = (j & 1) ? X2[j / 2].y : X2[j / 2].x, yj = (j & 1) ? Y2[j / 2].y : Y2[j / 2].x, zj = (j & 1) ? Z2[j / 2].y : Z2[j / 2].x;
                        const float pz = fmaf(r[8], xj, fmaf(r[9], yj, fmaf(r[10], zj, r[11])));
                        const unsigned long long g = __builtin_amdgcn_ballot_w64(!(fabsf(pz) >= zg32)) & livem[j];
                        um[j] |= g; any |= g;
                    }
                }
            }
            if (any && a.f32_tier != 2) { // wave-uniform (2: timing probe without the second tier)
                // undecided (hypothesis, point) pairs go to this wavefront's queue (its own region of LDS and a count in a scalar
                // register: no atomics, no barrier) and are decided in double precision on dense lanes after the hypothesis loop;
                // a full queue (inputs far outside anything a camera produces) decides the slot at once
#pragma unroll
                for (int j = 0; j < PNP32_PT; j++) {
                    const unsigned long long mask = um[j] & livem[j];
                    if (!mask) continue;
                    cnt -= __popcll(inm[j] & um[j]);
                    const int n = __popcll(mask);
                    if (wn + n <= PNP_WQCAP) {
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        if (__builtin_amdgcn_inverse_ballot_w64(mask)) wq[wn + rank] = (uint32_t)q << 16 | (uint32_t)(j * SCORE_THREADS + tid);
                        wn += n;
                    } else {
                        const double zg = 1e-7 * sf * (nrm + s_tn[q]) + 1e-300;
                        const int i = c0 + j * SCORE_THREADS + tid;
                        cnt += __popcll(pnp_slot_f64(a, Xw, x, i < P ? i : 0, mask, s_Rs[q], s_Rt[q], zg, fx, fy, cx, cy, inv_thr, eps));
                    }
                }
            }
            if (lane == 0) atomicAdd(&s_cnt[q], cnt);
        }
        // ---- second tier: this wavefront's queued pairs, one per lane, by the double-precision division-free test (and the exact
        // rule inside its shell); LDS operations of one wavefront complete in order, so its own queue writes are visible to it
        for (int k0 = 0; k0 < wn; k0 += 64) {
            const int k = k0 + lane;
            const bool on = k < wn;
            const uint32_t ent = on ? wq[k] : 0u;
            const int q = (int)(ent >> 16), i = c0 + (int)(ent & 0xFFFFu);
            const double* Rs = s_Rs[q];   // per-lane hypothesis
            const double* Rt = s_Rt[q];
            const int ic = on ? i : 0;
            const double Xd = Xw[3 * ic], Yd = Xw[3 * ic + 1], Zd = Xw[3 * ic + 2], ox = x[2 * ic], oy = x[2 * ic + 1];
            const double ocxd = (ox - cx) * inv_thr, ocyd = (oy - cy) * inv_thr;
            const double zg = 1e-7 * sf * (fabs(Xd) + fabs(Yd) + fabs(Zd) + s_tn[q]) + 1e-300;   // the guard with the point's own norm
            const double pxs = fma(Rs[0], Xd, fma(Rs[1], Yd, fma(Rs[2], Zd, Rs[3])));
            const double pys = fma(Rs[4], Xd, fma(Rs[5], Yd, fma(Rs[6], Zd, Rs[7])));
            const double pz = fma(Rt[8], Xd, fma(Rt[9], Yd, fma(Rt[10], Zd, Rt[11])));
            const double A = fma(-ocxd, pz, pxs), Bv = fma(-ocyd, pz, pys);
            const double pz2 = pz * pz;
            const double e = fma(A, A, fma(Bv, Bv, -pz2));
            const double m = eps * pz2;
            bool in;
            if (!(fabs(e) > m) || fabs(pz) < zg) {   // the exact cvProjectPoints2 evaluation (same operations as project_pt)
                const double px = fma(Rt[0], Xd, fma(Rt[1], Yd, fma(Rt[2], Zd, Rt[3])));
                const double py = fma(Rt[4], Xd, fma(Rt[5], Yd, fma(Rt[6], Zd, Rt[7])));
                const double z = pz ? 1. / pz : 1;
                const double u = fma(px * z, fx, cx), v = fma(py * z, fy, cy);
                const double dx = ox - u, dy = oy - v;
                in = fma(dx, dx, dy * dy) < a.thr2cut;
            } else {
                in = e < -m;
            }
            if (on && in) atomicAdd(&s_cnt[q], 1);
        }
    }
}

__global__ __launch_bounds__(SCORE_THREADS, 3)   // three wavefronts per SIMD (168 VGPRs): two more registers would leave two
void pnp_score_k(PnpArgs a)
{
    const int b = blockIdx.y, h0 = blockIdx.x * PNP_HPB, tid = threadIdx.x, lane = tid & 63;
    const int nh = min(PNP_HPB, a.H - h0);
    if (a.f32_tier && !pnp_has_dist(a.dist + b * 8)) return;   // pnp_score32_k scores this problem
    __shared__ double s_Rt[PNP_HPB][12];
    __shared__ double s_Rs[PNP_HPB][8]; // rows 0/1 scaled by fx/thr, fy/thr (division-free test only)
    __shared__ double s_tn[PNP_HPB];
    __shared__ int s_cnt[PNP_HPB];
    __shared__ int s_valid[PNP_HPB];
    {
        const double* K = a.K + b * 9;
        const double sfx = K[0] / a.thr, sfy = K[4] / a.thr;
        for (int i = tid; i < PNP_HPB * 12; i += SCORE_THREADS) {
            const int hh = i / 12, e = i % 12;
            const double v = (hh < nh) ? a.models[((size_t)b * a.H + h0 + hh) * 12 + e] : 0.0;
            s_Rt[hh][e] = v;
            if (e < 8) s_Rs[hh][e] = v * (e < 4 ? sfx : sfy);
        }
    }
    if (tid < PNP_HPB) s_cnt[tid] = 0;
    __syncthreads();
    if (tid < PNP_HPB) { // a hypothesis without a pose is all zero (row 2 of a rotation cannot be zero)
        s_valid[tid] = !(s_Rt[tid][8] == 0.0 && s_Rt[tid][9] == 0.0 && s_Rt[tid][10] == 0.0);
        s_tn[tid] = fabs(s_Rt[tid][3]) + fabs(s_Rt[tid][7]) + fabs(s_Rt[tid][11]);
    }
    __syncthreads();
    if (pnp_has_dist(a.dist + b * 8)) pnp_score_body<true>(a, b, h0, nh, tid, lane, s_Rt, s_Rs, s_tn, s_valid, s_cnt);
    else pnp_score_body<false>(a, b, h0, nh, tid, lane, s_Rt, s_Rs, s_tn, s_valid, s_cnt);
    __syncthreads();
    if (tid < nh) a.counts[(size_t)b * a.H + h0 + tid] = s_valid[tid] ? s_cnt[tid] : 0;
}

// The no-distortion problems of the batch with the single-precision first tier (a kernel of its own: the double bodies above need
// 168 VGPRs, this one fits four wavefronts per SIMD, which the paired issue of its f32 instructions needs).
__global__ __launch_bounds__(SCORE_THREADS, 4)
void pnp_score32_k(PnpArgs a)
{
    const int b = blockIdx.y, h0 = blockIdx.x * PNP_HPB, tid = threadIdx.x, lane = tid & 63;
    const int nh = min(PNP_HPB, a.H - h0);
    if (pnp_has_dist(a.dist + b * 8)) return;   // pnp_score_k scores this problem
    __shared__ double s_Rt[PNP_HPB][12];
    __shared__ double s_Rs[PNP_HPB][8];
    __shared__ double s_tn[PNP_HPB];
    __shared__ int s_cnt[PNP_HPB];
    __shared__ int s_valid[PNP_HPB];
    __shared__ PnpModel32 s_m32[PNP_HPB];
    __shared__ PnpBound32 s_b32[PNP_HPB];
    __shared__ float s_red[2 * SCORE_THREADS / 64];
    __shared__ uint32_t s_queue[SCORE_THREADS / 64][PNP_WQCAP];
    {
        const double* K = a.K + b * 9;
        const double sfx = K[0] / a.thr, sfy = K[4] / a.thr;
        for (int i = tid; i < PNP_HPB * 12; i += SCORE_THREADS) {
            const int hh = i / 12, e = i % 12;
            const double v = (hh < nh) ? a.models[((size_t)b * a.H + h0 + hh) * 12 + e] : 0.0;
            s_Rt[hh][e] = v;
            if (e < 8) s_Rs[hh][e] = v * (e < 4 ? sfx : sfy);
        }
    }
    if (tid < PNP_HPB) s_cnt[tid] = 0;
    __syncthreads();
    if (tid < PNP_HPB) {
        s_valid[tid] = !(s_Rt[tid][8] == 0.0 && s_Rt[tid][9] == 0.0 && s_Rt[tid][10] == 0.0);
        s_tn[tid] = fabs(s_Rt[tid][3]) + fabs(s_Rt[tid][7]) + fabs(s_Rt[tid][11]);
        // single-precision copy of the model and the row norms its error bound is made of (rounded up)
        const float UP = 1.0f + 0x1p-20f;
        PnpModel32 M;
#pragma unroll
        for (int e = 0; e < 8; e++) M.r[e] = (float)s_Rs[tid][e];
#pragma unroll
        for (int e = 8; e < 12; e++) M.r[e] = (float)s_Rt[tid][e];
        const double* R = s_Rs[tid];
        const float rho_x = (float)fmax(fmax(fmax(fabs(R[0]), fabs(R[1])), fabs(R[2])), fmax(fmax(fabs(R[4]), fabs(R[5])), fabs(R[6]))) * UP;
        const float tau_x = (float)fmax(fabs(R[3]), fabs(R[7])) * UP;
        const float rho_z = (float)fmax(fmax(fabs(s_Rt[tid][8]), fabs(s_Rt[tid][9])), fabs(s_Rt[tid][10])) * UP;
        const float tau_z = (float)fabs(s_Rt[tid][11]) * UP;
        // E = CE (rho_x nu + tau_x + ocm (rho_z nu + tau_z)), sigma = 2.4 (rho_z nu / 2 + tau_z): the per-hypothesis factors (rounded up)
        constexpr float CE = 8.5f * 0x1p-24f;
        PnpBound32 Bq;
        Bq.e_ocnu = CE * rho_z * UP; Bq.e_oc = CE * tau_z * UP; Bq.e_nu = CE * rho_x * UP; Bq.e_1 = CE * tau_x * UP;
        Bq.s_nu = 1.2f * rho_z; Bq.s_1 = 2.4f * tau_z + 1e-30f;
        Bq.tn = (float)s_tn[tid] * UP;
        M.c0 = M.c1 = M.zg = M.pad = 0.f;
        s_m32[tid] = M;
        s_b32[tid] = Bq;
    }
    __syncthreads();
    pnp_score_body_f32(a, b, h0, nh, tid, lane, s_Rt, s_Rs, s_tn, s_m32, s_b32, s_red, s_valid, s_cnt, s_queue[tid >> 6]);
    __syncthreads();
    if (tid < nh) a.counts[(size_t)b * a.H + h0 + tid] = s_valid[tid] ? s_cnt[tid] : 0;
}

__device__ void R_to_rodrigues(const double* R, double* r)
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

constexpr int FIN_THREADS = 256;
__global__ __launch_bounds__(FIN_THREADS)
void pnp_finalize_k(PnpArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int P = a.P[b];
    const int best = a.best[b];
    const int bc = a.best_cnt[b];
    uint8_t* mask = a.inliers + (size_t)b * a.capP;
    double* Rt_out = a.Rt + (size_t)b * 12;
    __shared__ double s_red[27 * FIN_THREADS];
    __shared__ double s_Rt[12];
    __shared__ int s_n, s_fail, s_done;
    if (best < 0 || bc < a.min_inliers) {
        for (int i = tid; i < P; i += FIN_THREADS) mask[i] = 0;
        if (tid < 12) Rt_out[tid] = 0.0;
        if (tid < 3) { if (a.rvec) a.rvec[b * 3 + tid] = 0.0; if (a.tvec) a.tvec[b * 3 + tid] = 0.0; }
        if (tid == 0) { a.n_inliers[b] = 0; if (a.status) a.status[b] = MSLAM_ERR_DEGENERATE; }
        return;
    }
    if (tid < 12) s_Rt[tid] = a.models[((size_t)b * a.H + best) * 12 + tid];
    if (tid == 0) { s_n = 0; s_fail = 0; s_done = 0; }
    __syncthreads();
    const double* K = a.K + b * 9;
    const double* k = a.dist + b * 8;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const bool has_dist = (k[0] != 0.0) | (k[1] != 0.0) | (k[2] != 0.0) | (k[3] != 0.0) | (k[4] != 0.0) | (k[5] != 0.0) | (k[6] != 0.0) | (k[7] != 0.0);
    const double* Xw = a.X + (size_t)b * a.capP * 3;
    const double* x = a.x + (size_t)b * a.capP * 2;
    int my = 0;
    for (int i = tid; i < P; i += FIN_THREADS) {
        double u, v;
        project_pt(s_Rt, fx, fy, cx, cy, k, has_dist, Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2], u, v);
        double dx = x[2 * i] - u, dy = x[2 * i + 1] - v;
        bool in = fma(dx, dx, dy * dy) < a.thr2cut;
        mask[i] = (uint8_t)in;
        my += in;
    }
    atomicAdd(&s_n, my);
    __syncthreads();
    if (tid == 0) { a.n_inliers[b] = s_n; if (a.status) a.status[b] = MSLAM_OK; }
    // ---- Gauss-Newton on the inliers: H (21 unique) and g (6) by block reduction, 6x6 Cholesky on lane 0
    for (int it = 0; it < a.refine_iters; it++) {
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; q++) acc[q] = 0.0;
        for (int i = tid; i < P; i += FIN_THREADS) {
            if (!mask[i]) continue;
            const double Xi0 = Xw[3 * i], Xi1 = Xw[3 * i + 1], Xi2 = Xw[3 * i + 2];
            double px = s_Rt[0] * Xi0 + s_Rt[1] * Xi1 + s_Rt[2] * Xi2 + s_Rt[3];
            double py = s_Rt[4] * Xi0 + s_Rt[5] * Xi1 + s_Rt[6] * Xi2 + s_Rt[7];
            double pz = s_Rt[8] * Xi0 + s_Rt[9] * Xi1 + s_Rt[10] * Xi2 + s_Rt[11];
            if (pz == 0.0) continue;
            double iz = 1.0 / pz, xx = px * iz, yy = py * iz;
            double r2 = xx * xx + yy * yy, r4 = r2 * r2, r6 = r4 * r2;
            double cd = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
            double icd2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
            double xd = xx * cd * icd2 + k[2] * 2 * xx * yy + k[3] * (r2 + 2 * xx * xx);
            double yd = yy * cd * icd2 + k[2] * (r2 + 2 * yy * yy) + k[3] * 2 * xx * yy;
            double ru = x[2 * i] - (xd * fx + cx), rv = x[2 * i + 1] - (yd * fy + cy);
            double dcd = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4;
            double dicd = -(k[5] + 2 * k[6] * r2 + 3 * k[7] * r4) * icd2 * icd2;
            double rad = cd * icd2, drad = dcd * icd2 + cd * dicd;
            double dxx = rad + xx * drad * 2 * xx + 2 * k[2] * yy + k[3] * (2 * xx + 4 * xx);
            double dxy = xx * drad * 2 * yy + 2 * k[2] * xx + k[3] * 2 * yy;
            double dyx = yy * drad * 2 * xx + k[2] * 2 * xx + 2 * k[3] * yy;
            double dyy = rad + yy * drad * 2 * yy + k[2] * (2 * yy + 4 * yy) + 2 * k[3] * xx;
            double a00 = iz, a02 = -xx * iz, a11 = iz, a12 = -yy * iz;
            double m[6];
            m[0] = fx * (dxx * a00); m[1] = fx * (dxy * a11); m[2] = fx * (dxx * a02 + dxy * a12);
            m[3] = fy * (dyx * a00); m[4] = fy * (dyy * a11); m[5] = fy * (dyx * a02 + dyy * a12);
            double Ju[6], Jv[6];
            Ju[0] = m[1] * (-pz) + m[2] * (py); Ju[1] = m[0] * (pz) + m[2] * (-px); Ju[2] = m[0] * (-py) + m[1] * (px);
            Ju[3] = m[0]; Ju[4] = m[1]; Ju[5] = m[2];
            Jv[0] = m[4] * (-pz) + m[5] * (py); Jv[1] = m[3] * (pz) + m[5] * (-px); Jv[2] = m[3] * (-py) + m[4] * (px);
            Jv[3] = m[3]; Jv[4] = m[4]; Jv[5] = m[5];
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) acc[q++] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[21 + r] += Ju[r] * ru + Jv[r] * rv;
        }
        block_reduce<27, FIN_THREADS>(acc, s_red, tid);
        if (tid == 0) {
            double Hm[36], g[6], Lc[36], yv[6], d[6];
            int q = 0;
            for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { Hm[r * 6 + c] = acc[q]; Hm[c * 6 + r] = acc[q]; q++; }
            for (int r = 0; r < 6; r++) g[r] = acc[21 + r];
            bool ok = true;
            for (int i = 0; i < 36; i++) Lc[i] = 0.0;
            for (int i = 0; i < 6 && ok; i++)
                for (int j = 0; j <= i; j++) {
                    double sum = Hm[i * 6 + j];
                    for (int w = 0; w < j; w++) sum -= Lc[i * 6 + w] * Lc[j * 6 + w];
                    if (i == j) { if (!(sum > 0.0)) { ok = false; break; } Lc[i * 6 + i] = sqrt(sum); }
                    else Lc[i * 6 + j] = sum / Lc[j * 6 + j];
                }
            if (!ok) s_fail = 1;
            else {
                for (int i = 0; i < 6; i++) { double sum = g[i]; for (int w = 0; w < i; w++) sum -= Lc[i * 6 + w] * yv[w]; yv[i] = sum / Lc[i * 6 + i]; }
                for (int i = 5; i >= 0; i--) { double sum = yv[i]; for (int w = i + 1; w < 6; w++) sum -= Lc[w * 6 + i] * d[w]; d[i] = sum / Lc[i * 6 + i]; }
                double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], th = sqrt(th2);
                double A_, B_, C_;
                if (th < 1e-8) { A_ = 1.0; B_ = 0.5; C_ = 1.0 / 6.0; }
                else { A_ = sin(th) / th; B_ = (1 - cos(th)) / th2; C_ = (th - sin(th)) / (th2 * th); }
                const double Om[9] = { 0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0 };
                double Om2[9], dR[9], V[9], nRt[12];
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int w = 0; w < 3; w++) s += Om[i * 3 + w] * Om[w * 3 + j]; Om2[i * 3 + j] = s; }
                for (int i = 0; i < 9; i++) { double I = (i % 4 == 0) ? 1.0 : 0.0; dR[i] = I + A_ * Om[i] + B_ * Om2[i]; V[i] = I + B_ * Om[i] + C_ * Om2[i]; }
                for (int i = 0; i < 3; i++) {
                    for (int j = 0; j < 4; j++) nRt[i * 4 + j] = dR[i * 3] * s_Rt[j] + dR[i * 3 + 1] * s_Rt[4 + j] + dR[i * 3 + 2] * s_Rt[8 + j];
                    nRt[i * 4 + 3] += V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5];
                }
                for (int i = 0; i < 12; i++) s_Rt[i] = nRt[i];
                // converged: the step is below double resolution of the pose
                if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5] < 1e-20) s_done = 1;
            }
        }
        __syncthreads();
        if (s_fail || s_done) break;
    }
    if (tid == 0) {
        for (int i = 0; i < 12; i++) Rt_out[i] = s_Rt[i];
        const double R[9] = { s_Rt[0], s_Rt[1], s_Rt[2], s_Rt[4], s_Rt[5], s_Rt[6], s_Rt[8], s_Rt[9], s_Rt[10] };
        if (a.rvec) R_to_rodrigues(R, a.rvec + b * 3);
        if (a.tvec) { a.tvec[b * 3] = s_Rt[3]; a.tvec[b * 3 + 1] = s_Rt[7]; a.tvec[b * 3 + 2] = s_Rt[11]; }
    }
}

// ------------------------------------------------------------------ triangulation (R11)
struct TriArgs { const double* x1; const double* x2; int n; double P1[12], P2[12], Rt1[12], Rt2[12]; double* X; uint8_t* front; int32_t* n_front; };

// symmetric 4x4 Jacobi in registers is awkward; the 4x4 problem is done per lane in scratch memory.
// One point: DLT rows x P3 - P1, y P3 - P2 of both views, null vector of A^T A, dehomogenise, z > 0 in both cameras.
__device__ __forceinline__ bool tri_point(const double* __restrict__ P1, const double* __restrict__ P2, const double* __restrict__ Rt1,
                                          const double* __restrict__ Rt2, const double* __restrict__ x1, const double* __restrict__ x2, int p,
                                          double* __restrict__ Xout)
{
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
    jacobi_mem(AtA, 4, w, V);
    const int k0 = argmin_first(w, 4);
    const double Xh[4] = { V[0 * 4 + k0], V[1 * 4 + k0], V[2 * 4 + k0], V[3 * 4 + k0] };
    const double X0 = Xh[0] / Xh[3], X1 = Xh[1] / Xh[3], X2 = Xh[2] / Xh[3];
    Xout[3 * p] = X0; Xout[3 * p + 1] = X1; Xout[3 * p + 2] = X2;
    const double z1 = Rt1[8] * X0 + Rt1[9] * X1 + Rt1[10] * X2 + Rt1[11];
    const double z2 = Rt2[8] * X0 + Rt2[9] * X1 + Rt2[10] * X2 + Rt2[11];
    return z1 > 0 && z2 > 0;
}

__global__ __launch_bounds__(64)
void triangulate_k(TriArgs a)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    bool f = false;
    if (p < a.n) {
        f = tri_point(a.P1, a.P2, a.Rt1, a.Rt2, a.x1, a.x2, p, a.X);
        if (a.front) a.front[p] = (uint8_t)f;
    }
    unsigned long long bal = __ballot(f);
    if (threadIdx.x == 0 && bal) atomicAdd(a.n_front, __popcll(bal));
}

// G groups per launch: group g = points [ofs[g], ofs[g+1]) with its own cameras cam[g] = {P1, P2, Rt1, Rt2} (48 doubles)
__global__ __launch_bounds__(64)
void triangulate_groups_k(const double* __restrict__ x1, const double* __restrict__ x2, const int32_t* __restrict__ ofs,
                          const double* __restrict__ cam, double* __restrict__ X, uint8_t* __restrict__ front, int32_t* __restrict__ n_front)
{
    const int g = blockIdx.y;
    const int p0 = ofs[g], n = ofs[g + 1] - p0;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= n) return;
    bool f = false;
    if (i < n) {
        const double* c = cam + (size_t)g * 48;
        f = tri_point(c, c + 12, c + 24, c + 36, x1, x2, p0 + i, X);
        if (front) front[p0 + i] = (uint8_t)f;
    }
    unsigned long long bal = __ballot(f);
    if (threadIdx.x == 0 && bal) atomicAdd(n_front + g, __popcll(bal));
}

// smallest double t with sqrt(t) >= thr, so that (sqrt(d2) < thr) == (d2 < t) for every d2
double sqrt_cut(double thr)
{
    if (!(thr > 0.0)) return 0.0;
    double t = thr * thr;
    while (std::sqrt(t) >= thr) t = std::nextafter(t, 0.0);
    while (std::sqrt(t) < thr) t = std::nextafter(t, INFINITY);
    return t;
}

} // namespace

// ---- device-side preparation of batched RANSAC inputs ---------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256)
void sample_tables_k(const int32_t* n, const uint64_t* seeds, int H, int k, int32_t* tables)
{
    const int b = blockIdx.y, h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    int32_t* row = tables + ((size_t)b * H + h) * k;
    const int nb = n[b];
    if (nb < k) { for (int i = 0; i < k; i++) row[i] = 0; return; }
    uint64_t s = splitmix64(seeds[b] + (uint64_t)(h + 1) * 0x9E3779B97F4A7C15ull);
    if (!s) s = 0x9E3779B97F4A7C15ull;
    int32_t v[8];
    for (int i = 0; i < k; i++) {
        bool dup;
        int32_t c;
        do { // rejection: k << n
            s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
            c = (int32_t)((uint32_t)((s * 0x2545F4914F6CDD1Dull) >> 32) % (uint32_t)nb);
            dup = false;
#pragma unroll
            for (int j = 0; j < 8; j++) dup |= (j < i) & (v[j] == c);
        } while (dup);
#pragma unroll
        for (int j = 0; j < 8; j++) if (j == i) v[j] = c;
        row[i] = c;
    }
}

__global__ __launch_bounds__(256)
void gather_matches_k(const mslam_keypoint* kq, const mslam_keypoint* kt, int cap, const int32_t* query, const int32_t* train,
                      const int32_t* count, int min_count, double* x1, double* x2, int32_t* M_out)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int m = count[b];
    if (i == 0 && M_out) M_out[b] = m >= min_count ? m : 0;
    if (i >= m || i >= cap) return;
    const size_t o = (size_t)b * cap;
    const int q = query[o + i], t = train[o + i];
    if (q < 0 || q >= cap || t < 0 || t >= cap) return; // not a match list this library wrote
    const mslam_keypoint a = kq[o + q], c = kt[o + t];
    double2* o1 = (double2*)(x1 + (o + i) * 2);
    double2* o2 = (double2*)(x2 + (o + i) * 2);
    *o1 = make_double2((double)a.x, (double)a.y);
    *o2 = make_double2((double)c.x, (double)c.y);
}


extern "C" {

void mslam_fmat_default_params(mslam_fmat_params* p)
{
    if (!p) return;
    p->metric = 1; p->threshold = 3.0; p->confidence = 0.85; p->refit = 2; // src/ParamConfig.h:24-25
}

void mslam_pnp_default_params(mslam_pnp_params* p)
{
    if (!p) return;
    p->reproj_thr = 8.0; p->confidence = 0.0; p->refine_iters = 10; p->min_inliers = 4; p->stop_inliers = 0;
}

int mslam_fmat_ransac8_batch_dev(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* M, int capM,
                                 const int32_t* samples, int H, int B, const mslam_fmat_params* prm, double* F,
                                 uint8_t* status, int32_t* inlier_counts, int32_t* best_h, int32_t* n_inliers)
{
    if (!ctx || !x1 || !x2 || !M || !samples || !prm || !F || !status || !best_h || !n_inliers || capM < 8 || H < 1 || B < 1 ||
        (prm->metric != 0 && prm->metric != 1) || prm->refit < 0 || prm->refit > 2 || !(prm->threshold > 0))
        return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_pts = reserve(sizeof(double) * 4 * (size_t)B * capM), o_maxt = reserve(sizeof(double) * B),
                 o_mod = reserve(sizeof(double) * 9 * (size_t)B * H), o_mods = reserve(sizeof(double) * 9 * (size_t)B * H),
                 o_cnt = reserve(sizeof(int32_t) * (size_t)B * H), o_state = reserve(sizeof(int32_t) * 3 * (size_t)B);
    void* scr = nullptr;
    MSLAM_TRY(ctx_scratch(ctx, bytes, &scr));
    char* base = (char*)scr;
    FmatArgs a{};
    a.x1 = x1; a.x2 = x2; a.M = M; a.capM = capM; a.samples = samples; a.H = H; a.B = B;
    a.metric = prm->metric; a.thr = prm->metric == 0 ? prm->threshold : prm->threshold * prm->threshold;
    a.confidence = prm->confidence; a.refit = prm->refit;
    a.pts = (double*)(base + o_pts); a.maxt = (double*)(base + o_maxt); a.models = (double*)(base + o_mod);
    a.models_s = (double*)(base + o_mods);
    a.counts = inlier_counts ? inlier_counts : (int32_t*)(base + o_cnt);
    a.best = best_h; a.F = F; a.status = status; a.n_inliers = n_inliers;
    // With an adaptive bound (0 < confidence < 1, the reference's 0.85) only the hypotheses before the point where the
    // sequential RANSAC loop would stop can win, typically a few dozen of H.  Unless the caller asks for every count, the
    // hypotheses go out in growing ranges and each range's select pass carries {niters, best, maxgood} to the next: blocks
    // past the bound return at once.  The winner is the one the single full pass finds (the scan is the same scan).
    const bool ranged = !inlier_counts && prm->confidence > 0. && prm->confidence < 1. && H > 64;
    a.state = ranged ? (int32_t*)(base + o_state) : nullptr;
    a.h_lo = 0; a.h_hi = H;
    LAUNCH(ctx, K_FMAT_PREPARE, fmat_prepare_k, dim3(B), dim3(256), 0, a);
    static const int bounds[] = { 64, 256, 1024 };
    for (int r = 0, lo = 0; lo < H; r++) {
        const int hi = (ranged && r < 3 && bounds[r] < H) ? bounds[r] : H;
        a.h_lo = lo; a.h_hi = hi;
        LAUNCH(ctx, K_FMAT_SOLVE, fmat_solve_k, dim3((hi - lo + 63) / 64, B), dim3(64), 0, a);
        LAUNCH(ctx, K_FMAT_SCORE, fmat_score_k, dim3((hi - lo + FM_HPB - 1) / FM_HPB, B), dim3(SCORE_THREADS), 0, a);
        LAUNCH(ctx, K_RANSAC_SELECT, ransac_select_k, dim3(B), dim3(64), 0, a.counts, M, H, 8, prm->confidence, best_h, (int32_t*)nullptr, lo, hi, a.state, 1, 0);
        lo = hi;
    }
    LAUNCH(ctx, K_FMAT_FINALIZE, fmat_finalize_k, dim3(B), dim3(256), 0, a);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_fmat_ransac8(mslam_ctx* ctx, const double* x1, const double* x2, int M, const int32_t* samples, int H,
                       const mslam_fmat_params* prm, double F[9], uint8_t* status, int32_t* inlier_counts, int* best_h,
                       int* n_inliers)
{
    if (!ctx || !x1 || !x2 || !samples || !prm || !F || !status) return MSLAM_ERR_INVALID;
    if (M < 8 || H < 1) return MSLAM_ERR_TOO_FEW;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_x1 = reserve(sizeof(double) * 2 * M), o_x2 = reserve(sizeof(double) * 2 * M), o_M = reserve(16),
                 o_s = reserve(sizeof(int32_t) * 8 * (size_t)H), o_F = reserve(72), o_st = reserve(M),
                 o_c = reserve(sizeof(int32_t) * (size_t)H), o_b = reserve(16);
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    char* base = (char*)stg;
    hipStream_t st = ctx->stream;
    int32_t hM = M;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x1, x1, sizeof(double) * 2 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x2, x2, sizeof(double) * 2 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_M, &hM, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_s, samples, sizeof(int32_t) * 8 * (size_t)H, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int32_t* db = (int32_t*)(base + o_b);
    MSLAM_TRY(mslam_fmat_ransac8_batch_dev(ctx, (double*)(base + o_x1), (double*)(base + o_x2), (int32_t*)(base + o_M), M,
                                           (int32_t*)(base + o_s), H, 1, prm, (double*)(base + o_F), (uint8_t*)(base + o_st),
                                           (int32_t*)(base + o_c), db, db + 1));
    int32_t hb[2] = { -1, 0 };
    HIP_TRY(ctx, hipMemcpyAsync(hb, db, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(F, base + o_F, 72, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(status, base + o_st, M, hipMemcpyDeviceToHost, st));
    if (inlier_counts) HIP_TRY(ctx, hipMemcpyAsync(inlier_counts, base + o_c, sizeof(int32_t) * (size_t)H, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (best_h) *best_h = hb[0];
    if (n_inliers) *n_inliers = hb[1];
    if (hb[0] < 0) return MSLAM_ERR_DEGENERATE;
    return MSLAM_OK;
}

// findFundamentalMat(CV_FM_RANSAC) as the reference's live path calls it (src/CameraPoseEstimator.cpp:563): 7-point samples
int mslam_fmat_ransac7_batch_dev(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* M, int capM,
                                 const int32_t* samples, int H, int B, const mslam_fmat_params* prm, int method, double* F,
                                 uint8_t* status, int32_t* inlier_counts, int32_t* best_model, int32_t* n_inliers)
{
    if (!ctx || !x1 || !x2 || !M || !samples || !prm || !F || !status || !best_model || !n_inliers || capM < 7 || H < 1 || B < 1 ||
        prm->metric != 1 || (prm->refit != 0 && prm->refit != 2) || !(prm->threshold > 0) || method < 0 || method > 2 ||
        (method == 2 && capM > FM7_LMEDS_MAXM) || (size_t)H * 3 > 0x7fffffffu)
        return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    const int HM = 3 * H;
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_pts = reserve(sizeof(double) * 4 * (size_t)B * capM), o_maxt = reserve(sizeof(double) * B),
                 o_mod = reserve(sizeof(double) * 9 * (size_t)B * HM), o_cnt = reserve(sizeof(int32_t) * (size_t)B * HM),
                 o_thr = reserve(sizeof(double) * B), o_state = reserve(sizeof(int32_t) * 3 * (size_t)B);
    void* scr = nullptr;
    MSLAM_TRY(ctx_scratch(ctx, bytes, &scr));
    char* base = (char*)scr;
    FmatArgs a{};
    a.x1 = x1; a.x2 = x2; a.M = M; a.capM = capM; a.samples = samples; a.H = HM; a.B = B;
    a.metric = 1; a.thr = prm->threshold * prm->threshold; a.confidence = prm->confidence; a.refit = prm->refit;
    a.pts = (double*)(base + o_pts); a.maxt = (double*)(base + o_maxt); a.models = (double*)(base + o_mod); a.models_s = a.models;
    a.counts = inlier_counts ? inlier_counts : (int32_t*)(base + o_cnt);
    a.best = best_model; a.F = F; a.status = status; a.n_inliers = n_inliers;
    // as in the eight-point form: unless every count is asked for, the samples go out in growing ranges and blocks past the adaptive
    // bound return at once (LMedS problems keep the bound at H: their counts stay zero, their matrices are all formed)
    const bool ranged = !inlier_counts && prm->confidence > 0. && prm->confidence < 1. && H > 64;
    a.state = ranged ? (int32_t*)(base + o_state) : nullptr; a.h_lo = 0; a.h_hi = HM;
    a.mp7 = 1; a.method = method; a.H7 = H; a.thr_b = (double*)(base + o_thr);
    {   // runLMeDS's iteration count: confidence outside (0, 1) reads 0.99, outlier ratio 0.45, at least 3, at most the samples given
        double conf = prm->confidence;
        if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
        long it = lrint(log(1 - conf) / log(1 - pow(1 - 0.45, 7.0)));
        it = it > 3 ? it : 3;
        a.lmeds_iters = (int)(it < H ? it : H);
    }
    LAUNCH(ctx, K_FMAT_PREPARE, fmat_prepare_k, dim3(B), dim3(256), 0, a);
    // (counts of problems the score kernel leaves -- LMedS / direct -- are read by the select pass and perhaps by the caller: zero)
    HIP_TRY(ctx, hipMemsetAsync(a.counts, 0, sizeof(int32_t) * (size_t)B * HM, ctx->stream));
    static const int bounds[] = { 64, 256, 1024 };
    for (int r = 0, lo = 0; lo < H; r++) {
        const int hi = (ranged && r < 3 && bounds[r] < H) ? bounds[r] : H;
        a.h_lo = 3 * lo; a.h_hi = 3 * hi;
        LAUNCH(ctx, K_FMAT7_SOLVE, fmat7_solve_k, dim3((hi - lo + 63) / 64, B), dim3(64), 0, a);
        LAUNCH(ctx, K_FMAT_SCORE, fmat_score_k, dim3((3 * (hi - lo) + FM_HPB - 1) / FM_HPB, B), dim3(SCORE_THREADS), 0, a);
        LAUNCH(ctx, K_RANSAC_SELECT, ransac_select_k, dim3(B), dim3(64), 0, a.counts, M, HM, 7, prm->confidence, best_model, (int32_t*)nullptr,
               3 * lo, 3 * hi, a.state, 3, 0);
        lo = hi;
    }
    a.h_lo = 0; a.h_hi = HM;
    LAUNCH(ctx, K_FMAT7_LMEDS, fmat7_lmeds_k, dim3(B), dim3(64), 0, a);
    LAUNCH(ctx, K_FMAT_FINALIZE, fmat_finalize_k, dim3(B), dim3(256), 0, a);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_fmat_ransac7(mslam_ctx* ctx, const double* x1, const double* x2, int M, const int32_t* samples, int H,
                       const mslam_fmat_params* prm, int method, double F[9], uint8_t* status, int32_t* inlier_counts,
                       int* best_model, int* n_inliers)
{
    if (!ctx || !x1 || !x2 || !samples || !prm || !F || !status) return MSLAM_ERR_INVALID;
    if (M < 7 || H < 1) return MSLAM_ERR_TOO_FEW;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_x1 = reserve(sizeof(double) * 2 * M), o_x2 = reserve(sizeof(double) * 2 * M), o_M = reserve(16),
                 o_s = reserve(sizeof(int32_t) * 7 * (size_t)H), o_F = reserve(72), o_st = reserve(M),
                 o_c = reserve(sizeof(int32_t) * 3 * (size_t)H), o_b = reserve(16);
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    char* base = (char*)stg;
    hipStream_t st = ctx->stream;
    int32_t hM = M;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x1, x1, sizeof(double) * 2 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x2, x2, sizeof(double) * 2 * M, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_M, &hM, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_s, samples, sizeof(int32_t) * 7 * (size_t)H, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int32_t* db = (int32_t*)(base + o_b);
    MSLAM_TRY(mslam_fmat_ransac7_batch_dev(ctx, (double*)(base + o_x1), (double*)(base + o_x2), (int32_t*)(base + o_M), M,
                                           (int32_t*)(base + o_s), H, 1, prm, method, (double*)(base + o_F), (uint8_t*)(base + o_st),
                                           (int32_t*)(base + o_c), db, db + 1));
    int32_t hb[2] = { -1, 0 };
    HIP_TRY(ctx, hipMemcpyAsync(hb, db, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(F, base + o_F, 72, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(status, base + o_st, M, hipMemcpyDeviceToHost, st));
    if (inlier_counts) HIP_TRY(ctx, hipMemcpyAsync(inlier_counts, base + o_c, sizeof(int32_t) * 3 * (size_t)H, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (best_model) *best_model = hb[0];
    if (n_inliers) *n_inliers = hb[1];
    if (hb[0] < 0) return MSLAM_ERR_DEGENERATE;
    return MSLAM_OK;
}

int mslam_pnp_ransac_batch_dev(mslam_ctx* ctx, const double* X, const double* x, const int32_t* P, int capP,
                               const double* K, const double* dist, const int32_t* samples, int H, int B,
                               const mslam_pnp_params* prm, double* Rt, double* rvec, double* tvec, uint8_t* inliers,
                               int32_t* inlier_counts, int32_t* best_h, int32_t* n_inliers)
{
    if (!ctx || !X || !x || !P || !K || !dist || !samples || !prm || !Rt || !inliers || !best_h || !n_inliers || capP < 4 ||
        H < 1 || B < 1 || !(prm->reproj_thr > 0) || prm->refine_iters < 0)
        return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_xn = reserve(sizeof(double) * 2 * (size_t)B * capP), o_mod = reserve(sizeof(double) * 12 * (size_t)B * H),
                 o_cnt = reserve(sizeof(int32_t) * (size_t)B * H), o_bc = reserve(sizeof(int32_t) * B);
    void* scr = nullptr;
    MSLAM_TRY(ctx_scratch(ctx, bytes, &scr));
    char* base = (char*)scr;
    PnpArgs a{};
    a.X = X; a.x = x; a.P = P; a.capP = capP; a.K = K; a.dist = dist; a.samples = samples; a.H = H; a.B = B;
    a.thr = prm->reproj_thr; a.thr2cut = sqrt_cut(prm->reproj_thr); a.confidence = prm->confidence;
    a.refine_iters = prm->refine_iters; a.min_inliers = prm->min_inliers;
    { const char* e = getenv("MSLAM_PNP_F32"); a.f32_tier = e ? atoi(e) : 1; }   // read per call: a host can switch between calls
    a.xn = (double*)(base + o_xn); a.models = (double*)(base + o_mod);
    a.counts = inlier_counts ? inlier_counts : (int32_t*)(base + o_cnt);
    a.best = best_h; a.best_cnt = (int32_t*)(base + o_bc);
    a.Rt = Rt; a.rvec = rvec; a.tvec = tvec; a.inliers = inliers; a.n_inliers = n_inliers; a.status = nullptr;
    LAUNCH(ctx, K_PNP_PREPARE, pnp_prepare_k, dim3((capP + 255) / 256, B), dim3(256), 0, a);
    LAUNCH(ctx, K_PNP_SOLVE, pnp_solve_k, dim3((H + 63) / 64, B), dim3(64), 0, a);
    // each of the two scoring kernels leaves at once the problems the other one owns (distortion or not is device data)
    if (a.f32_tier) LAUNCH(ctx, K_PNP_SCORE32, pnp_score32_k, dim3((H + PNP_HPB - 1) / PNP_HPB, B), dim3(SCORE_THREADS), 0, a);
    LAUNCH(ctx, K_PNP_SCORE, pnp_score_k, dim3((H + PNP_HPB - 1) / PNP_HPB, B), dim3(SCORE_THREADS), 0, a);
    LAUNCH(ctx, K_RANSAC_SELECT, ransac_select_k, dim3(B), dim3(64), 0, a.counts, P, H, 4, prm->confidence, best_h, a.best_cnt, 0, H, (int32_t*)nullptr, 1, prm->stop_inliers);
    LAUNCH(ctx, K_PNP_FINALIZE, pnp_finalize_k, dim3(B), dim3(FIN_THREADS), 0, a);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_pnp_ransac(mslam_ctx* ctx, const double* X, const double* x, int P, const double K[9], const double dist[8],
                     const int32_t* samples, int H, const mslam_pnp_params* prm, double rvec[3], double tvec[3],
                     double Rt[12], uint8_t* inliers, int32_t* inlier_counts, int* best_h, int* n_inliers)
{
    if (!ctx || !X || !x || !K || !dist || !samples || !prm || !inliers) return MSLAM_ERR_INVALID;
    if (P < 4 || H < 1) return MSLAM_ERR_TOO_FEW;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_X = reserve(sizeof(double) * 3 * P), o_x = reserve(sizeof(double) * 2 * P), o_P = reserve(16), o_K = reserve(72),
                 o_d = reserve(64), o_s = reserve(sizeof(int32_t) * 4 * (size_t)H), o_Rt = reserve(96), o_r = reserve(24),
                 o_t = reserve(24), o_in = reserve(P), o_c = reserve(sizeof(int32_t) * (size_t)H), o_b = reserve(16);
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    char* base = (char*)stg;
    hipStream_t st = ctx->stream;
    int32_t hP = P;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_X, X, sizeof(double) * 3 * P, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x, x, sizeof(double) * 2 * P, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_P, &hP, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_K, K, 72, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_d, dist, 64, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_s, samples, sizeof(int32_t) * 4 * (size_t)H, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int32_t* db = (int32_t*)(base + o_b);
    MSLAM_TRY(mslam_pnp_ransac_batch_dev(ctx, (double*)(base + o_X), (double*)(base + o_x), (int32_t*)(base + o_P), P,
                                         (double*)(base + o_K), (double*)(base + o_d), (int32_t*)(base + o_s), H, 1, prm,
                                         (double*)(base + o_Rt), (double*)(base + o_r), (double*)(base + o_t),
                                         (uint8_t*)(base + o_in), (int32_t*)(base + o_c), db, db + 1));
    int32_t hb[2] = { -1, 0 };
    double hRt[12], hr[3], ht[3];
    HIP_TRY(ctx, hipMemcpyAsync(hb, db, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(hRt, base + o_Rt, 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(hr, base + o_r, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(ht, base + o_t, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(inliers, base + o_in, P, hipMemcpyDeviceToHost, st));
    if (inlier_counts) HIP_TRY(ctx, hipMemcpyAsync(inlier_counts, base + o_c, sizeof(int32_t) * (size_t)H, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (best_h) *best_h = hb[0];
    if (n_inliers) *n_inliers = hb[1];
    if (Rt) for (int i = 0; i < 12; i++) Rt[i] = hRt[i];
    if (rvec) for (int i = 0; i < 3; i++) rvec[i] = hr[i];
    if (tvec) for (int i = 0; i < 3; i++) tvec[i] = ht[i];
    if (hb[0] < 0 || hb[1] < prm->min_inliers) return MSLAM_ERR_DEGENERATE;
    return MSLAM_OK;
}

int mslam_triangulate(mslam_ctx* ctx, const double* x1, const double* x2, int n, const double Rt1[12], const double Rt2[12],
                      const double K1[9], const double K2[9], double* X, uint8_t* front, int* n_front)
{
    if (!ctx || !x1 || !x2 || !Rt1 || !Rt2 || !K1 || !K2 || !X || n < 0) return MSLAM_ERR_INVALID;
    if (n_front) *n_front = 0;
    if (n == 0) return MSLAM_OK;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t m) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += m; return o; };
    const size_t o_x1 = reserve(sizeof(double) * 2 * n), o_x2 = reserve(sizeof(double) * 2 * n), o_X = reserve(sizeof(double) * 3 * n),
                 o_f = reserve(n), o_n = reserve(16);
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    char* base = (char*)stg;
    hipStream_t st = ctx->stream;
    TriArgs a{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) { // P = K * Rt (:91)
            double p1 = 0, p2 = 0;
            for (int k = 0; k < 3; k++) { p1 += K1[i * 3 + k] * Rt1[k * 4 + j]; p2 += K2[i * 3 + k] * Rt2[k * 4 + j]; }
            a.P1[i * 4 + j] = p1; a.P2[i * 4 + j] = p2;
        }
    for (int i = 0; i < 12; i++) { a.Rt1[i] = Rt1[i]; a.Rt2[i] = Rt2[i]; }
    a.x1 = (double*)(base + o_x1); a.x2 = (double*)(base + o_x2); a.n = n; a.X = (double*)(base + o_X);
    a.front = (uint8_t*)(base + o_f); a.n_front = (int32_t*)(base + o_n);
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x1, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x2, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(base + o_n, 0, 16, st));
    LAUNCH(ctx, K_TRIANGULATE, triangulate_k, dim3((n + 63) / 64), dim3(64), 0, a);
    HIP_TRY(ctx, hipGetLastError());
    int32_t hn = 0;
    HIP_TRY(ctx, hipMemcpyAsync(X, base + o_X, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
    if (front) HIP_TRY(ctx, hipMemcpyAsync(front, base + o_f, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&hn, base + o_n, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (n_front) *n_front = hn;
    return MSLAM_OK;
}

int mslam_triangulate_batch(mslam_ctx* ctx, const double* x1, const double* x2, const int32_t* ofs, int G, const double* Rt1,
                            const double* Rt2, const double* K1, const double* K2, double* X, uint8_t* front, int32_t* n_front)
{
    if (!ctx || !x1 || !x2 || !ofs || G < 1 || !Rt1 || !Rt2 || !K1 || !K2 || !X || ofs[0] != 0) return MSLAM_ERR_INVALID;
    int max_n = 0;
    for (int g = 0; g < G; g++) { const int n = ofs[g + 1] - ofs[g]; if (n < 0) return MSLAM_ERR_INVALID; max_n = n > max_n ? n : max_n; }
    const int N = ofs[G];
    if (n_front) for (int g = 0; g < G; g++) n_front[g] = 0;
    if (N == 0) return MSLAM_OK;
    MSLAM_TRY(ctx_set_device(ctx));
    size_t bytes = 0;
    auto reserve = [&](size_t m) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += m; return o; };
    const size_t o_x1 = reserve(sizeof(double) * 2 * N), o_x2 = reserve(sizeof(double) * 2 * N), o_X = reserve(sizeof(double) * 3 * N),
                 o_f = reserve(N), o_n = reserve(sizeof(int32_t) * G), o_ofs = reserve(sizeof(int32_t) * (G + 1)), o_cam = reserve(sizeof(double) * 48 * G);
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    char* base = (char*)stg;
    hipStream_t st = ctx->stream;
    std::vector<double> cam((size_t)G * 48);
    for (int g = 0; g < G; g++) {
        double* c = &cam[(size_t)g * 48];
        const double *k1 = K1 + 9 * g, *k2 = K2 + 9 * g, *r1 = Rt1 + 12 * g, *r2 = Rt2 + 12 * g;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) { // P = K * Rt (:91), the same operation order as mslam_triangulate
                double p1 = 0, p2 = 0;
                for (int k = 0; k < 3; k++) { p1 += k1[i * 3 + k] * r1[k * 4 + j]; p2 += k2[i * 3 + k] * r2[k * 4 + j]; }
                c[i * 4 + j] = p1; c[12 + i * 4 + j] = p2;
            }
        for (int i = 0; i < 12; i++) { c[24 + i] = r1[i]; c[36 + i] = r2[i]; }
    }
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x1, x1, sizeof(double) * 2 * N, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x2, x2, sizeof(double) * 2 * N, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_ofs, ofs, sizeof(int32_t) * (G + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_cam, cam.data(), sizeof(double) * 48 * G, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(base + o_n, 0, sizeof(int32_t) * G, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));   // `cam` is a local
    LAUNCH(ctx, K_TRIANGULATE, triangulate_groups_k, dim3((max_n + 63) / 64, G), dim3(64), 0, (const double*)(base + o_x1), (const double*)(base + o_x2),
           (const int32_t*)(base + o_ofs), (const double*)(base + o_cam), (double*)(base + o_X), front ? (uint8_t*)(base + o_f) : (uint8_t*)nullptr,
           (int32_t*)(base + o_n));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(X, base + o_X, sizeof(double) * 3 * N, hipMemcpyDeviceToHost, st));
    if (front) HIP_TRY(ctx, hipMemcpyAsync(front, base + o_f, N, hipMemcpyDeviceToHost, st));
    if (n_front) HIP_TRY(ctx, hipMemcpyAsync(n_front, base + o_n, sizeof(int32_t) * G, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return MSLAM_OK;
}

int mslam_sample_tables_dev(mslam_ctx* ctx, const int32_t* n, const uint64_t* seeds, int H, int k, int B, int32_t* tables)
{
    if (!ctx || !n || !seeds || !tables || H < 1 || B < 1 || k < 1 || k > 8) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(sample_tables_k, dim3((H + 255) / 256, B), dim3(256), 0, st, n, seeds, H, k, tables);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_gather_matches_dev(mslam_ctx* ctx, const mslam_keypoint* kps_q, const mslam_keypoint* kps_t, int cap, const int32_t* query,
                             const int32_t* train, const int32_t* count, int B, int min_count, double* x1, double* x2, int32_t* M_out)
{
    if (!ctx || !kps_q || !kps_t || !query || !train || !count || !x1 || !x2 || cap < 1 || B < 1) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(gather_matches_k, dim3((cap + 255) / 256, B), dim3(256), 0, st, kps_q, kps_t, cap, query, train, count, min_count, x1, x2, M_out);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

} // extern "C"
