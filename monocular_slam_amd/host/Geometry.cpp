#include "Geometry.h"
#include <cmath>
#include <cstring>

namespace geom {

void jacobiEig(const double* Ain, int n, double* w, double* V)
{
    const int NM = 12;
    double a[NM][NM], v[NM][NM], d[NM], acc[NM], z[NM];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { a[i][j] = Ain[i * n + j]; v[i][j] = i == j ? 1.0 : 0.0; }
        d[i] = acc[i] = a[i][i]; z[i] = 0.0;
    }
    for (int sweep = 1; sweep <= 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p + 1 < n; p++) for (int q = p + 1; q < n; q++) off += std::fabs(a[p][q]);
        if (off == 0.0) break;
        const double thresh = sweep < 4 ? 0.2 * off / (n * n) : 0.0;
        for (int p = 0; p + 1 < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double g = 100.0 * std::fabs(a[p][q]);
                if (sweep > 4 && std::fabs(d[p]) + g == std::fabs(d[p]) && std::fabs(d[q]) + g == std::fabs(d[q])) { a[p][q] = 0.0; continue; }
                if (!(std::fabs(a[p][q]) > thresh)) continue;
                double h = d[q] - d[p], t;
                if (std::fabs(h) + g == std::fabs(h)) t = a[p][q] / h;
                else {
                    const double theta = 0.5 * h / a[p][q];
                    t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
                    if (theta < 0.0) t = -t;
                }
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
                h = t * a[p][q];
                z[p] -= h; z[q] += h; d[p] -= h; d[q] += h; a[p][q] = 0.0;
                auto rot = [&](double& x, double& y) { const double xo = x, yo = y; x = xo - s * (yo + xo * tau); y = yo + s * (xo - yo * tau); };
                for (int j = 0; j < p; j++) rot(a[j][p], a[j][q]);
                for (int j = p + 1; j < q; j++) rot(a[p][j], a[j][q]);
                for (int j = q + 1; j < n; j++) rot(a[p][j], a[q][j]);
                for (int j = 0; j < n; j++) rot(v[j][p], v[j][q]);
            }
        for (int i = 0; i < n; i++) { acc[i] += z[i]; d[i] = acc[i]; z[i] = 0.0; }
    }
    int order[NM];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 1; i < n; i++) { // stable insertion sort, ascending
        const int o = order[i];
        int j = i - 1;
        while (j >= 0 && d[order[j]] > d[o]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    for (int i = 0; i < n; i++) { w[i] = d[order[i]]; for (int j = 0; j < n; j++) V[j * n + i] = v[j][order[i]]; }
}

double det3(const double A[9])
{
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

bool inv3(const double M[9], double I[9])
{
    const double id = 1.0 / det3(M);
    I[0] = (M[4] * M[8] - M[5] * M[7]) * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = (M[5] * M[6] - M[3] * M[8]) * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = (M[3] * M[7] - M[4] * M[6]) * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    return std::isfinite(id);
}

void matmul3(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j]; C[i * 3 + j] = s; }
}

static void cross(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

void svd3(const double A[9], double U[9], double s[3], double Vt[9])
{
    double AtA[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double acc = 0; for (int k = 0; k < 3; k++) acc += A[k * 3 + i] * A[k * 3 + j]; AtA[i * 3 + j] = acc; }
    jacobiEig(AtA, 3, w, V);
    double u[3][3], vc[3][3];
    bool have[3] = { false, false, false };
    for (int c = 0; c < 3; c++) {
        const int src = 2 - c;
        s[c] = std::sqrt(w[src] > 0 ? w[src] : 0.0);
        for (int r = 0; r < 3; r++) vc[c][r] = V[r * 3 + src];
    }
    const double smax = s[0] > 0 ? s[0] : 1.0;
    for (int c = 0; c < 3; c++)
        // singular values from eig(A^T A) below sqrt(eps)*smax are noise: complete U by orthogonality instead
        if (s[c] > 1e-6 * smax) {
            for (int r = 0; r < 3; r++) u[c][r] = (A[r * 3] * vc[c][0] + A[r * 3 + 1] * vc[c][1] + A[r * 3 + 2] * vc[c][2]) / s[c];
            have[c] = true;
        }
    if (have[0] && have[1] && !have[2]) cross(u[0], u[1], u[2]);
    else if (!(have[0] && have[1] && have[2])) {
        if (!have[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
        double e[3] = { 0, 0, 0 };
        const double ax = std::fabs(u[0][0]), ay = std::fabs(u[0][1]), az = std::fabs(u[0][2]);
        if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
        const double dp = e[0] * u[0][0] + e[1] * u[0][1] + e[2] * u[0][2];
        double nn = 0;
        for (int r = 0; r < 3; r++) { u[1][r] = e[r] - dp * u[0][r]; nn += u[1][r] * u[1][r]; }
        nn = std::sqrt(nn);
        for (int r = 0; r < 3; r++) u[1][r] /= nn;
        cross(u[0], u[1], u[2]);
    }
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { U[r * 3 + c] = u[c][r]; Vt[c * 3 + r] = vc[c][r]; }
}

void essentialFromF(const double F[9], const double K1[9], const double K2[9], double E[9])
{
    double K2t[9], tmp[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K2t[i * 3 + j] = K2[j * 3 + i];
    matmul3(K2t, F, tmp);
    matmul3(tmp, K1, E);
}

void extractRTfromE(const double E[9], double R[2][9], double t[2][3])
{
    double U[9], s[3], Vt[9], UW[9];
    svd3(E, U, s, Vt);
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 };
    matmul3(U, W, UW); matmul3(UW, Vt, R[0]);   // HZ 9.19
    matmul3(U, Wt, UW); matmul3(UW, Vt, R[1]);
    // reference: if |det(R1) + 1| < EPSILON { E = -E; redo } -- an SVD of -E that keeps U is U S (-V)^T
    if (std::fabs(det3(R[0]) + 1.0) < 1e-3)
        for (int c = 0; c < 2; c++) for (int a = 0; a < 9; a++) R[c][a] = -R[c][a];
    for (int i = 0; i < 3; i++) { t[0][i] = U[i * 3 + 2]; t[1][i] = -U[i * 3 + 2]; }
}

void concatenateRts(const double A[12], const double B[12], double out[12])
{
    // The reference pads both operands to 4x4 with Mat::zeros, so the fourth row is (0 0 0 0), not (0 0 0 1)
    // (src/CameraPoseEstimator.cpp:249-252): the product is [R1 R2 | R1 t2] and t1 never enters.  It is only
    // called with Rt1 = the identity pose of frame 0 (:356), where both readings agree; restated as written.
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 4 + k] * B[k * 4 + j];
            out[i * 4 + j] = s;
        }
}

// --- SE3 (Eigen's Quaterniond(Matrix3d), quaternion product, toRotationMatrix restated)
static void quatFromR(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
static void quatNormalize(double q[4])
{
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quatToR(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void quatRotate(const double q[4], const double v[3], double o[3])
{
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

void RtToSE3Quat(const double Rt[12], double p[7])
{
    const double R[9] = { Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10] };
    quatFromR(R, p + 3); quatNormalize(p + 3);
    p[0] = Rt[3]; p[1] = Rt[7]; p[2] = Rt[11];
}

void SE3QuatToRt(const double p[7], double Rt[12])
{
    double R[9];
    quatToR(p + 3, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j]; Rt[i * 4 + 3] = p[i]; }
}

void se3ExpLeft(const double d[6], double p[7])
{
    const double om[3] = { d[0], d[1], d[2] };
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double Om[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double Om2[9], R[9], V[9];
    matmul3(Om, Om, Om2);
    if (theta < 0.00001) { // as written in se3quat.h:237-243 (no 1/2 on the quadratic term)
        for (int i = 0; i < 9; i++) { const double I = i % 4 == 0 ? 1.0 : 0.0; R[i] = I + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; i++) { const double I = i % 4 == 0 ? 1.0 : 0.0; R[i] = I + a * Om[i] + b * Om2[i]; V[i] = I + b * Om[i] + c * Om2[i]; }
    }
    double qe[4], te[3], rt[3], q[4];
    quatFromR(R, qe); quatNormalize(qe);
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5];
    quatRotate(qe, p, rt);
    const double* b4 = p + 3;
    q[3] = qe[3] * b4[3] - qe[0] * b4[0] - qe[1] * b4[1] - qe[2] * b4[2];
    q[0] = qe[3] * b4[0] + qe[0] * b4[3] + qe[1] * b4[2] - qe[2] * b4[1];
    q[1] = qe[3] * b4[1] + qe[1] * b4[3] + qe[2] * b4[0] - qe[0] * b4[2];
    q[2] = qe[3] * b4[2] + qe[2] * b4[3] + qe[0] * b4[1] - qe[1] * b4[0];
    quatNormalize(q);
    for (int i = 0; i < 3; i++) p[i] = te[i] + rt[i];
    for (int i = 0; i < 4; i++) p[3 + i] = q[i];
}

bool ldltSolve(const double* S, int n, const double* rhs, double* x)
{
    std::vector<double> L((size_t)n * n, 0.0), D((size_t)n, 0.0);
    for (int j = 0; j < n; j++) {
        double d = S[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k] * D[k];
        if (d == 0.0 || !std::isfinite(d)) return false;
        D[j] = d; L[j * n + j] = 1.0;
        for (int i = j + 1; i < n; i++) {
            double v = S[j * n + i];
            for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k] * D[k];
            L[i * n + j] = v / d;
        }
    }
    for (int i = 0; i < n; i++) { double v = rhs[i]; for (int k = 0; k < i; k++) v -= L[i * n + k] * x[k]; x[i] = v; }
    for (int i = 0; i < n; i++) x[i] /= D[i];
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k]; x[i] = v; }
    return true;
}

uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Rows are independent (row h has its own generator state), so mslam_sample_tables_dev can draw one row per thread and
// give the same table.
std::vector<int32_t> sampleTable(int n, int H, int k, uint64_t seed)
{
    std::vector<int32_t> out((size_t)H * k, 0);
    if (n < k) return out;
    for (int h = 0; h < H; h++) {
        Rng rng(splitmix64(seed + (uint64_t)(h + 1) * 0x9E3779B97F4A7C15ull));
        for (int i = 0; i < k; i++) {
            int32_t v;
            bool dup;
            do { // rejection: k << n
                v = (int32_t)rng.below((uint32_t)n);
                dup = false;
                for (int j = 0; j < i; j++) dup |= out[(size_t)h * k + j] == v;
            } while (dup);
            out[(size_t)h * k + i] = v;
        }
    }
    return out;
}

} // namespace geom
