// Geometry.h -- host-side small dense linear algebra and two-view geometry used by the stages.
// These are the pieces of the path that stay on the CPU in the product, exactly as in the reference
// (E decomposition and candidate vote: src/CameraPoseEstimator.cpp:154-194, :316-356; pose concatenation :248-255;
// Rt <-> SE3Quat: src/Converter.cpp:8-31; SE3 exp: ThirdParty/g2o/g2o/types/se3quat.h:223-257; the reduced pose
// system solve: ThirdParty/g2o/g2o/core/block_solver.hpp:354-486).  OpenCV/Eigen are absent: cv::SVD is replaced
// by a cyclic Jacobi eigen-solver (every consumer is sign-invariant).
#pragma once
#include <cstdint>
#include <vector>

namespace geom {

// eigenvalues ascending in w, eigenvectors in the columns of V (row-major n x n), n <= 12
void jacobiEig(const double* A, int n, double* w, double* V);
// A = U diag(s) Vt, s descending
void svd3(const double A[9], double U[9], double s[3], double Vt[9]);
double det3(const double A[9]);
bool inv3(const double M[9], double I[9]);
void matmul3(const double A[9], const double B[9], double C[9]);

// E = K2^T F K1 (computeEssentialMatrix)
void essentialFromF(const double F[9], const double K1[9], const double K2[9], double E[9]);
// ExtractRTfromE + the det(R1) == -1 sign fix: R[0] = U W Vt, R[1] = U W^T Vt, t[0] = u3, t[1] = -u3
void extractRTfromE(const double E[9], double R[2][9], double t[2][3]);
// concatenateRts: (Rt1 as 4x4) * (Rt2 as 4x4), top 3x4
void concatenateRts(const double Rt1[12], const double Rt2[12], double out[12]);

// pose7 = tx,ty,tz,qx,qy,qz,qw
void RtToSE3Quat(const double Rt[12], double pose7[7]);
void SE3QuatToRt(const double pose7[7], double Rt[12]);
void se3ExpLeft(const double delta6[6], double pose7[7]); // T <- exp(delta) * T, delta = (omega, upsilon)

// dense LDL^T solve (upper triangle authoritative); false if a pivot is zero / not finite
bool ldltSolve(const double* S, int n, const double* rhs, double* x);

// xorshift64* -- the stated PRNG behind every sample table the stages generate
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) % n); }
};
uint64_t splitmix64(uint64_t z);
// H rows of k distinct indices in [0, n): row h is drawn by Rng(splitmix64(seed + (h + 1) * 0x9E3779B97F4A7C15)) with repeats
// within the row redrawn (zero table if n < k).  mslam_sample_tables_dev (include/mslam.h) generates the same table on the device.
std::vector<int32_t> sampleTable(int n, int H, int k, uint64_t seed);

} // namespace geom
