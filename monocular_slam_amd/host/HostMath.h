// HostMath.h -- the small conversion / algebra helpers the reference's stages call around the hot path, with the
// reference's names, on this repo's Mat (OpenCV / Eigen / g2o types are absent):
//   namespace Converter  (src/Converter.h, src/Converter.cpp:8-31): cvMatToSE3Quat / SE3QuatToCvMat with a plain SE3Quat value
//   TakeSVD, solveHLS, dehomogenize, Mat2Point3d  (src/CommonMath.h, src/CommonMath.cpp:8-35)
// Everything forwards to host/Geometry (Jacobi-based; singular vectors are defined up to sign, as with cv::SVD).
#pragma once
#include <vector>
#include "Geometry.h"
#include "Mat.h"

struct SE3Quat {            // g2o::SE3Quat reduced to its state: translation + unit quaternion (x, y, z, w), w >= 0
    double t[3] = { 0, 0, 0 };
    double q[4] = { 0, 0, 0, 1 };
};

namespace Converter {
inline SE3Quat cvMatToSE3Quat(const cv::Mat& mat)   // mat: 3x4 or 4x4, CV_64F
{
    double Rt[12], p[7];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) Rt[i * 4 + j] = mat.at<double>(i, j);
    geom::RtToSE3Quat(Rt, p);
    SE3Quat s;
    for (int i = 0; i < 3; i++) s.t[i] = p[i];
    for (int i = 0; i < 4; i++) s.q[i] = p[3 + i];
    return s;
}
inline cv::Mat SE3QuatToCvMat(const SE3Quat& s)     // 3x4 pose matrix
{
    const double p[7] = { s.t[0], s.t[1], s.t[2], s.q[0], s.q[1], s.q[2], s.q[3] };
    double Rt[12];
    geom::SE3QuatToRt(p, Rt);
    cv::Mat m(3, 4, cv::CV_64F);
    for (int i = 0; i < 12; i++) m.at<double>(i / 4, i % 4) = Rt[i];
    return m;
}
inline cv::Point3d vector3dToPoint3d(const double v[3]) { return cv::Point3d(v[0], v[1], v[2]); }
} // namespace Converter

// E: 3x3 CV_64F -> u (3x3), vt (3x3), w (3x1, descending)
inline void TakeSVD(const cv::Mat& E, cv::Mat& svd_u, cv::Mat& svd_vt, cv::Mat& svd_w)
{
    double A[9], U[9], s[3], Vt[9];
    for (int i = 0; i < 9; i++) A[i] = E.at<double>(i / 3, i % 3);
    geom::svd3(A, U, s, Vt);
    svd_u.create(3, 3, cv::CV_64F); svd_vt.create(3, 3, cv::CV_64F); svd_w.create(3, 1, cv::CV_64F);
    for (int i = 0; i < 9; i++) { svd_u.at<double>(i / 3, i % 3) = U[i]; svd_vt.at<double>(i / 3, i % 3) = Vt[i]; }
    for (int i = 0; i < 3; i++) svd_w.at<double>(i, 0) = s[i];
}

// homogeneous least squares: the right singular vector of A (m x n, n <= 12) for the smallest singular value, as a 1 x n row
inline void solveHLS(const cv::Mat& A, cv::Mat& x)
{
    const int n = A.cols;
    std::vector<double> AtA((size_t)n * n, 0.0), w((size_t)n), V((size_t)n * n);
    for (int r = 0; r < A.rows; r++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) AtA[(size_t)i * n + j] += A.at<double>(r, i) * A.at<double>(r, j);
    geom::jacobiEig(AtA.data(), n, w.data(), V.data());   // ascending: column 0 belongs to the smallest eigenvalue
    x.create(1, n, cv::CV_64F);
    for (int i = 0; i < n; i++) x.at<double>(0, i) = V[(size_t)i * n];
}

inline void dehomogenize(cv::Mat& X)
{
    const double w = X.at<double>(3);
    for (int i = 0; i < X.rows * X.cols; i++) X.at<double>(i) /= w;
}

inline void Mat2Point3d(const cv::Mat& mat, cv::Point3d& p) { p.x = mat.at<double>(0); p.y = mat.at<double>(1); p.z = mat.at<double>(2); }
