// geometry_check -- CPU-only test hook for host/Geometry.cpp (no GPU library behind it): reads commands from stdin,
// prints results with 17 significant digits.
//   jacobi n a00 a01 ...        -> w[n] V[n*n]
//   svd3 a0..a8                 -> U[9] s[3] Vt[9]
//   ldlt n S[n*n] rhs[n]        -> ok x[n]
//   rtquat Rt[12]               -> pose7 then Rt again (round trip)
//   se3exp d[6] pose7[7]        -> pose7
//   extract E[9]                -> R0[9] R1[9] t0[3] t1[3]
//   concat Rt1[12] Rt2[12]      -> Rt[12]
//   table n H k seed            -> H*k indices
//   hls m n A[m*n]              -> x[n]   (solveHLS; then dehomogenize when n == 4)
//   conv Rt[12]                 -> Converter::cvMatToSE3Quat -> SE3QuatToCvMat round trip (12 values)
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>
#include "Geometry.h"
#include "HostMath.h"

static void out(const double* v, int n) { for (int i = 0; i < n; i++) std::printf(" %.17g", v[i]); }

int main()
{
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd == "jacobi") {
            int n; std::cin >> n;
            std::vector<double> A((size_t)n * n), w((size_t)n), V((size_t)n * n);
            for (double& a : A) std::cin >> a;
            geom::jacobiEig(A.data(), n, w.data(), V.data());
            out(w.data(), n); out(V.data(), n * n);
        } else if (cmd == "svd3") {
            double A[9], U[9], s[3], Vt[9];
            for (double& a : A) std::cin >> a;
            geom::svd3(A, U, s, Vt);
            out(U, 9); out(s, 3); out(Vt, 9);
        } else if (cmd == "ldlt") {
            int n; std::cin >> n;
            std::vector<double> S((size_t)n * n), r((size_t)n), x((size_t)n);
            for (double& a : S) std::cin >> a;
            for (double& a : r) std::cin >> a;
            const bool ok = geom::ldltSolve(S.data(), n, r.data(), x.data());
            std::printf(" %d", ok ? 1 : 0); out(x.data(), n);
        } else if (cmd == "rtquat") {
            double Rt[12], p[7], back[12];
            for (double& a : Rt) std::cin >> a;
            geom::RtToSE3Quat(Rt, p); geom::SE3QuatToRt(p, back);
            out(p, 7); out(back, 12);
        } else if (cmd == "se3exp") {
            double d[6], p[7];
            for (double& a : d) std::cin >> a;
            for (double& a : p) std::cin >> a;
            geom::se3ExpLeft(d, p);
            out(p, 7);
        } else if (cmd == "extract") {
            double E[9], R[2][9], t[2][3];
            for (double& a : E) std::cin >> a;
            geom::extractRTfromE(E, R, t);
            out(R[0], 9); out(R[1], 9); out(t[0], 3); out(t[1], 3);
        } else if (cmd == "concat") {
            double A[12], B[12], C[12];
            for (double& a : A) std::cin >> a;
            for (double& a : B) std::cin >> a;
            geom::concatenateRts(A, B, C);
            out(C, 12);
        } else if (cmd == "table") {
            int n, H, k; unsigned long long seed; std::cin >> n >> H >> k >> seed;
            for (int32_t v : geom::sampleTable(n, H, k, seed)) std::printf(" %d", v);
        } else if (cmd == "hls") {
            int m, n; std::cin >> m >> n;
            cv::Mat A(m, n, cv::CV_64F), x;
            for (int i = 0; i < m * n; i++) std::cin >> A.at<double>(i / n, i % n);
            solveHLS(A, x);
            if (n == 4) dehomogenize(x);
            out(x.ptr<double>(0), n);
        } else if (cmd == "conv") {
            cv::Mat Rt(3, 4, cv::CV_64F);
            for (int i = 0; i < 12; i++) std::cin >> Rt.at<double>(i / 4, i % 4);
            cv::Mat back = Converter::SE3QuatToCvMat(Converter::cvMatToSE3Quat(Rt));
            out(back.ptr<double>(0), 12);
        } else return 2;
        std::printf("\n");
    }
    return 0;
}
