// pipeline_demo -- runs the re-stated ProcessingPipeline (FeatureExtractor -> CameraPoseEstimator [-> Optimiser]) over a
// sequence of raw frames, the way the reference's main() does (src/main.cpp:27-54) minus file loading and GUI stages.
// Input  (binary): int32 n, w, h, channels; double K[9]; then n*h*w*channels bytes.
// Output (text):   "frame i nfeat r00 .. r23" per frame, "mappoints M", then "id x y z nobs" per map point,
//                  "ba_chi2 v0 v1 ..." if --ba was given.
#include <cstdio>
#include <cstring>
#include <fstream>
#include "CameraPoseEstimator.h"
#include "FeatureExtractor.h"
#include "Optimiser.h"
#include "Util.h"

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.txt [--ba] [--ba-window K] [--fmat7] [--pnp-stop N] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T]\n", argv[0]); return 2; }
    bool ba = false, nodist = false, twinCheck = false;
    int fmatPts = 8, pnpStop = -1;
    int nfeatures = 500, pnpH = 100, baWindow = 0;
    double pnpThr = 8.0;
    for (int i = 3; i < argc; i++) {
        if (!std::strcmp(argv[i], "--ba")) ba = true;
        else if (!std::strcmp(argv[i], "--no-dist")) nodist = true;
        else if (!std::strcmp(argv[i], "--twin-check")) twinCheck = true;
        else if (!std::strcmp(argv[i], "--ba-window") && i + 1 < argc) { ba = true; baWindow = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--pnp-stop") && i + 1 < argc) pnpStop = std::atoi(argv[++i]);   // minInliersCount (0: off)
        else if (!std::strcmp(argv[i], "--fmat7")) fmatPts = 7;   // OpenCV's 7-point estimator (the reference's live findFundamentalMat call)
        else if (!std::strcmp(argv[i], "--nfeatures") && i + 1 < argc) nfeatures = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-hyps") && i + 1 < argc) pnpH = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-thr") && i + 1 < argc) pnpThr = std::atof(argv[++i]);
    }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    int32_t hdr[4];
    double K[9];
    in.read((char*)hdr, sizeof(hdr));
    in.read((char*)K, sizeof(K));
    const int n = hdr[0], w = hdr[1], h = hdr[2], ch = hdr[3];
    DataManager dm;
    for (int i = 0; i < n; i++) {
        Frame f;
        f.meta.frameID = i; f.meta.timestamp = i;
        f.frameBuffer.create(h, w, cv::CV_MAKETYPE(cv::CV_8U, ch));
        in.read((char*)f.frameBuffer.data, (std::streamsize)h * w * ch);
        f.K.create(3, 3, cv::CV_64F);
        for (int k = 0; k < 9; k++) f.K.at<double>(k / 3, k % 3) = K[k];
        dm.frames.push_back(f);
    }
    if (!in) { std::fprintf(stderr, "short read\n"); return 2; }
    if (twinCheck) {
        // Descriptors stay on the device as the twin of Features::descriptors.  cv::Mat allows in-place edits, so the twin must
        // not be trusted after one: match (twin path), edit frame 1's descriptors in place, match again, and compare both
        // results with the host-pointer form on clones (clone() drops the twin).
        try {
            if (n < 2) { std::fprintf(stderr, "--twin-check needs two frames\n"); return 2; }
            FeatureExtractor fe; fe.params.nfeatures = nfeatures;
            CameraPoseEstimator cpe;
            fe.process(dm, 0); fe.process(dm, 1);
            cv::Mat& d0 = dm.frames[0].features.descriptors;
            cv::Mat& d1 = dm.frames[1].features.descriptors;
            auto same = [](const std::vector<cv::DMatch>& a, const std::vector<cv::DMatch>& b) {
                if (a.size() != b.size()) return false;
                for (size_t i = 0; i < a.size(); i++) if (a[i].queryIdx != b[i].queryIdx || a[i].trainIdx != b[i].trainIdx || a[i].distance != b[i].distance) return false;
                return true;
            };
            std::vector<cv::DMatch> twin0, host0, twin1, host1;
            cpe.match(d1, d0, twin0);
            cpe.match(d1.clone(), d0.clone(), host0);
            const int rows = d0.rows < 64 ? d0.rows : 64;
            for (int r = 0; r < rows; r++) for (int c = 0; c < 32; c++) d0.at<uint8_t>(r, c) = (uint8_t)(d0.at<uint8_t>(r, c) ^ 0xFF);   // in-place edit
            cpe.match(d1, d0, twin1);
            cpe.match(d1.clone(), d0.clone(), host1);
            const bool ok = same(twin0, host0) && same(twin1, host1) && !same(twin0, twin1) && !twin0.empty();
            FILE* out = std::fopen(argv[2], "w");
            std::fprintf(out, "twin_check %s %zu %zu\n", ok ? "ok" : "FAILED", twin0.size(), twin1.size());
            std::fclose(out);
            return ok ? 0 : 1;
        } catch (const std::exception& e) { std::fprintf(stderr, "twin check failed: %s\n", e.what()); return 1; }
    }
    try {
        ProcessingPipeline slam;
        FeatureExtractor* fe = new FeatureExtractor();
        fe->params.nfeatures = nfeatures;
        CameraPoseEstimator* cpe = new CameraPoseEstimator();
        cpe->pnpHypotheses = pnpH; cpe->pnp.reproj_thr = pnpThr; cpe->fmatSamplePoints = fmatPts; if (pnpStop >= 0) cpe->pnp.stop_inliers = pnpStop;
        if (nodist) for (double& d : cpe->distCoeff) d = 0.0;
        slam.addStage(fe);
        slam.addStage(cpe);
        if (ba) { Optimiser* opt = new Optimiser(); opt->window = baWindow; slam.addStage(opt); }
        for (int i = 0; i < n; i++) slam.process(dm, i);
        FILE* out = std::fopen(argv[2], "w");
        for (int i = 0; i < n; i++) {
            const Frame& f = dm.frames[(size_t)i];
            std::fprintf(out, "frame %d %zu", i, f.features.positions.size());
            for (int k = 0; k < 12; k++) std::fprintf(out, " %.17g", f.Rt.at<double>(k / 4, k % 4));
            std::fprintf(out, "\n");
        }
        std::fprintf(out, "mappoints %zu\n", dm.mapPoints.size());
        for (const MapPoint& m : dm.mapPoints)
            std::fprintf(out, "%lu %.17g %.17g %.17g %zu\n", m.id, m.worldPosition.x, m.worldPosition.y, m.worldPosition.z, m.observers().size());
        if (ba) {
            std::fprintf(out, "ba_chi2");
            for (double v : Util::lastChi2History) std::fprintf(out, " %.17g", v);
            std::fprintf(out, "\n");
        }
        std::fclose(out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "pipeline failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
