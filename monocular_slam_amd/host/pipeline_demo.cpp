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
    if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.txt [--ba] [--ba-window K] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T]\n", argv[0]); return 2; }
    bool ba = false, nodist = false;
    int nfeatures = 500, pnpH = 100, baWindow = 0;
    double pnpThr = 8.0;
    for (int i = 3; i < argc; i++) {
        if (!std::strcmp(argv[i], "--ba")) ba = true;
        else if (!std::strcmp(argv[i], "--no-dist")) nodist = true;
        else if (!std::strcmp(argv[i], "--ba-window") && i + 1 < argc) { ba = true; baWindow = std::atoi(argv[++i]); }
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
    try {
        ProcessingPipeline slam;
        FeatureExtractor* fe = new FeatureExtractor();
        fe->params.nfeatures = nfeatures;
        CameraPoseEstimator* cpe = new CameraPoseEstimator();
        cpe->pnpHypotheses = pnpH; cpe->pnp.reproj_thr = pnpThr;
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
