// sequence_demo -- src/main.cpp:27-54 of the reference: FrameLoader(directory, begin, end, step).load(dm), then
// ProcessingPipeline(FeatureExtractor -> CameraPoseEstimator [-> Optimiser]) over every frame.  Output as pipeline_demo:
// "frame i nfeat r00 .. r23" per frame (+ "gt r00 .. r23" when the data set carries ground truth), "mappoints M", ...
//   sequence_demo <directory> <begin> <end> <step> out.txt [--ba] [--ba-window K] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T]
#include <cstdio>
#include <cstring>
#include "CameraPoseEstimator.h"
#include "FeatureExtractor.h"
#include "FrameLoader.h"
#include "Optimiser.h"

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s directory begin end step out.txt [--ba] [--ba-window K] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T]\n", argv[0]); return 2; }
    bool ba = false, nodist = false;
    int nfeatures = 500, pnpH = 100, baWindow = 0;
    double pnpThr = 8.0;
    for (int i = 6; i < argc; i++) {
        if (!std::strcmp(argv[i], "--ba")) ba = true;
        else if (!std::strcmp(argv[i], "--no-dist")) nodist = true;
        else if (!std::strcmp(argv[i], "--ba-window") && i + 1 < argc) { ba = true; baWindow = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--nfeatures") && i + 1 < argc) nfeatures = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-hyps") && i + 1 < argc) pnpH = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-thr") && i + 1 < argc) pnpThr = std::atof(argv[++i]);
    }
    try {
        DataManager dm;
        FrameLoader loader(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]));
        loader.load(dm);
        if (dm.frames.empty()) { std::fprintf(stderr, "no frames loaded from %s\n", argv[1]); return 2; }
        ProcessingPipeline slam;
        FeatureExtractor* fe = new FeatureExtractor();
        fe->params.nfeatures = nfeatures;
        CameraPoseEstimator* cpe = new CameraPoseEstimator();
        cpe->pnpHypotheses = pnpH; cpe->pnp.reproj_thr = pnpThr;
        if (nodist) for (double& d : cpe->distCoeff) d = 0.0;
        slam.addStage(fe);
        slam.addStage(cpe);
        if (ba) { Optimiser* opt = new Optimiser(); opt->window = baWindow; slam.addStage(opt); }
        const int n = (int)dm.frames.size();
        for (int i = 0; i < n; i++) slam.process(dm, i);
        FILE* out = std::fopen(argv[5], "w");
        if (!out) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
        for (int i = 0; i < n; i++) {
            const Frame& f = dm.frames[(size_t)i];
            std::fprintf(out, "frame %d %zu", i, f.features.positions.size());
            for (int k = 0; k < 12; k++) std::fprintf(out, " %.17g", f.Rt.empty() ? 0.0 : f.Rt.at<double>(k / 4, k % 4));
            std::fprintf(out, "\n");
            if (!f.RtGt.empty()) {
                std::fprintf(out, "gt %d", i);
                for (int k = 0; k < 12; k++) std::fprintf(out, " %.17g", f.RtGt.at<double>(k / 4, k % 4));
                std::fprintf(out, "\n");
            }
        }
        std::fprintf(out, "mappoints %zu\n", dm.mapPoints.size());
        for (const MapPoint& mp : dm.mapPoints)
            std::fprintf(out, "%lu %.17g %.17g %.17g %zu\n", mp.id, mp.worldPosition.x, mp.worldPosition.y, mp.worldPosition.z, mp.observers().size());
        std::fclose(out);
    } catch (const std::exception& e) { std::fprintf(stderr, "sequence_demo failed: %s\n", e.what()); return 1; }
    return 0;
}
