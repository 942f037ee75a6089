// batched_demo -- BatchedPipeline over S raw-frame sequences (the pipeline_demo input format, one file per stream), next to
// what pipeline_demo gives for each of them alone.  Output: per stream "stream s", then the pipeline_demo lines
// ("frame i nfeat r00 .. r23", "mappoints M", "id x y z nobs").
//   batched_demo out.txt in0.bin in1.bin ... [--fmat7] [--pnp-stop N] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T] [--check-descriptors]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include "BatchedPipeline.h"

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s out.txt in0.bin [in1.bin ...] [--nfeatures N] [--no-dist] [--pnp-hyps H] [--pnp-thr T] [--check-descriptors] [--replicate R] [--time]\n", argv[0]); return 2; }
    std::vector<const char*> inputs;
    bool nodist = false, checkDesc = false, timeit = false;
    int fmatPts = 8, pnpStop = -1;
    int nfeatures = 500, pnpH = 100, replicate = 1;
    double pnpThr = 8.0;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--no-dist")) nodist = true;
        else if (!std::strcmp(argv[i], "--check-descriptors")) checkDesc = true;
        else if (!std::strcmp(argv[i], "--time")) timeit = true;
        else if (!std::strcmp(argv[i], "--replicate") && i + 1 < argc) replicate = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-stop") && i + 1 < argc) pnpStop = std::atoi(argv[++i]);   // minInliersCount (0: off)
        else if (!std::strcmp(argv[i], "--fmat7")) fmatPts = 7;   // OpenCV's 7-point estimator (the reference's live findFundamentalMat call)
        else if (!std::strcmp(argv[i], "--nfeatures") && i + 1 < argc) nfeatures = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-hyps") && i + 1 < argc) pnpH = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--pnp-thr") && i + 1 < argc) pnpThr = std::atof(argv[++i]);
        else inputs.push_back(argv[i]);
    }
    try {
        std::vector<std::unique_ptr<DataManager>> own;
        std::vector<DataManager*> dms;
        int n = -1;
        for (const char* path : inputs) {
            std::ifstream in(path, std::ios::binary);
            if (!in) { std::fprintf(stderr, "cannot open %s\n", path); return 2; }
            int32_t hdr[4];
            double K[9];
            in.read((char*)hdr, sizeof(hdr));
            in.read((char*)K, sizeof(K));
            if (n >= 0 && hdr[0] != n) { std::fprintf(stderr, "all streams must have the same number of frames\n"); return 2; }
            n = hdr[0];
            own.emplace_back(new DataManager());
            for (int i = 0; i < n; i++) {
                Frame f;
                f.meta.frameID = i; f.meta.timestamp = i;
                f.frameBuffer.create(hdr[2], hdr[1], cv::CV_MAKETYPE(cv::CV_8U, hdr[3]));
                in.read((char*)f.frameBuffer.data, (std::streamsize)hdr[2] * hdr[1] * hdr[3]);
                f.K.create(3, 3, cv::CV_64F);
                for (int k = 0; k < 9; k++) f.K.at<double>(k / 3, k % 3) = K[k];
                own.back()->frames.push_back(f);
            }
            if (!in) { std::fprintf(stderr, "short read on %s\n", path); return 2; }
            dms.push_back(own.back().get());
        }
        // --replicate R: every input sequence R times (independent DataManagers): more streams per launch for timing
        const size_t n_in = own.size();
        for (int r = 1; r < replicate; r++)
            for (size_t s = 0; s < n_in; s++) {
                own.emplace_back(new DataManager());
                own.back()->frames = own[s]->frames;
                for (Frame& f : own.back()->frames) { f.frameBuffer = f.frameBuffer.clone(); f.K = f.K.clone(); }
                dms.push_back(own.back().get());
            }
        FeatureExtractor fe;
        fe.params.nfeatures = nfeatures;
        CameraPoseEstimator cpe;
        cpe.pnpHypotheses = pnpH; cpe.pnp.reproj_thr = pnpThr; cpe.fmatSamplePoints = fmatPts; if (pnpStop >= 0) cpe.pnp.stop_inliers = pnpStop;
        if (nodist) for (double& d : cpe.distCoeff) d = 0.0;
        BatchedPipeline slam(dms, fe, cpe);
        std::vector<double> tick_ms;
        for (int i = 0; i < n; i++) {
            if (i == 2) for (double& v : slam.phaseMs) v = 0.0;   // phases are reported for the steady ticks (2 ..), like the rate
            const auto t0 = std::chrono::steady_clock::now();
            slam.process(i);
            tick_ms.push_back(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (timeit) {
            std::printf("{\"streams\": %zu, \"frames\": %d, \"tick_ms\": [", dms.size(), n);
            for (size_t i = 0; i < tick_ms.size(); i++) std::printf("%s%.3f", i ? ", " : "", tick_ms[i]);
            double tail = 0; int cnt = 0;
            for (size_t i = 2; i < tick_ms.size(); i++) { tail += tick_ms[i]; cnt++; }
            std::printf("], \"phase_ms_ticks_2_on\": {");
            for (int p = 0; p < BatchedPipeline::PH_COUNT; p++) std::printf("%s\"%s\": %.2f", p ? ", " : "", BatchedPipeline::phaseName(p), slam.phaseMs[p]);
            // steady state: the median tick once the feature ring is full (the first ticks also grow buffers and create streams)
            std::vector<double> steady(tick_ms.begin() + std::min<size_t>(tick_ms.size(), 6), tick_ms.end());
            std::sort(steady.begin(), steady.end());
            const double med = steady.empty() ? 0.0 : steady[steady.size() / 2];
            std::printf("}, \"tracking_frames_per_s\": %.1f, \"steady_tick_ms_median\": %.3f, \"steady_tracking_frames_per_s\": %.1f}\n",
                        cnt ? 1e3 * dms.size() * cnt / tail : 0.0, med, med > 0 ? 1e3 * dms.size() / med : 0.0);
        }
        FILE* out = std::fopen(argv[1], "w");
        if (!out) { std::fprintf(stderr, "cannot write %s\n", argv[1]); return 2; }
        for (size_t s = 0; s < dms.size(); s++) {
            std::fprintf(out, "stream %zu\n", s);
            for (int i = 0; i < n; i++) {
                const Frame& f = dms[s]->frames[(size_t)i];
                std::fprintf(out, "frame %d %zu", i, f.features.positions.size());
                for (int k = 0; k < 12; k++) std::fprintf(out, " %.17g", f.Rt.at<double>(k / 4, k % 4));
                std::fprintf(out, "\n");
            }
            std::fprintf(out, "mappoints %zu\n", dms[s]->mapPoints.size());
            for (const MapPoint& m : dms[s]->mapPoints)
                std::fprintf(out, "%lu %.17g %.17g %.17g %zu\n", m.id, m.worldPosition.x, m.worldPosition.y, m.worldPosition.z, m.observers().size());
            if (checkDesc) {   // lazily downloaded descriptors of the newest frame: a checksum line
                const cv::Mat& d = slam.descriptors((int)s, n - 1);
                unsigned long long sum = 0;
                for (int r = 0; r < d.rows; r++) for (int c = 0; c < 32; c++) sum = sum * 1099511628211ull + d.at<uint8_t>(r, c);
                std::fprintf(out, "descsum %d %llu\n", d.rows, sum);
            }
        }
        std::fclose(out);
    } catch (const std::exception& e) { std::fprintf(stderr, "batched_demo failed: %s\n", e.what()); return 1; }
    return 0;
}
