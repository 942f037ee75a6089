// mslam_bench -- the C++ twin of bench.py's timed loop: frames/sec of (ORB extract + Hamming match vs the previous frame +
// PnP RANSAC) at 1920x1080 / 2000 keypoints, B independent streams resident in HBM, driven from C++ through
// MultiStreamRunner (the batched, device-resident path a C++ host reaches; bench.py reaches the same entry points through
// ctypes).  Prints one JSON line.  Single GPU; the multi-GPU launch protocol lives in bench.py.
//   mslam_bench [--steps K] [--warmup W] [--batch B] [--sub-batches S] [--seq-len T] [--w W --h H --nfeatures N] [--device D]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include "Geometry.h"
#include "MultiStreamRunner.h"
#include "Synth.h"

int main(int argc, char** argv)
{
    int steps = 20, warmup = 3, B = 64, NS = 2, T = 16, w = 1920, h = 1080, nf = 2000, device = 0, P = 2000, H = 4096;
    double thr = 2.0;
    setenv("MSLAM_PNP_F32", "0", 0);   // like bench.py: the headline keeps the PnP inlier test in double precision unless the caller's environment says otherwise
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char* n) { return !std::strcmp(argv[i], n) && i + 1 < argc; };
        if (arg("--steps")) steps = std::atoi(argv[++i]);
        else if (arg("--warmup")) warmup = std::atoi(argv[++i]);
        else if (arg("--batch")) B = std::atoi(argv[++i]);
        else if (arg("--sub-batches")) NS = std::atoi(argv[++i]);
        else if (arg("--seq-len")) T = std::atoi(argv[++i]);
        else if (arg("--w")) w = std::atoi(argv[++i]);
        else if (arg("--h")) h = std::atoi(argv[++i]);
        else if (arg("--nfeatures")) nf = std::atoi(argv[++i]);
        else if (arg("--pnp-points")) P = std::atoi(argv[++i]);
        else if (arg("--pnp-hyps")) H = std::atoi(argv[++i]);
        else if (arg("--device")) device = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "usage: %s [--steps K] [--warmup W] [--batch B] [--sub-batches S] [--seq-len T] [--w W --h H --nfeatures N] [--pnp-points P --pnp-hyps H] [--device D]\n", argv[0]); return 2; }
    }
    if (B % NS) { std::fprintf(stderr, "--batch must be a multiple of --sub-batches\n"); return 2; }
    const int SB = B / NS;
    try {
        std::vector<std::unique_ptr<MultiStreamRunner>> runners;
        const size_t fb = (size_t)w * h;
        std::vector<uint8_t> base((size_t)4 * fb), setA((size_t)SB * fb), setB((size_t)SB * fb);
        for (int s = 0; s < NS; s++) {
            MultiStreamRunner::Config cfg;
            cfg.device = device; cfg.streams = SB; cfg.w = w; cfg.h = h; cfg.orb.nfeatures = nf; cfg.ring = 2; cfg.frameSets = 2;
            runners.emplace_back(new MultiStreamRunner(cfg));
            MultiStreamRunner& r = *runners.back();
            const int nbase = SB < 4 ? SB : 4;
            for (int i = 0; i < nbase; i++) synth::texturedFrame(w, h, 5005 + 17 * s + i, &base[(size_t)i * fb]);
            for (int i = 0; i < SB; i++) {
                synth::shiftedFrame(&base[(size_t)(i % nbase) * fb], w, h, 3 * (i / nbase), 2 * (i / nbase), 5105 + 17 * s + i, &setA[(size_t)i * fb]);
                synth::shiftedFrame(&setA[(size_t)i * fb], w, h, 5 + (i % 3), 3 - (i % 2), 5505 + 17 * s + i, &setB[(size_t)i * fb]);
            }
            r.uploadFrames(0, setA.data());
            r.uploadFrames(1, setB.data());
            std::vector<double> X((size_t)SB * P * 3), x((size_t)SB * P * 2), K((size_t)SB * 9), D((size_t)SB * 8, 0.0);
            std::vector<int32_t> Pn((size_t)SB, P), S((size_t)SB * H * 4);
            const std::vector<int32_t> table = geom::sampleTable(P, H, 4, 3003 + 17 * s);
            for (int i = 0; i < SB; i++) {
                synth::pnpProblem(P, 3003 + 17 * s + i, synth::K_MINITAUR, w, h, &X[(size_t)i * P * 3], &x[(size_t)i * P * 2]);
                std::memcpy(&K[(size_t)i * 9], synth::K_MINITAUR, sizeof(double) * 9);
                for (size_t k = 0; k < table.size(); k++) S[(size_t)i * H * 4 + k] = (table[k] + 7 * i) % P;   // distinct indices stay distinct
            }
            mslam_pnp_params prm;
            mslam_pnp_default_params(&prm);
            prm.reproj_thr = thr;
            r.setPnpProblems(P, H, X.data(), x.data(), Pn.data(), K.data(), D.data(), S.data(), prm);
        }
        long tick = 0;
        auto step = [&]() {
            for (int t = 0; t < T; t++, tick++) {
                const int cur = (int)(tick & 1), prev = 1 - cur;   // feature slot == frame set: frame k of a stream is set k % 2
                for (auto& r : runners) { r->extract(cur, cur); r->match(cur, prev, 0.8f); r->pnp(); }
            }
        };
        auto barrier = [&]() { for (auto& r : runners) r->sync(); };
        for (int i = 0; i < warmup; i++) step();
        barrier();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; i++) step();
        barrier();
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bool ok = true;
        long n_sum = 0, m_sum = 0, inl_sum = 0;
        for (auto& r : runners) {
            const int last = (int)((tick - 1) & 1);
            std::vector<int32_t> n, st, mc, mq, mt, best, ninl;
            std::vector<float> md;
            std::vector<double> Rt;
            r->downloadCounts(last, n, &st);
            r->downloadMatches(mc, mq, mt, md);
            r->downloadPnp(Rt, best, ninl);
            for (int i = 0; i < SB; i++) {
                ok = ok && st[(size_t)i] == MSLAM_OK && n[(size_t)i] > nf / 2 && mc[(size_t)i] > nf / 8 && ninl[(size_t)i] > P / 2;
                n_sum += n[(size_t)i]; m_sum += mc[(size_t)i]; inl_sum += ninl[(size_t)i];
            }
        }
        const double frames = (double)B * T * steps;
        std::printf("{\"metric\": \"frames/sec (extract+match+PnP) at %dx%d, %d kpts/frame\", \"value\": %.3f, \"unit\": \"frames/s\", \"n_gpus\": 1, "
                    "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"timed_seconds\": %.4f, \"higher_is_better\": true, \"host\": \"C++ (MultiStreamRunner over the C ABI)\", "
                    "\"config\": {\"streams_per_gpu\": %d, \"frames_per_stream_per_step\": %d, \"sub_batches\": %d, \"frames_per_launch\": %d, \"pnp\": \"%d pts x %d hyps @ %.1f px\", "
                    "\"frames_resident_in_hbm\": true, \"frame_sets_per_stream\": 2}, \"valid_outputs\": %s, "
                    "\"mean_keypoints\": %.1f, \"mean_ratio_survivors\": %.1f, \"mean_pnp_inliers\": %.1f}\n",
                    w, h, nf, frames / elapsed, steps, warmup, 1e3 * elapsed / steps, elapsed, B, T, NS, SB, P, H, thr, ok ? "true" : "false",
                    (double)n_sum / B, (double)m_sum / B, (double)inl_sum / B);
        return ok ? 0 : 3;
    } catch (const std::exception& e) { std::fprintf(stderr, "mslam_bench failed: %s\n", e.what()); return 1; }
}
