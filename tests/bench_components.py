#!/usr/bin/env python3
"""Per-component throughput on one MI355X next to the CPU oracle (single thread) -- fills the table of BASELINE.md section 2.

Configs: C1 extract (1920x1080, 2000 kpts), C1m match 2000x2000, C2 PnP RANSAC 2000 x 4096 @ 2 px,
F-RANSAC 2000 matches x 4096 hypotheses, C3 BA build 10 x 5000.  GPU timings: HIP events on the context stream around
`reps` batched launches, inputs resident in HBM.  Prints JSON and a markdown table."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
from monocular_slam_amd import api, synth  # noqa: E402
from monocular_slam_amd.tracking import TrackingBatch  # noqa: E402
from oracle import binding as ob  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = 10
W, H, NF = 1920, 1080, 2000


KERN = {}


def gpu_time(ctx, fn, reps=REPS, warm=2, tag=None):
    for _ in range(warm):
        fn()
    ctx.sync()
    if tag:  # per-kernel split from a separate event-instrumented pass
        ctx.profile_enable(64 * reps)
        for _ in range(reps):
            fn()
        KERN[tag] = {k: round(v[0] / reps, 4) for k, v in ctx.profile_read().items()}
        ctx.profile_enable(0)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps * 1e-3


def cpu_time(fn, min_s=2.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn(); n += 1
    return (time.perf_counter() - t0) / n


def main():
    ob.build()
    ctx = api.Context(0, W, H, 4096, B)
    out = {}
    D = api.DevBuf.from_array
    # ---- C1 extract
    tb = TrackingBatch(ctx, B, W, H, NF, cap=NF + 1024)
    frames = np.stack([synth.textured_frame(W, H, 5005 + i) for i in range(min(B, 4))])
    frames = np.stack([frames[i % len(frames)] for i in range(B)])
    tb.upload_frames(frames)
    t = gpu_time(ctx, tb.extract, tag='extract')
    op = ob.orb_default_params(NF)
    tc = cpu_time(lambda: ob.orb_extract(frames[0], op, cap=8192))
    out["C1 extract"] = dict(unit="frames/s", gpu=B / t, cpu=1 / tc, alg_bytes=W * H + 56 * NF)
    # ---- C1m match
    tb.step(); tb.step()
    t = gpu_time(ctx, tb.match_prev, tag='match')
    a, b, _ = synth.descriptor_sets(2000, 2003)
    tc = cpu_time(lambda: ob.match_knn2(b, a))
    out["C1m match 2000x2000"] = dict(unit="sets/s", gpu=B / t, cpu=1 / tc, alg_bytes=32 * 4000 + 16 * 2000)
    # ---- C2 PnP
    P, Hh = 2000, 4096
    probs = [synth.pnp_problem(P, 3003 + i) for i in range(B)]
    S = synth.sample_table(P, Hh, 4, 77)
    Ss = np.stack([(S + 7 * i) % P for i in range(B)]).astype(np.int32)
    pp = api.default_pnp_params(); pp.reproj_thr = 2.0
    bufs = dict(X=D(ctx, np.stack([q["X"] for q in probs])), x=D(ctx, np.stack([q["x"] for q in probs])),
                K=D(ctx, np.stack([q["K"] for q in probs])), d=D(ctx, np.stack([q["dist"] for q in probs])), S=D(ctx, Ss),
                P=D(ctx, np.full(B, P, np.int32)), Rt=api.DevBuf(ctx, B * 96), inl=api.DevBuf(ctx, B * P),
                cnt=api.DevBuf(ctx, B * Hh * 4), best=api.DevBuf(ctx, B * 4), ninl=api.DevBuf(ctx, B * 4))
    t = gpu_time(ctx, lambda: ctx.pnp_ransac_batch_dev(bufs["X"].ptr, bufs["x"].ptr, bufs["P"].ptr, P, bufs["K"].ptr, bufs["d"].ptr, bufs["S"].ptr,
                                                       Hh, B, pp, bufs["Rt"].ptr, 0, 0, bufs["inl"].ptr, bufs["cnt"].ptr, bufs["best"].ptr, bufs["ninl"].ptr), tag='pnp')
    opp = ob.pnp_default_params(); opp.reproj_thr = 2.0
    tc = cpu_time(lambda: ob.pnp_ransac(probs[0]["X"], probs[0]["x"], probs[0]["K"], probs[0]["dist"], S, opp))
    out["C2 PnP RANSAC 2000x4096"] = dict(unit="problems/s", gpu=B / t, cpu=1 / tc, alg_bytes=40 * P + 16 * Hh + 48 + P + 4 * Hh,
                                          gpu_hyps_per_s=B * Hh / t)
    # ---- F RANSAC
    M = 2000
    tvs = [synth.two_view_problem(M, 21 + i, K=synth.K_MINITAUR, w=W, h=H) for i in range(B)]
    S8 = synth.sample_table(M, Hh, 8, 6)
    S8s = np.stack([(S8 + 5 * i) % M for i in range(B)]).astype(np.int32)
    fp = api.default_fmat_params(); fp.confidence = 0.0
    fb = dict(x1=D(ctx, np.stack([q["x1"] for q in tvs])), x2=D(ctx, np.stack([q["x2"] for q in tvs])), M=D(ctx, np.full(B, M, np.int32)),
              S=D(ctx, S8s), F=api.DevBuf(ctx, B * 72), st=api.DevBuf(ctx, B * M), cnt=api.DevBuf(ctx, B * Hh * 4),
              best=api.DevBuf(ctx, B * 4), ninl=api.DevBuf(ctx, B * 4))
    t = gpu_time(ctx, lambda: ctx.fmat_ransac8_batch_dev(fb["x1"].ptr, fb["x2"].ptr, fb["M"].ptr, M, fb["S"].ptr, Hh, B, fp, fb["F"].ptr,
                                                         fb["st"].ptr, fb["cnt"].ptr, fb["best"].ptr, fb["ninl"].ptr), tag='fmat')
    ofp = ob.fmat_default_params(); ofp.confidence = 0.0
    tc = cpu_time(lambda: ob.fmat_ransac8(tvs[0]["x1"], tvs[0]["x2"], S8, ofp))
    out["F eight-point RANSAC 2000x4096"] = dict(unit="problems/s", gpu=B / t, cpu=1 / tc, alg_bytes=32 * M + 32 * Hh + 72 + M + 4 * Hh)
    # as the stages call it (src/CameraPoseEstimator.cpp:563): confidence 0.85, no per-hypothesis counts -> the library stops at the
    # adaptive iteration bound (growing hypothesis ranges), like the sequential loop it stands in for
    fpa = api.default_fmat_params()
    t = gpu_time(ctx, lambda: ctx.fmat_ransac8_batch_dev(fb["x1"].ptr, fb["x2"].ptr, fb["M"].ptr, M, fb["S"].ptr, Hh, B, fpa, fb["F"].ptr,
                                                         fb["st"].ptr, 0, fb["best"].ptr, fb["ninl"].ptr), tag='fmat_adaptive')
    ofa = ob.fmat_default_params()
    tca = cpu_time(lambda: ob.fmat_ransac8(tvs[0]["x1"], tvs[0]["x2"], S8, ofa))
    out["F eight-point RANSAC 2000x4096, confidence 0.85 without counts (GPU stops at the adaptive bound; the oracle scores every hypothesis, then selects)"] = dict(unit="problems/s", gpu=B / t, cpu=1 / tca, alg_bytes=32 * M + 72 + M)
    # ---- the 7-point estimator (findFundamentalMat(CV_FM_RANSAC) as OpenCV runs it): H samples, up to three matrices each
    S7 = synth.sample_table(M, Hh, 7, 7)
    S7s = np.stack([(S7 + 5 * i) % M for i in range(B)]).astype(np.int32)
    f7 = dict(S=D(ctx, S7s), cnt=api.DevBuf(ctx, B * Hh * 3 * 4))
    t = gpu_time(ctx, lambda: ctx.fmat_ransac7_batch_dev(fb["x1"].ptr, fb["x2"].ptr, fb["M"].ptr, M, f7["S"].ptr, Hh, B, fp, 0, fb["F"].ptr,
                                                         fb["st"].ptr, f7["cnt"].ptr, fb["best"].ptr, fb["ninl"].ptr), tag='fmat7')
    tc = cpu_time(lambda: ob.fmat_ransac7(tvs[0]["x1"], tvs[0]["x2"], S7, ofp))
    out["F seven-point RANSAC 2000x4096 (up to 3 matrices per sample)"] = dict(unit="problems/s", gpu=B / t, cpu=1 / tc, alg_bytes=32 * M + 28 * Hh + 72 + M + 12 * Hh)
    t = gpu_time(ctx, lambda: ctx.fmat_ransac7_batch_dev(fb["x1"].ptr, fb["x2"].ptr, fb["M"].ptr, M, f7["S"].ptr, Hh, B, fpa, 0, fb["F"].ptr,
                                                         fb["st"].ptr, 0, fb["best"].ptr, fb["ninl"].ptr), tag='fmat7_adaptive')
    tca = cpu_time(lambda: ob.fmat_ransac7(tvs[0]["x1"], tvs[0]["x2"], S7, ofa))
    out["F seven-point RANSAC 2000x4096, confidence 0.85 without counts (as the stages call it with fmatSamplePoints = 7)"] = dict(unit="problems/s", gpu=B / t, cpu=1 / tca, alg_bytes=32 * M + 72 + M)
    # ---- C3 BA build
    pr = synth.ba_problem(10, 5000, 4004)
    plan = api.BaPlan(ctx, pr)
    t = gpu_time(ctx, plan.build_dev, reps=50, tag='ba')
    tc = cpu_time(lambda: ob.ba_build(pr))
    E = pr["n_edges"]
    out["C3 BA build 10x5000"] = dict(unit="builds/s", gpu=1 / t, cpu=1 / tc, alg_bytes=216 * E, gpu_edges_per_s=E / t, edges=E)
    plan.close()
    # ---- C3 x B: B independent local-BA windows (one per stream) as ONE block-diagonal problem -> one build per launch
    prB = synth.ba_concat([synth.ba_problem(10, 5000, 4004 + i) for i in range(B)])
    planB = api.BaPlan(ctx, prB)
    t = gpu_time(ctx, planB.build_dev, reps=20, tag='ba_x%d' % B)
    EB = prB["n_edges"]
    out["C3 BA build, %d windows per launch" % B] = dict(unit="builds/s", gpu=B / t, cpu=1 / tc, alg_bytes=216 * EB / B, gpu_edges_per_s=EB / t, edges=EB)
    # ---- one LM iteration's device work for those B windows: build + Schur (own lambda per window) + back-substitution
    pose_w = np.repeat(np.arange(B, dtype=np.int32), 10); point_w = np.repeat(np.arange(B, dtype=np.int32), 5000)
    planB.set_windows(pose_w, point_w, B)
    lamB = np.full(B, 10.0)
    n6 = 6 * planB.view.n_free_poses
    xp = np.zeros(n6)

    def lm_iter():
        planB.build_dev(); planB.schur_windows_dev(lamB); planB.backsub_dev(xp, True)
    t = gpu_time(ctx, lm_iter, reps=10, tag='lm_x%d' % B)
    out["C3 LM iteration (build + Schur + back-subst), %d windows per launch" % B] = dict(unit="window-iterations/s", gpu=B / t, cpu=1 / cpu_time(lambda: ob.ba_schur(pr, 10.0)),
                                                                                         alg_bytes=216 * EB / B, gpu_edges_per_s=EB / t, edges=EB)
    planB.close()
    for k, v in out.items():
        v["gpu_alg_GBps"] = v["gpu"] * v["alg_bytes"] / 1e9
        v["frac_of_8TBps"] = v["gpu_alg_GBps"] / 8000.0
    print(json.dumps(out, indent=1))
    print("kernel ms per call:", json.dumps(KERN))
    print("\n| config | unit | CPU oracle, 1 thread | 1 x MI355X (batch %d) | x | algorithmic bytes/unit | achieved GB/s (alg.) | %% of 8 TB/s |" % B)
    print("|---|---|---|---|---|---|---|---|")
    for k, v in out.items():
        print("| %s | %s | %.2f | %.0f | %.0fx | %.2f MB | %.1f | %.2f %% |" % (k, v["unit"], v["cpu"], v["gpu"], v["gpu"] / v["cpu"],
                                                                               v["alg_bytes"] / 1e6, v["gpu_alg_GBps"], 100 * v["frac_of_8TBps"]))


if __name__ == "__main__":
    main()
