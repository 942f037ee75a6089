#!/usr/bin/env python3
"""Benchmark of the tracking hot path on MI355X: frames/sec of (ORB extract + Hamming match + PnP RANSAC)
at 1920x1080, 2000 keypoints/frame (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step is one pass of the hot path over one batch of B independent streams per GPU, frames resident in
HBM: K1-K6 extract (1080p, nfeatures 2000), K7 kNN-2 + ratio test against the previous frame of the same
stream, K8b PnP RANSAC on a synthetic 2000-correspondence / 4096-hypothesis / 2 px problem per frame
(BASELINE configs[1] + configs[2]).  Streams are independent, so ranks share nothing: weak scaling, no
data-path collective; torch.distributed is used only for the barrier and the
max-over-ranks time.  The timed region holds only C-ABI calls into libmslam_hip.so (no CPU fallback
exists); the oracle is loaded only for the separately reported cpu_baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT = 1920, 1080, 2000
PNP_P, PNP_H, PNP_THR = 2000, 4096, 2.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_ISSUE_PEAK_G = 256 * 4 * 2.4 / 4  # G wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU op per 4 clocks at 2.4 GHz

# algorithmic bytes per unit: extract W*H + 56*N; match 32*(Nq+Nt) + 16*Nq;
# PnP 40*P + 4*s*H + 48 + P + 4*H
ALG_BYTES = {
    "extract": W * H + 56 * NFEAT,
    "match": 32 * (NFEAT + NFEAT) + 16 * NFEAT,
    "pnp": 40 * PNP_P + 4 * 4 * PNP_H + 48 + PNP_P + 4 * PNP_H,
}
KERNEL_UNIT = {  # which per-frame unit a kernel's launch is priced against
    "bgr2gray_k": "extract", "resize_k": "extract", "fast_nms_k": "extract", "select_k": "extract", "describe_k": "extract",
    "hamming_knn2_partial": "match", "knn2_finalize": "match",
    "pnp_prepare_k": "pnp", "pnp_solve_k": "pnp", "pnp_score_k": "pnp", "ransac_select_k": "pnp", "pnp_finalize_k": "pnp",
}


def make_frames(B: int, seed0: int) -> np.ndarray:
    from monocular_slam_amd import synth
    nbase = min(B, 4)
    base = [synth.textured_frame(W, H, seed0 + i) for i in range(nbase)]
    return np.stack([synth.shifted_frame(base[i % nbase], 3 * (i // nbase), 2 * (i // nbase), seed0 + 100 + i) for i in range(B)])


def make_pnp(B: int, seed0: int):
    from monocular_slam_amd import synth
    probs = [synth.pnp_problem(PNP_P, seed0 + b) for b in range(B)]
    base = synth.sample_table(PNP_P, PNP_H, 4, seed0)
    samples = np.stack([(base + 7 * b) % PNP_P for b in range(B)]).astype(np.int32)  # distinct indices stay distinct
    return (np.stack([q["X"] for q in probs]), np.stack([q["x"] for q in probs]), np.stack([q["K"] for q in probs]),
            np.stack([q["dist"] for q in probs]), samples, probs)


def cpu_baseline(frames: np.ndarray, pnp, n_frames: int) -> dict:
    """The oracle (CPU restatement) on this host's cores, single thread like the reference (src/Pipeline.h:57-64)."""
    frames = frames[: min(len(frames), 8)]
    from oracle import binding as ob
    ob.build()
    X, x, K, dist, samples, _ = pnp
    p = ob.orb_default_params(NFEAT)
    pp = ob.pnp_default_params()
    pp.reproj_thr = PNP_THR
    B = frames.shape[0]
    prev = None
    t0 = time.perf_counter()
    done = 0
    for i in range(n_frames + 1):
        b = i % B
        kps, desc = ob.orb_extract(frames[b], p, cap=8192)
        if prev is not None:
            rc, idx, dist_ = ob.match_knn2(desc, prev)
            ob.ratio_filter(idx, dist_, 0.8)
            ob.pnp_ransac(X[b], x[b], K[b], dist[b], samples[b], pp)
            done += 1
        else:
            t0 = time.perf_counter()  # the first frame only primes `prev`
        prev = desc
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} frames of the same workload (1920x1080 extract 2000 kpts + 2000x2000 match + PnP 2000x4096), "
                     f"oracle/liboracle.so single thread, {dt:.1f} s"}
    # all host cores, frame-level parallelism over independent streams (separate processes; the headline stays 1 core,
    # like the single-threaded reference)
    try:
        import subprocess
        ncpu = min(16, len(os.sched_getaffinity(0)))  # the one-GPU box's CPU share
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_mp.py"), str(ncpu), str(max(2, n_frames // 8))],
                           capture_output=True, text=True, timeout=300)
        out["all_cores"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # informational only
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="independent streams per GPU per step")
    ap.add_argument("--sub-batches", type=int, default=2, help="the batch is enqueued as this many sub-batches on separate HIP streams")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the second, event-instrumented pass")
    ap.add_argument("--overlap", type=int, default=1, help="1: PnP chain on a second context/stream, overlapping extract+match")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    dist = None
    # rehearsal knobs (never set by the driver): run several ranks on one GPU over gloo to exercise the N > 1 code path
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    tdev = "cuda" if backend == "nccl" else "cpu"

    from monocular_slam_amd import api
    from monocular_slam_amd.tracking import TrackingBatch

    B, NS = args.batch, max(1, args.sub_batches)
    assert B % NS == 0, "--batch must be a multiple of --sub-batches"
    SB = B // NS
    # Each sub-batch owns a context (= HIP stream) for extract+match and, with --overlap, a second one for the PnP
    # chain: kernels of different phases then share the chip (FAST/describe are latency-bound, PnP is f64-VALU-bound).
    pp = api.default_pnp_params()
    pp.reproj_thr = PNP_THR
    tbs, ctxs, frames, pnp = [], [], None, None
    for i in range(NS):
        c1 = api.Context(local_rank, W, H, 4096, SB)
        c2 = api.Context(local_rank, W, H, 4096, SB) if args.overlap else None
        tb = TrackingBatch(c1, SB, W, H, NFEAT, cap=NFEAT + 1024, ratio=0.8, pnp_points=PNP_P, pnp_hyps=PNP_H, pnp_ctx=c2)
        f_i = make_frames(SB, 5005 + 1000 * rank + 17 * i)
        tb.upload_frames(f_i)
        p_i = make_pnp(SB, 3003 + 1000 * rank + 17 * i)
        tb.set_pnp_problems(*p_i[:5], pp)
        if i == 0:
            frames, pnp = f_i, p_i
        tbs.append(tb)
        ctxs += [c for c in (c1, c2) if c is not None]

    def step():
        for tb in tbs:
            tb.step()

    def barrier():
        for tb in tbs:
            tb.sync()
        if dist is not None:
            import torch
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = True
    for tb in tbs:
        res = tb.results()
        ok = ok and bool((res["status"] == 0).all() and (res["n"] > NFEAT // 2).all() and (res["ninl"] > PNP_P // 2).all())

    # second pass (rank 0 reports): the same steps with every launch bracketed by HIP events on the
    # context's stream -> per-kernel average duration for the roofline of the dominant kernel
    kern = {}
    if not args.no_kernel_events:
        for c in ctxs:
            c.profile_enable(64 * args.steps)
        for _ in range(args.steps):
            step()
        for c in ctxs:
            for k, (ms, n) in c.profile_read().items():
                kern[k] = (kern.get(k, (0.0, 0))[0] + ms, kern.get(k, (0.0, 0))[1] + n)
            c.profile_enable(0)

    # third pass: the same launches one phase at a time (nothing else on the chip) -> exclusive kernel durations, reported
    # beside the concurrent ones; the headline value never uses them
    kern_x = {}
    if not args.no_kernel_events:
        for c in ctxs:
            c.profile_enable(64 * 3)
        for _ in range(3):
            for tb in tbs:
                tb.extract(); tb.match_prev(); tb.sync()
                tb.pnp_ransac(); tb.sync()
                tb.cur = 1 - tb.cur
        for c in ctxs:
            for k, (ms, n) in c.profile_read().items():
                kern_x[k] = (kern_x.get(k, (0.0, 0))[0] + ms, kern_x.get(k, (0.0, 0))[1] + n)
            c.profile_enable(0)

    if rank == 0:
        frames_total = world * B * args.steps
        value = frames_total / elapsed
        out = {
            "metric": "frames/sec (extract+match+PnP) at 1920x1080, 2k kpts/frame", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 (extract, match) + f64 (PnP)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]+[2]: 1920x1080 gray frames, ORB nfeatures 2000 (8 levels, 1.2), "
                                   "kNN-2 Hamming + ratio 0.8 vs previous frame, PnP RANSAC 2000 pts x 4096 hyps @ 2 px",
                       "streams_per_gpu": B, "sub_batches": NS, "frames_per_launch": SB,
                       "pnp_on_second_stream": bool(args.overlap),
                       "parallelism": f"{world} GPU(s) x {B} independent streams (no collective)"},
            "valid_outputs": ok,
        }
        if kern:
            per = {k: v[0] / max(1, v[1]) for k, v in kern.items()}          # avg ms per launch
            tot = {k: v[0] for k, v in kern.items()}
            dom = max(tot, key=tot.get)
            unit = KERNEL_UNIT.get(dom, "extract")
            launches_per_step = kern[dom][1] / args.steps
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            def roof(kname, dur_ms=None):
                u = KERNEL_UNIT.get(kname, "extract")
                lps = kern[kname][1] / args.steps
                alg_k = ALG_BYTES[u] * B / lps
                dur_ms = per[kname] if dur_ms is None else dur_ms
                ach = alg_k / (dur_ms * 1e-3) / 1e9
                tr = valu = None
                if os.path.exists(pmc_path):
                    try:  # PMC passes were collected at 32 frames per launch: only comparable per launch at that size
                        pmc = json.load(open(pmc_path))
                        if int(pmc.get("_batch", 32)) == SB and world == 1:
                            tr = pmc.get(kname, {}).get("hbm_bytes_per_launch")
                            valu = pmc.get(kname, {}).get("valu_insts_per_launch")
                    except Exception:
                        tr = valu = None
                out_r = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": tr, "algorithmic_bytes_per_launch": alg_k, "avg_launch_ms": dur_ms, "unit_priced": u}
                if valu is not None:  # what actually bounds these kernels: wave-instruction issue (1024 SIMDs, one VALU op per 4 clocks at 2.4 GHz)
                    out_r["valu_issue"] = {"achieved": valu / (dur_ms * 1e-3) / 1e9, "peak": VALU_ISSUE_PEAK_G, "unit": "G wave-instr/s",
                                           "frac": valu / (dur_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_G, "valu_insts_per_launch": valu}
                return out_r

            out["roofline"] = roof(dom)
            out["roofline"]["note"] = ("algorithmic bytes (SURVEY 8d) of the %s unit x %d frames per launch; launch duration measured with HIP "
                                       "events on the launching stream while the other %d streams run; VALU/LDS-bound, not HBM-bound "
                                       "(DESIGN.md section 4)" % (out["roofline"]["unit_priced"], SB, len(ctxs) - 1))
            out["roofline_top_kernels"] = [roof(k) for k in sorted(tot, key=tot.get, reverse=True)[:4]]
            out["kernel_ms_per_step"] = {k: v[0] / args.steps for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])}
            if kern_x:  # same kernels alone on the chip (one phase at a time, not part of the timed value)
                per_x = {k: v[0] / max(1, v[1]) for k, v in kern_x.items()}
                out["roofline_exclusive"] = [roof(k, per_x[k]) for k in sorted(tot, key=tot.get, reverse=True)[:4] if k in per_x]
                out["kernel_ms_per_launch_exclusive"] = {k: round(v, 5) for k, v in sorted(per_x.items(), key=lambda kv: -kv[1])}
        if args.cpu_frames > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, pnp, args.cpu_frames)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
