#!/usr/bin/env python3
"""Benchmark of the tracking hot path on MI355X: frames/sec of (ORB extract + Hamming match + PnP RANSAC)
at 1920x1080, 2000 keypoints/frame (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step is one pass of the hot path over one batch of synthetic input: B independent streams per GPU, each advanced
by T consecutive frames (B x T frames per step, frames resident in HBM): K1-K6 extract (1080p, nfeatures 2000), K7
kNN-2 + ratio test against the previous frame of the same stream (the two resident frame sets of a stream alternate,
so current != previous), K8b PnP RANSAC on a synthetic 2000-correspondence / 4096-hypothesis / 2 px problem per frame
(BASELINE configs[1] + configs[2]).  Streams are independent, so ranks share no frame data: weak scaling.  At N > 1
every frame-step additionally all-gathers the int32[4096] inlier counts of each rank's stream 0 over RCCL
(mslam_comm_allgather_i32_dev on the PnP stream -- BASELINE configs[4]); torch.distributed (gloo, CPU) is the control
plane only: rendezvous of the RCCL id, the barrier and the max-over-ranks time.  The timed region holds only C-ABI
calls into libmslam_hip.so (no CPU fallback exists); the oracle is loaded afterwards, as the checker of the timed
configuration's outputs and for the separately reported cpu_baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

METRIC = "frames/sec (extract+match+PnP) at 1920x1080, 2k kpts/frame"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The headline is measured in the reference's arithmetic widths: the PnP inlier test stays in double precision (the library's
# single-precision first tier -- identical counts, scoring alone 26 % faster -- changes nothing under this concurrent schedule:
# 50.7 k frames/s either way, profiles/r03_*; MSLAM_PNP_F32=1 in the environment measures it)
os.environ.setdefault("MSLAM_PNP_F32", "0")

W, H, NFEAT = 1920, 1080, 2000
PNP_P, PNP_H, PNP_THR = 2000, 4096, 2.0
RATIO = 0.8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# G wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU op per 4 clocks at 2.4 GHz.  That is the rate of the instruction forms
# this path is made of (SDWA byte operands, packed 16-bit, compares, dot products, 3-operand integer forms, f64); plain 32-bit
# add/sub/logic/shift-right and f32 add/mul/fma issue at twice that rate when two wavefronts of a SIMD have one ready each
# (measured per opcode: tools/ubench/valu_rate, profiles/r03_valu_rate_*.txt)
ISSUE_PEAK_G = 256 * 4 * 2.4 / 4
SALU_PEAK_G = 256 * 2.4           # one scalar instruction per CU per clock (an upper bound; shared by the CU's 4 SIMDs)

# algorithmic bytes per unit: extract W*H + 56*N; match 32*(Nq+Nt) + 16*Nq;
# PnP 40*P + 4*s*H + 48 + P + 4*H
ALG_BYTES = {
    "extract": W * H + 56 * NFEAT,
    "match": 32 * (NFEAT + NFEAT) + 16 * NFEAT,
    "pnp": 40 * PNP_P + 4 * 4 * PNP_H + 48 + PNP_P + 4 * PNP_H,
}
KERNEL_UNIT = {  # which per-frame unit a kernel's launch is priced against
    "bgr2gray_k": "extract", "resize_k": "extract", "pyramid2_k": "extract", "fast_nms_k": "extract", "select_k": "extract",
    "describe_k": "extract", "hamming_knn2_partial": "match", "expand_pm1_k": "match", "hamming_mfma_k": "match", "knn2_finalize": "match",
    "pnp_prepare_k": "pnp", "pnp_solve_k": "pnp", "pnp_score_k": "pnp", "pnp_score32_k": "pnp", "ransac_select_k": "pnp", "pnp_finalize_k": "pnp",
}
# f64 flops per (hypothesis, point) pair of pnp_score_k's sign test (3 rows x (3 fma) + e = A^2 + B^2 - z^2: DESIGN.md 4)
PNP_SCORE_FLOP_PER_PAIR = 2 * 9 + 2 * 3 + 2


def make_frames(B: int, seed0: int) -> np.ndarray:
    from monocular_slam_amd import synth
    nbase = min(B, 4)
    base = [synth.textured_frame(W, H, seed0 + i) for i in range(nbase)]
    return np.stack([synth.shifted_frame(base[i % nbase], 3 * (i // nbase), 2 * (i // nbase), seed0 + 100 + i) for i in range(B)])


def make_next_frames(frames: np.ndarray, seed0: int) -> np.ndarray:
    """The next frame of every stream: the same scene moved by a few pixels with fresh sensor noise."""
    from monocular_slam_amd import synth
    return np.stack([synth.shifted_frame(frames[i], 5 + (i % 3), 3 - (i % 2), seed0 + 500 + i) for i in range(frames.shape[0])])


def make_pnp(B: int, seed0: int):
    from monocular_slam_amd import synth
    probs = [synth.pnp_problem(PNP_P, seed0 + b) for b in range(B)]
    base = synth.sample_table(PNP_P, PNP_H, 4, seed0)
    samples = np.stack([(base + 7 * b) % PNP_P for b in range(B)]).astype(np.int32)  # distinct indices stay distinct
    return (np.stack([q["X"] for q in probs]), np.stack([q["x"] for q in probs]), np.stack([q["K"] for q in probs]),
            np.stack([q["dist"] for q in probs]), samples, probs)


def cpu_baseline(frames_a: np.ndarray, frames_b: np.ndarray, pnp, n_frames: int) -> dict:
    """The oracle (CPU restatement) on this host's cores, single thread like the reference (src/Pipeline.h:57-64)."""
    from oracle import binding as ob
    ob.build()
    X, x, K, dist, samples, _ = pnp
    p = ob.orb_default_params(NFEAT)
    pp = ob.pnp_default_params()
    pp.reproj_thr = PNP_THR
    B = min(frames_a.shape[0], 8)
    prev = None
    t0 = time.perf_counter()
    done = 0
    for i in range(n_frames + 1):
        b = (i // 2) % B
        f = (frames_a, frames_b)[i % 2][b]
        kps, desc = ob.orb_extract(f, p, cap=8192)
        if prev is not None:
            rc, idx, dist_ = ob.match_knn2(desc, prev)
            ob.ratio_filter(idx, dist_, RATIO)
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


def pct(v, q):
    return float(np.percentile(np.asarray(v, np.float64), q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="independent streams per GPU")
    ap.add_argument("--seq-len", type=int, default=48, help="consecutive frames per stream per step (a step = batch x seq-len frames; 48: the driver's --steps 20 times more than a second)")
    ap.add_argument("--no-ba", action="store_true", help="skip the BA build leg (roofline_ba: BASELINE configs[3], the path's HBM-bound kernel)")
    ap.add_argument("--sub-batches", type=int, default=2, help="the batch is enqueued as this many sub-batches on separate HIP streams")
    ap.add_argument("--cpu-frames", type=int, default=64, help="frames of the cpu_baseline sample (0 = skip; 64 frames = about 12 s of one host core)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the event-instrumented passes (roofline block)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the separately timed H2D-inclusive leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed configuration's outputs")
    ap.add_argument("--overlap", type=int, default=1, help="1: PnP chain on a second context/stream, overlapping extract+match")
    ap.add_argument("--force-collective", action="store_true", help="rehearsal: create the RCCL communicator and all-gather at N = 1 too")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: gloo ("[Gloo] Rank 0 is connected to ...") and RCCL (a version banner when
    # the first communicator comes up) write to file descriptor 1 from native code, so the descriptor points at stderr until the
    # line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    # rehearsal knob (never set by the driver): several ranks on one GPU to exercise the N > 1 control path
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])

    # The HIP library first: it binds the system HIP runtime (and, lazily, the system RCCL) exactly as in the N = 1 run.
    # torch is imported afterwards and only for torch.distributed over gloo (CPU): its bundled HIP runtime is never
    # initialised, so one runtime owns the device in every configuration.
    from monocular_slam_amd import api
    from monocular_slam_amd.tracking import TrackingBatch
    api.lib()
    rccl_path = None
    if world > 1 or args.force_collective:
        try:
            rccl_path = api.comm_preload()   # the RCCL beside this library's HIP runtime, before torch brings its own copies
        except api.MslamError as e:
            rccl_path = "unavailable: %s" % e
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")

    B, NS, T = args.batch, max(1, args.sub_batches), max(1, args.seq_len)
    assert B % NS == 0, "--batch must be a multiple of --sub-batches"
    SB = B // NS
    # Each sub-batch owns a context (= HIP stream) for extract+match and, with --overlap, a second one for the PnP
    # chain: kernels of different phases then share the chip (FAST/describe are latency-bound, PnP is f64-VALU-bound).
    pp = api.default_pnp_params()
    pp.reproj_thr = PNP_THR
    tbs, ctxs, fr_a, fr_b, pnps = [], [], [], [], []
    for i in range(NS):
        c1 = api.Context(local_rank, W, H, 4096, SB)
        c2 = api.Context(local_rank, W, H, 4096, SB) if args.overlap else None
        tb = TrackingBatch(c1, SB, W, H, NFEAT, cap=NFEAT + 1024, ratio=RATIO, pnp_points=PNP_P, pnp_hyps=PNP_H, pnp_ctx=c2)
        f_a = make_frames(SB, 5005 + 1000 * rank + 17 * i)
        f_b = make_next_frames(f_a, 5005 + 1000 * rank + 17 * i)
        tb.upload_frames(f_a, f_b)
        p_i = make_pnp(SB, 3003 + 1000 * rank + 17 * i)
        tb.set_pnp_problems(*p_i[:5], pp)
        fr_a.append(f_a); fr_b.append(f_b); pnps.append(p_i)
        tbs.append(tb)
        ctxs += [c for c in (c1, c2) if c is not None]

    # ---- the collective of BASELINE configs[4]: RCCL all-gather of int32[H] inlier counts, one per frame-step
    collective = None
    if world > 1 or args.force_collective:
        collective = {"what": "all-gather of int32[%d] PnP inlier counts per rank per frame-step" % PNP_H, "ranks": world,
                      "bytes": 4 * PNP_H * world, "per_step": T}
        # Every rank takes the same decision: the id is broadcast unconditionally, the communicator is created and tried once
        # (a 4-int all-gather) under a watchdog, and the ranks agree over gloo on whether ALL of them succeeded -- a rank that
        # cannot initialise RCCL must not leave the others blocked inside a collective.
        import threading
        uid, uid_err = bytes(128), None
        if rank == 0:
            try:
                uid = api.Comm.unique_id()
            except api.MslamError as e:
                uid_err = str(e)
        if world > 1:
            import torch
            t = torch.tensor(list(uid), dtype=torch.uint8)
            dist.broadcast(t, src=0)
            uid = bytes(t.tolist())
        box = {}

        def make_comm():
            try:
                c = api.Comm(tbs[0].pnp_ctx, uid, rank, world)
                probe = c.allgather_i32(np.arange(4, dtype=np.int32) + 10 * rank)
                assert (probe.reshape(world, 4) == np.arange(4)[None, :] + 10 * np.arange(world)[:, None]).all(), "all-gather probe mismatch"
                box["comm"] = c
            except Exception as e:   # MslamError (RCCL refused / unavailable) or a failed probe
                box["err"] = str(e)[:300]
        if any(uid):
            th = threading.Thread(target=make_comm, daemon=True)
            th.start()
            th.join(float(os.environ.get("BENCH_COMM_TIMEOUT", "120")))
            if th.is_alive():
                # the creation thread may still finish later and would then share tbs[0].pnp_ctx's stream with the timed loop,
                # and the peers may still be inside ncclCommInitRank: nothing measured after this point could be trusted.
                # Report and leave (no in-process restart); the launcher ends the other ranks.
                emit({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "rank": rank,
                      "error": "RCCL communicator creation did not return within the watchdog time (BENCH_COMM_TIMEOUT)"})
                os._exit(3)
        else:
            box["err"] = uid_err or "no RCCL unique id"
        all_ok = "comm" in box
        if world > 1:
            import torch
            t = torch.tensor([1 if all_ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            all_ok = bool(t.item())
        if all_ok:
            tbs[0].set_comm(box["comm"])
            collective.update(backend="rccl (mslam_comm_allgather_i32_dev, on the PnP stream)", library=rccl_path, **box["comm"].info())
        else:   # e.g. several rehearsal ranks on one device: RCCL refuses duplicate GPUs
            collective.update(backend="gloo (host, once per step) -- RCCL communicator unavailable on at least one rank", library=rccl_path,
                              error=box.get("err", "another rank failed"))

    def gloo_gather_fallback():
        if collective is not None and tbs[0].comm is None and world > 1:
            import torch
            c = tbs[0].pnp["counts"].download(np.int32, (SB, PNP_H))[0]   # syncs the PnP stream
            out = [torch.empty(PNP_H, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(c.copy()))

    # step-time distribution: a timing stream of its own waits (on the device) for every stream that carries work of a step --
    # both sub-batches' extract+match streams and their PnP streams -- and records one event per step; no host sync inside the
    # loop and no dependency between the working streams
    ev_ctx = api.Context(local_rank, W, H, 64, 1)
    step_streams = []
    for tb in tbs:
        step_streams.append(tb.ctx)
        if tb.pnp_ctx is not tb.ctx:
            step_streams.append(tb.pnp_ctx)
    fence_evs = [[c.event() for c in step_streams] for _ in range(2)]

    def step_fence(k, ev):
        for c, e in zip(step_streams, fence_evs[k & 1]):
            c.record(e)
            ev_ctx.wait_event(e)
        ev_ctx.record(ev)

    def step():
        for _ in range(T):
            for tb in tbs:
                tb.step()
        gloo_gather_fallback()

    def barrier():
        for tb in tbs:
            tb.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    # per-step HIP events on the first sub-batch's stream (no host sync inside the loop): step-time distribution
    evs = [ev_ctx.event() for _ in range(args.steps + 1)]
    step_fence(0, evs[0])
    t0 = time.perf_counter()
    # At most IN_FLIGHT steps are enqueued ahead of the device.  The four streams of a rank are independent (the PnP chain does
    # not consume the match), so an unbounded host lets the PnP streams finish all K steps while the extract+match streams
    # are at step K/2: the wall time is the same (measured: 26.0 ms per step for depths 1, 2, 4, 8 and unbounded), but the
    # per-step times then say 30 ms for the first half and 22.6 ms for the second.  Bounded, every step is a whole step.
    depth = max(1, int(os.environ.get("BENCH_IN_FLIGHT", "2")))
    for k in range(args.steps):
        if k >= depth:
            ev_ctx.elapsed_ms(evs[0], evs[k - depth + 1])   # host waits for the fence of step k - depth
        step()
        step_fence(k + 1, evs[k + 1])
    barrier()
    ev_ctx.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [ev_ctx.elapsed_ms(evs[k], evs[k + 1]) for k in range(args.steps)]
    if os.environ.get("BENCH_DUMP_STEPS") and rank == 0:
        print("step_ms:", " ".join("%.2f" % v for v in step_ms), file=sys.stderr)

    # ---- outputs of the timed configuration: status / counts, then parity against the oracle (outside the timed region)
    ok = True
    results = [tb.results() for tb in tbs]
    for res in results:
        ok = ok and bool((res["status"] == 0).all() and (res["n"] > NFEAT // 2).all() and (res["ninl"] > PNP_P // 2).all()
                         and (res["m_count"] > NFEAT // 8).all())
    coll_ok = None
    if collective is not None and tbs[0].comm is not None:
        g = results[0]["gathered"]
        coll_ok = bool((g[rank] == results[0]["counts"][0]).all())
        if dist is not None:   # every rank must hold the same gathered vector, hence the same global winner
            import torch
            best = int(np.argmax(g.reshape(-1)))
            tt = torch.tensor([best, -best], dtype=torch.int64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            coll_ok = coll_ok and int(tt[0]) == best and int(-tt[1]) == best
        collective["verified"] = coll_ok
        ok = ok and coll_ok
    parity = None
    if not args.no_parity and rank == 0:
        from oracle import binding as ob
        from oracle import parity as opar
        ob.build()
        opp = ob.pnp_default_params()
        opp.reproj_thr = PNP_THR
        bad, checked = [], 0
        for i, tb in enumerate(tbs):
            cur_set, prev_set = tb.last_frame_sets()
            sets = (fr_a[i], fr_b[i])
            streams = sorted({0, SB - 1})
            bad += [f"sub-batch {i}: {m}" for m in opar.check_against_oracle(results[i], streams, sets[cur_set], sets[prev_set], NFEAT,
                                                                               RATIO, pnps[i], opp)]
            checked += len(streams)
        parity = {"parity_checked_streams": checked, "mismatches": bad[:8],
                  "compared": "keypoints (x, y, size, angle, response, octave), descriptors, kNN-2 idx/dist, ratio survivors: bit-exact; "
                              "PnP all 4096 counts, winner, inlier mask: identical; Rt <= 1e-9"}
        ok = ok and not bad
    del results

    # ---- second pass: the same steps with every launch bracketed by HIP events on the launching stream (concurrent
    # schedule, as timed); third pass: one phase at a time (exclusive durations).  Never part of `value`.
    kern, kern_x = {}, {}
    NPROF = 8  # frame-steps per instrumented pass
    if not args.no_kernel_events and rank == 0:
        # only rank 0 runs these passes: the collective must stay out of them (the other ranks are not there to answer it)
        saved_comm = tbs[0].comm
        tbs[0].comm = None

        def read(into):
            for c in ctxs:
                for k, (ms, n) in c.profile_read().items():
                    into[k] = (into.get(k, (0.0, 0))[0] + ms, into.get(k, (0.0, 0))[1] + n)
                c.profile_enable(0)
        for c in ctxs:
            c.profile_enable(64 * NPROF)
        for _ in range(NPROF):
            for tb in tbs:
                tb.step()
        read(kern)
        for c in ctxs:
            c.profile_enable(64 * NPROF)
        for _ in range(NPROF):
            for tb in tbs:
                tb.extract(); tb.sync()
                tb.match_prev(); tb.sync()
                tb.pnp_ransac(); tb.sync()
                tb.advance()
        read(kern_x)
        tbs[0].comm = saved_comm

    # ---- separately timed leg: the same steps with the library's default PnP scoring (single-precision first tier of the inlier
    # test, identical counts).  The headline above keeps the test in double precision; this is what a caller gets by default.
    f32_leg = None
    if world == 1 and os.environ.get("MSLAM_PNP_F32") == "0" and not os.environ.get("BENCH_NO_F32_LEG"):
        os.environ["MSLAM_PNP_F32"] = "1"   # read by the library at every PnP call
        n_f32 = max(2, min(args.steps, 8))
        step(); barrier()
        tq = time.perf_counter()
        for _ in range(n_f32):
            step()
        barrier()
        dq = time.perf_counter() - tq
        res = [tb.results() for tb in tbs]
        same = all(bool((r["status"] == 0).all() and (r["ninl"] > PNP_P // 2).all()) for r in res)
        f32_leg = {"value": B * T * n_f32 / dq, "unit": "frames/s", "steps": n_f32, "valid_outputs": same,
                   "note": "MSLAM_PNP_F32=1 (library default): pnp_score32_k decides the PnP inlier test in packed single precision under a rigorous "
                           "error bound and hands the undecided pairs to the double-precision test; counts identical (tests/test_ransac_gpu.py)"}
        del res
        os.environ["MSLAM_PNP_F32"] = "0"

    # ---- separately timed leg: the same steps with every frame set uploaded from pinned host memory on a copy stream
    # (double-buffered against the previous step's kernels).  PCIe-inclusive; reported beside `value`, never as it.
    ingest = None
    if not args.no_ingest:
        copy_ctxs = [api.Context(local_rank, W, H, 64, 1) for _ in tbs]
        for tb, cc, a, b in zip(tbs, copy_ctxs, fr_a, fr_b):
            tb.enable_ingest(cc, a, b)
        n_ing = max(2, min(args.steps, 10))
        step(); barrier()
        t1 = time.perf_counter()
        for _ in range(n_ing):
            step()
        barrier()
        dt = time.perf_counter() - t1
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ingest = {"value": world * B * T * n_ing / dt, "unit": "frames/s", "steps": n_ing,
                  "h2d_GBps": world * B * T * n_ing * W * H / dt / 1e9,
                  "note": "every frame copied host->device inside the timed loop (pinned buffers, copy stream, double-buffered); "
                          "PCIe Gen5 x16 spec 63 GB/s = 30.4 k frames/s of 2.07 MB"}
        for tb in tbs:
            tb.ingest = None

    if rank == 0:
        frames_total = world * B * T * args.steps
        value = frames_total / elapsed
        out = {
            "metric": METRIC, "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 (extract, match) + f64 (PnP)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]+[2]: 1920x1080 gray frames, ORB nfeatures 2000 (8 levels, 1.2), "
                                   "kNN-2 Hamming + ratio 0.8 vs previous frame, PnP RANSAC 2000 pts x 4096 hyps @ 2 px",
                       "streams_per_gpu": B, "frames_per_stream_per_step": T, "frames_per_step": B * T * world,
                       "sub_batches": NS, "frames_per_launch": SB,
                       "frames_resident_in_hbm": True, "frame_sets_per_stream": 2,
                       "pnp_decoupled_from_match": True, "pnp_f32_tier": os.environ.get("MSLAM_PNP_F32") != "0", "pnp_on_second_stream": bool(args.overlap),
                       "parallelism": f"{world} GPU(s) x {B} independent streams"
                                      + (" + RCCL all-gather of PnP inlier counts" if collective else " (no collective)")},
            "timed_seconds": elapsed,
            "step_ms": {"median": pct(step_ms, 50), "p10": pct(step_ms, 10), "p90": pct(step_ms, 90), "n": len(step_ms),
                        "frames_per_s_median": B * T / (pct(step_ms, 50) * 1e-3),
                        "how": "one HIP event per step on a timing stream that waits on the device for all working streams of the step (extract+match and PnP of every sub-batch); the host stays at most 2 steps ahead (BENCH_IN_FLIGHT)"},
            "valid_outputs": ok,
            "runtime_libs": api.loaded_runtime_libs(),
        }
        if parity is not None:
            out.update(parity_checked_streams=parity["parity_checked_streams"], parity=parity)
        if collective is not None:
            out["collective"] = collective
        if ingest is not None:
            out["ingest_inclusive"] = ingest
        if f32_leg is not None:
            out["with_pnp_f32_tier"] = f32_leg
        if kern:
            out.update(roofline_blocks(kern, kern_x, NPROF, SB, NS, value, world))
        if not args.no_ba and world == 1:
            out["roofline_ba"] = ba_roofline(api, ctxs[0])
        if args.cpu_frames > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(fr_a[0], fr_b[0], pnps[0], args.cpu_frames)
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


def ba_roofline(api, ctx, windows: int = 32, reps: int = 20) -> dict:
    """BASELINE configs[3] (localBA residual/Jacobian build, 10 keyframes x 5000 map points, blocks to block-CSR), `windows`
    independent windows per launch: the one kernel family of the path that streams.  Measured live with HIP events on the context's
    stream, after the timed region (never part of `value`); algorithmic bytes = SURVEY 8d's 216 B per edge x the edges of a launch."""
    from monocular_slam_amd import synth
    pr = synth.ba_concat([synth.ba_problem(10, 5000, 4004 + i) for i in range(windows)])
    plan = api.BaPlan(ctx, pr)
    for _ in range(3):
        plan.build_dev()
    ctx.sync()
    ctx.profile_enable(16 * reps)
    for _ in range(reps):
        plan.build_dev()
    prof = {k: v[0] / max(1, v[1]) for k, v in ctx.profile_read().items()}
    ctx.profile_enable(0)
    plan.close()
    E = int(pr["n_edges"])
    alg = 216.0 * E
    dom = max(prof, key=prof.get)
    total = sum(prof.values())
    return {"bound": "hbm", "kernel": dom, "achieved": alg / (prof[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (prof[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": prof[dom], "edges_per_launch": E, "windows_per_launch": windows,
            "whole_build": {"kernels_ms": {k: round(v, 5) for k, v in prof.items()}, "ms": total, "achieved": alg / (total * 1e-3) / 1e9,
                            "frac": alg / (total * 1e-3) / 1e9 / HBM_PEAK_GBS, "window_builds_per_s": windows / (total * 1e-3)},
            "workload": "BASELINE configs[3]: BA residual/Jacobian build, 10 keyframes x 5000 map points per window, J^T J blocks to block-CSR",
            "note": "separate leg after the timed region, one stream, kernels alone on the chip; `frac` prices the dominant kernel with SURVEY 8d's 216 B per edge (the figure of the whole build: the kernel itself moves 195 B per edge of counter traffic, profiles/r03_pmc_summary_ba_probe.json), `whole_build.frac` the same bytes over all three kernels of the build"}


def roofline_blocks(kern, kern_x, nprof, SB, NS, value, world) -> dict:
    """roofline (dominant extract+match kernel, chosen by exclusive time), roofline_path, compute, per-kernel tables."""
    out = {}
    per = {k: v[0] / max(1, v[1]) for k, v in kern.items()}          # avg ms per launch, concurrent schedule (as timed)
    per_x = {k: v[0] / max(1, v[1]) for k, v in kern_x.items()}      # alone on the chip
    frame_steps = nprof * NS                                          # (sub-batch, frame-step) pairs per pass
    tot_x = {k: v[0] for k, v in kern_x.items()} if kern_x else {k: v[0] for k, v in kern.items()}
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:  # PMC passes are collected at a given frames-per-launch: only comparable per launch at that size
            pmc = json.load(open(pmc_path))
            if int(pmc.get("_batch", 32)) != SB or world != 1:
                pmc = {}
        except Exception:
            pmc = {}

    def entry(kname):
        u = KERNEL_UNIT.get(kname, "extract")
        lps = kern[kname][1] / frame_steps                 # launches per sub-batch frame-step (pyramid kernels: > 1)
        alg_k = ALG_BYTES[u] * SB / lps
        dur, dur_x = per[kname], per_x.get(kname)
        ach = alg_k / (dur * 1e-3) / 1e9
        p = pmc.get(kname, {})
        r = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": p.get("hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": alg_k, "avg_launch_ms": dur,
             "launches_per_frame_step": lps, "unit_priced": u}
        if dur_x:
            r["exclusive"] = {"avg_launch_ms": dur_x, "achieved": alg_k / (dur_x * 1e-3) / 1e9, "frac": alg_k / (dur_x * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if p.get("hbm_bytes_per_launch"):
                r["exclusive"]["traffic_GBps"] = p["hbm_bytes_per_launch"] / (dur_x * 1e-3) / 1e9
        for nm, key, peak in (("valu_issue", "valu_insts_per_launch", ISSUE_PEAK_G), ("salu_issue", "salu_insts_per_launch", SALU_PEAK_G)):
            if p.get(key) is not None:
                r[nm] = {"insts_per_launch": p[key], "peak": peak, "unit": "G wave-instr/s",
                         "frac": p[key] / (dur * 1e-3) / 1e9 / peak}
                if dur_x:
                    r[nm]["frac_exclusive"] = p[key] / (dur_x * 1e-3) / 1e9 / peak
        return r

    em = [k for k in tot_x if KERNEL_UNIT.get(k) in ("extract", "match") and k in kern]
    if em:
        dom = max(em, key=lambda k: tot_x[k])   # by exclusive time summed over its launches: deterministic
        out["roofline"] = entry(dom)
        out["roofline"]["note"] = ("dominant extract+match kernel by exclusive time; algorithmic bytes = SURVEY 8d extract unit "
                                   "(W*H + 56*N) x %d frames per launch / launches per frame-step; avg_launch_ms from HIP events on the "
                                   "launching stream under the timed (concurrent, %d streams) schedule -- the figure rocprofv3 "
                                   "--kernel-trace reports for the same command; `exclusive` = the same launch alone on the chip. "
                                   "The kernel is VALU/SALU-issue-bound, not HBM-bound (DESIGN.md section 4)" % (SB, 2 * NS))
        out["roofline_kernels"] = [entry(k) for k in sorted(em, key=lambda k: -tot_x[k])]
    # whole extract+match(+PnP) path: algorithmic and counter bytes per frame x measured frames/s
    alg_frame = ALG_BYTES["extract"] + ALG_BYTES["match"] + ALG_BYTES["pnp"]
    path = {"algorithmic_bytes_per_frame": alg_frame, "algorithmic_GBps": alg_frame * value / world / 1e9,
            "frac": alg_frame * value / world / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    if pmc:
        trf = 0.0
        for k in kern:
            b = pmc.get(k, {}).get("hbm_bytes_per_launch")
            if b:
                trf += b * (kern[k][1] / frame_steps) / SB
        path.update(counter_bytes_per_frame=trf, counter_GBps=trf * value / world / 1e9, counter_frac=trf * value / world / 1e9 / HBM_PEAK_GBS)
    out["roofline_path"] = path
    # f64 RANSAC kernels: priced in work units, not bytes
    comp = {}
    # the scoring kernel that did the work: pnp_score32_k (single-precision first tier, the default) or pnp_score_k (MSLAM_PNP_F32=0;
    # with the tier on it only leaves the problems it does not own)
    sk = "pnp_score32_k" if "pnp_score32_k" in per else "pnp_score_k"
    if sk in per:
        pairs = float(PNP_H) * PNP_P * SB
        d, dx = per[sk], per_x.get(sk)
        comp[sk] = {"G_hyp_pt_per_s": pairs / (d * 1e-3) / 1e9, "avg_launch_ms": d,
                    "arithmetic": "f32 sign test + error bound, f64 for the undecided pairs" if sk == "pnp_score32_k" else "f64 sign test"}
        if sk == "pnp_score_k":
            comp[sk]["GFLOPs_f64"] = pairs * PNP_SCORE_FLOP_PER_PAIR / (d * 1e-3) / 1e9
        if dx:
            comp[sk]["exclusive"] = {"avg_launch_ms": dx, "G_hyp_pt_per_s": pairs / (dx * 1e-3) / 1e9}
            if sk == "pnp_score_k":
                comp[sk]["exclusive"]["GFLOPs_f64"] = pairs * PNP_SCORE_FLOP_PER_PAIR / (dx * 1e-3) / 1e9
    if "pnp_solve_k" in per:
        d, dx = per["pnp_solve_k"], per_x.get("pnp_solve_k")
        comp["pnp_solve_k"] = {"M_hyp_per_s": PNP_H * SB / (d * 1e-3) / 1e6, "avg_launch_ms": d}
        if dx:
            comp["pnp_solve_k"]["exclusive"] = {"avg_launch_ms": dx, "M_hyp_per_s": PNP_H * SB / (dx * 1e-3) / 1e6}
    if comp:
        out["compute"] = comp
    out["kernel_ms_per_frame_step"] = {k: round(v[0] / frame_steps, 5) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])}
    if kern_x:
        out["kernel_ms_per_frame_step_exclusive"] = {k: round(v[0] / frame_steps, 5) for k, v in sorted(kern_x.items(), key=lambda kv: -kv[1][0])}
    return out


if __name__ == "__main__":
    main()
