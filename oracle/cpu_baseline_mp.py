#!/usr/bin/env python3
"""All-cores CPU figure for bench.py's cpu_baseline: N independent worker processes, each running the single-threaded
oracle (CPU restatement) on its own stream of the bench workload -- frame-level parallelism, what a multi-camera host
would do with the reference.  Never touches the GPU (run as a separate process by bench.py).
  cpu_baseline_mp.py <workers> <frames_per_worker>  ->  one JSON line"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # oracle/ -> repo root
sys.path.insert(0, ROOT)


def work(args):
    wid, n_frames = args
    import numpy as np
    import bench
    from oracle import binding as ob
    frames = bench.make_frames(2, 5005 + wid)
    X, x, K, dist, samples, _ = bench.make_pnp(2, 3003 + wid)
    p = ob.orb_default_params(bench.NFEAT)
    pp = ob.pnp_default_params(); pp.reproj_thr = bench.PNP_THR
    _, prev = ob.orb_extract(frames[0], p, cap=8192)
    t0 = time.perf_counter()
    for i in range(n_frames):
        b = (i + 1) % 2
        _, desc = ob.orb_extract(frames[b], p, cap=8192)
        rc, idx, d = ob.match_knn2(desc, prev)
        ob.ratio_filter(idx, d, 0.8)
        ob.pnp_ransac(X[b], x[b], K[b], dist[b], samples[b], pp)
        prev = desc
    return time.perf_counter() - t0


if __name__ == "__main__":
    workers, n_frames = int(sys.argv[1]), int(sys.argv[2])
    from oracle import binding as ob
    ob.build()
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        times = pool.map(work, [(w, n_frames) for w in range(workers)])
    wall = max(times)
    print(json.dumps({"value": workers * n_frames / wall, "unit": "frames/s", "cores": workers,
                      "sample": f"{workers} processes x {n_frames} frames, slowest worker {wall:.1f} s"}))
