#!/usr/bin/env python3
"""Experiment: split the batch over several contexts (HIP streams) to co-schedule kernels of different phases."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from monocular_slam_amd import api
from monocular_slam_amd.tracking import TrackingBatch

B = int(sys.argv[1]); NS = int(sys.argv[2]); steps = 20
sub = B // NS
tbs = []
for i in range(NS):
    c1 = api.Context(0, bench.W, bench.H, 4096, sub); c2 = api.Context(0, bench.W, bench.H, 4096, sub)
    tb = TrackingBatch(c1, sub, bench.W, bench.H, bench.NFEAT, cap=bench.NFEAT + 1024, ratio=0.8, pnp_points=bench.PNP_P, pnp_hyps=bench.PNP_H, pnp_ctx=c2)
    tb.upload_frames(bench.make_frames(sub, 5005 + 10 * i))
    pp = api.default_pnp_params(); pp.reproj_thr = bench.PNP_THR
    tb.set_pnp_problems(*bench.make_pnp(sub, 3003 + i)[:5], pp)
    tbs.append(tb)
for _ in range(3):
    for tb in tbs: tb.step()
for tb in tbs: tb.sync()
t0 = time.perf_counter()
for _ in range(steps):
    for tb in tbs: tb.step()
for tb in tbs: tb.sync()
dt = time.perf_counter() - t0
print(f"B={B} sub-batches={NS}: {B * steps / dt:.0f} frames/s, {1e3 * dt / steps:.3f} ms/step")
