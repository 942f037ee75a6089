#!/usr/bin/env python3
"""Bare pinned host-to-device copy rate (64 MiB copies on one and on two streams): the ceiling of bench.py's ingest_inclusive leg."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api
ctxs = [api.Context(0, 1920, 1080, 64, 1) for _ in range(2)]
N = 64 << 20
hosts = [api.PinnedBuf(c, (N,), np.uint8) for c in ctxs]
devs = [api.DevBuf(c, N) for c in ctxs]
for nstreams in (1, 2):
    for _ in range(3):
        for i in range(nstreams): ctxs[i].memcpy_h2d_async(devs[i].ptr, hosts[i].ptr, N)
    for c in ctxs: c.sync()
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        for i in range(nstreams): ctxs[i].memcpy_h2d_async(devs[i].ptr, hosts[i].ptr, N)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(nstreams, "stream(s):", round(reps * nstreams * N / dt / 1e9, 1), "GB/s")
