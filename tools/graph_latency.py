#!/usr/bin/env python3
"""Extract latency of small batches with and without HIP-graph replay (mslam_set_graphs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
from monocular_slam_amd.tracking import TrackingBatch
w, h, nf = 1920, 1080, 2000
for B in (1, 2, 8):
    ctx = api.Context(0, w, h, 4096, B)
    tb = TrackingBatch(ctx, B, w, h, nf)
    tb.upload_frames(np.stack([synth.textured_frame(w, h, 5 + i) for i in range(B)]))
    res = {}
    for g in (False, True):
        ctx.set_graphs(g)
        for _ in range(4):
            tb.extract(); tb.cur = 1 - tb.cur
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(50):
            tb.extract(); tb.cur = 1 - tb.cur
        ctx.record(e1)
        res[g] = ctx.elapsed_ms(e0, e1) / 50
    print(f"B={B}: extract {res[False] * 1e3:.1f} us/launch-chain direct, {res[True] * 1e3:.1f} us with graph replay")
    ctx.close()
