#!/usr/bin/env python3
"""Quick timing probe: batched extract+match at 1920x1080 / 2000 kpts (HIP events on the context stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
from monocular_slam_amd.tracking import TrackingBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h, nf = 1920, 1080, 2000
ctx = api.Context(0, w, h, 4096, B)
tb = TrackingBatch(ctx, B, w, h, nf)
base = [synth.textured_frame(w, h, 5005 + i) for i in range(min(B, 4))]
frames = np.stack([synth.shifted_frame(base[i % len(base)], 3 * (i // len(base)), 2, i) for i in range(B)])
tb.upload_frames(frames)
for _ in range(3):
    tb.step()
ctx.sync()
for name, fn in (("extract", tb.extract), ("match", tb.match_prev), ("step", tb.step)):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(steps):
        fn()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1) / steps
    print(f"{name}: {ms:.3f} ms/batch of {B} -> {ms * 1e3 / B:.1f} us/frame, {B / ms * 1e3:.0f} frames/s")
r = tb.results()
print("n:", r["n"][:8], "matches:", r["m_count"][:8], "status:", r["status"][:4])
