#!/usr/bin/env python3
"""Development probe: time of fast_nms_k with its phases cut short (MSLAM_FAST_ABLATE=1..4) on a batch of 1080p frames."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from monocular_slam_amd import api, synth
    from monocular_slam_amd.tracking import TrackingBatch
    B, w, h = 32, 1920, 1080
    ctx = api.Context(0, w, h, 4096, B)
    tb = TrackingBatch(ctx, B, w, h, 2000)
    base = [synth.textured_frame(w, h, 5005 + i) for i in range(4)]
    tb.upload_frames(np.stack([base[i % 4] for i in range(B)]))
    for _ in range(2):
        tb.extract()
    ctx.sync()
    ctx.profile_enable(64 * 5)
    for _ in range(5):
        tb.extract()
    t = ctx.profile_read()
    print(os.environ.get("MSLAM_FAST_ABLATE", "0"), {k: round(v[0] / 5, 4) for k, v in t.items()})
else:
    for a in ("0", "1", "2", "3", "4"):
        env = dict(os.environ, MSLAM_FAST_ABLATE=a)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
