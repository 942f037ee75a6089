#!/usr/bin/env python3
"""F eight-point RANSAC (2000 matches x 4096 hypotheses, 32 problems per launch) alone, for rocprofv3 (kernel stats / PMC)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
B, M, H = 32, 2000, 4096
ctx = api.Context(0, 1920, 1080, 4096, B)
D = api.DevBuf.from_array
tvs = [synth.two_view_problem(M, 21 + i, K=synth.K_MINITAUR, w=1920, h=1080) for i in range(B)]
S8 = synth.sample_table(M, H, 8, 6)
S8s = np.stack([(S8 + 5 * i) % M for i in range(B)]).astype(np.int32)
fp = api.default_fmat_params(); fp.confidence = 0.0
fb = dict(x1=D(ctx, np.stack([q["x1"] for q in tvs])), x2=D(ctx, np.stack([q["x2"] for q in tvs])), M=D(ctx, np.full(B, M, np.int32)),
          S=D(ctx, S8s), F=api.DevBuf(ctx, B * 72), st=api.DevBuf(ctx, B * M), cnt=api.DevBuf(ctx, B * H * 4),
          best=api.DevBuf(ctx, B * 4), ninl=api.DevBuf(ctx, B * 4))
def run():
    ctx.fmat_ransac8_batch_dev(fb["x1"].ptr, fb["x2"].ptr, fb["M"].ptr, M, fb["S"].ptr, H, B, fp, fb["F"].ptr, fb["st"].ptr, fb["cnt"].ptr,
                               fb["best"].ptr, fb["ninl"].ptr)
for _ in range(3): run()
ctx.sync()
ctx.profile_enable(64 * 10)
for _ in range(10): run()
print(json.dumps({k: round(v[0] / 10, 4) for k, v in ctx.profile_read().items()}))
ctx.close()
