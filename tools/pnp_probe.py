#!/usr/bin/env python3
"""PnP RANSAC (2000 points x 4096 hypotheses, 32 problems per launch) alone, for rocprofv3 (kernel stats / PMC)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
B, P, H = 32, 2000, 4096
ctx = api.Context(0, 1920, 1080, 4096, B)
D = api.DevBuf.from_array
probs = [synth.pnp_problem(P, 3003 + i) for i in range(B)]
S = synth.sample_table(P, H, 4, 77)
Ss = np.stack([(S + 7 * i) % P for i in range(B)]).astype(np.int32)
pp = api.default_pnp_params(); pp.reproj_thr = 2.0
b = dict(X=D(ctx, np.stack([q["X"] for q in probs])), x=D(ctx, np.stack([q["x"] for q in probs])), K=D(ctx, np.stack([q["K"] for q in probs])),
         d=D(ctx, np.stack([q["dist"] for q in probs])), S=D(ctx, Ss), P=D(ctx, np.full(B, P, np.int32)), Rt=api.DevBuf(ctx, B * 96),
         inl=api.DevBuf(ctx, B * P), cnt=api.DevBuf(ctx, B * H * 4), best=api.DevBuf(ctx, B * 4), ninl=api.DevBuf(ctx, B * 4))
def run():
    ctx.pnp_ransac_batch_dev(b["X"].ptr, b["x"].ptr, b["P"].ptr, P, b["K"].ptr, b["d"].ptr, b["S"].ptr, H, B, pp, b["Rt"].ptr, 0, 0, b["inl"].ptr,
                             b["cnt"].ptr, b["best"].ptr, b["ninl"].ptr)
for _ in range(3): run()
ctx.sync()
ctx.profile_enable(64 * 10)
for _ in range(10): run()
print(json.dumps({k: round(v[0] / 10, 4) for k, v in ctx.profile_read().items()}))
ctx.close()
