#!/usr/bin/env python3
"""Hamming kNN-2 (2000 x 2000 descriptors, 32 frame pairs per launch) alone, for rocprofv3 (kernel stats / PMC)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api
B, N, CAP = 32, 2000, 3024
ctx = api.Context(0, 1920, 1080, 4096, B)
rng = np.random.default_rng(1)
q = np.zeros((B, CAP, 32), np.uint8); t = np.zeros((B, CAP, 32), np.uint8)
q[:, :N] = rng.integers(0, 256, (B, N, 32), dtype=np.uint8); t[:, :N] = rng.integers(0, 256, (B, N, 32), dtype=np.uint8)
D = api.DevBuf.from_array
b = dict(q=D(ctx, q), t=D(ctx, t), n=D(ctx, np.full(B, N, np.int32)), idx=api.DevBuf(ctx, B * CAP * 8), dist=api.DevBuf(ctx, B * CAP * 8),
         mq=api.DevBuf(ctx, B * CAP * 4), mt=api.DevBuf(ctx, B * CAP * 4), md=api.DevBuf(ctx, B * CAP * 4), mc=api.DevBuf(ctx, B * 4))
def run():
    ctx.match_knn2_batch_dev(b["q"].ptr, b["n"].ptr, b["t"].ptr, b["n"].ptr, CAP, B, b["idx"].ptr, b["dist"].ptr, 0.8, b["mq"].ptr, b["mt"].ptr, b["md"].ptr, b["mc"].ptr)
for _ in range(3): run()
ctx.sync()
ctx.profile_enable(64 * 10)
for _ in range(10): run()
print(json.dumps({k: round(v[0] / 10, 4) for k, v in ctx.profile_read().items()}))
ctx.close()
