#!/usr/bin/env python3
"""Per-kernel timing of one LM iteration's device work (build + Schur + back-substitution) for B concatenated 10 x 5000 windows."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = api.Context(0, 640, 480, 1024, 1)
pr = synth.ba_concat([synth.ba_problem(10, 5000, 4004 + i) for i in range(B)])
plan = api.BaPlan(ctx, pr)
plan.set_windows(np.repeat(np.arange(B, dtype=np.int32), 10), np.repeat(np.arange(B, dtype=np.int32), 5000), B)
lam = np.full(B, 10.0); xp = np.zeros(6 * plan.view.n_free_poses)
def it():
    plan.build_dev(); plan.schur_windows_dev(lam); plan.backsub_dev(xp, True)
for _ in range(3): it()
ctx.sync()
ctx.profile_enable(64 * 10)
for _ in range(10): it()
print(json.dumps({k: round(v[0] / 10, 4) for k, v in ctx.profile_read().items()}))
ctx.profile_enable(0)
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(10): it()
ctx.record(e1)
print("ms per iteration (incl. host uploads/syncs):", ctx.elapsed_ms(e0, e1) / 10, "edges", pr["n_edges"])
plan.close(); ctx.close()
