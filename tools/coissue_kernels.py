#!/usr/bin/env python3
"""Do the extract chain (integer kernels) and the PnP chain share the chip better than by taking turns?
Times N iterations of extract alone (stream A), N of the PnP RANSAC alone (stream B), and both enqueued together, by wall clock
around device synchronisation.  MSLAM_PNP_F32=0/1 selects the scoring kernel."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth
from monocular_slam_amd.tracking import TrackingBatch
B, w, h, P, H, N = 32, 1920, 1080, 2000, 4096, 30
ctxE = api.Context(0, w, h, 4096, B)
ctxP = api.Context(0, w, h, 4096, B)
tb = TrackingBatch(ctxE, B, w, h, 2000)
base = [synth.textured_frame(w, h, 5005 + i) for i in range(4)]
tb.upload_frames(np.stack([base[i % 4] for i in range(B)]))
D = api.DevBuf.from_array
probs = [synth.pnp_problem(P, 3003 + i) for i in range(B)]
S = synth.sample_table(P, H, 4, 77)
Ss = np.stack([(S + 7 * i) % P for i in range(B)]).astype(np.int32)
pp = api.default_pnp_params(); pp.reproj_thr = 2.0
b = dict(X=D(ctxP, np.stack([q["X"] for q in probs])), x=D(ctxP, np.stack([q["x"] for q in probs])), K=D(ctxP, np.stack([q["K"] for q in probs])),
         d=D(ctxP, np.stack([q["dist"] for q in probs])), S=D(ctxP, Ss), P=D(ctxP, np.full(B, P, np.int32)), Rt=api.DevBuf(ctxP, B * 96),
         inl=api.DevBuf(ctxP, B * P), cnt=api.DevBuf(ctxP, B * H * 4), best=api.DevBuf(ctxP, B * 4), ninl=api.DevBuf(ctxP, B * 4))
def pnp():
    ctxP.pnp_ransac_batch_dev(b["X"].ptr, b["x"].ptr, b["P"].ptr, P, b["K"].ptr, b["d"].ptr, b["S"].ptr, H, B, pp, b["Rt"].ptr, 0, 0, b["inl"].ptr,
                              b["cnt"].ptr, b["best"].ptr, b["ninl"].ptr)
def timed(fe, fp):
    ctxE.sync(); ctxP.sync()
    t = time.perf_counter()
    for _ in range(N):
        if fe: tb.extract()
        if fp: pnp()
    ctxE.sync(); ctxP.sync()
    return (time.perf_counter() - t) / N * 1e3
for _ in range(3): tb.extract(); pnp()
r = dict(mode=os.environ.get("MSLAM_PNP_F32", "1"), extract_ms=round(timed(True, False), 4), pnp_ms=round(timed(False, True), 4), both_ms=round(timed(True, True), 4))
r["sum_ms"] = round(r["extract_ms"] + r["pnp_ms"], 4)
print(json.dumps(r))
