#!/usr/bin/env python3
"""Single-frame latency through the host-pointer (drop-in) entry points, PCIe and synchronisation included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocular_slam_amd import api, synth

def med(fn, n=30):
    fn(); fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))

for (w, h, nf) in ((640, 480, 500), (1920, 1080, 2000)):
    with api.Context(0, w, h, 4096, 1) as ctx:
        a = synth.textured_frame(w, h, 11); b = synth.shifted_frame(a, 3, 2, 12)
        p = api.default_orb_params(nf)
        ka, da = ctx.orb_extract(a, p); kb, db = ctx.orb_extract(b, p)
        t_e = med(lambda: ctx.orb_extract(a, p))
        t_m = med(lambda: ctx.match_knn2(db, da))
        pr = synth.pnp_problem(nf, 5); S = synth.sample_table(nf, 256, 4, 6)
        pp = api.default_pnp_params(); pp.reproj_thr = 2.0
        t_p = med(lambda: ctx.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, pp))
        print(f"{w}x{h} nfeatures {nf}: orb_extract {t_e:.3f} ms, match_knn2 {t_m:.3f} ms, pnp_ransac(256 hyps) {t_p:.3f} ms per call (host pointers, synchronous)")
