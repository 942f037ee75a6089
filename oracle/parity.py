"""Checker for the batched tracking front end: compares the outputs of the last TrackingBatch step with this CPU oracle.

TEST INFRASTRUCTURE (like everything under oracle/): used by tests/, by bench.py's post-run parity check (which shares
the oracle runs of its cpu_baseline leg) and by __graft_entry__.smoke().  Nothing in the product package imports it.
"""
from __future__ import annotations

import numpy as np

from . import binding as ob


def check_against_oracle(res: dict, streams, frames_cur: np.ndarray, frames_prev: np.ndarray, nfeatures: int, ratio: float,
                         pnp=None, pnp_params=None) -> list:
    """Compare the outputs of the last TrackingBatch step (`res` = results()) for the given streams with the CPU oracle
    Returns a list of mismatch descriptions (empty = parity)."""
    bad = []
    p = ob.orb_default_params(nfeatures)
    for s in streams:
        okc, odc = ob.orb_extract(frames_cur[s], p, cap=8192)
        okp, odp = ob.orb_extract(frames_prev[s], p, cap=8192)
        n = int(res["n"][s])
        if int(res["status"][s]) != 0 or n != len(okc):
            bad.append(f"stream {s}: status {int(res['status'][s])}, n {n} vs oracle {len(okc)}")
            continue
        for f in ("x", "y", "size", "angle", "response", "octave"):
            if not (res["kps"][s][:n][f] == okc[f]).all():
                bad.append(f"stream {s}: keypoint field {f}")
        if not (res["desc"][s][:n] == odc).all():
            bad.append(f"stream {s}: descriptors")
        npv = int(res["n_prev"][s])
        if npv != len(okp) or not (res["desc_prev"][s][:npv] == odp).all():
            bad.append(f"stream {s}: previous-frame descriptors")
            continue
        rc, oidx, odist = ob.match_knn2(odc, odp)
        if rc != 0 or not (res["idx"][s][:n] == oidx).all() or not (res["dist"][s][:n] == odist).all():
            bad.append(f"stream {s}: kNN-2 indices/distances")
        oq, ot, od = ob.ratio_filter(oidx, odist, ratio)
        m = int(res["m_count"][s])
        if m != len(oq) or not (res["m_query"][s][:m] == oq).all() or not (res["m_train"][s][:m] == ot).all() or \
                not (res["m_dist"][s][:m] == od).all():
            bad.append(f"stream {s}: ratio-test survivors ({m} vs {len(oq)})")
        if pnp is not None:
            X, x, K, dist, samples = pnp[:5]
            o = ob.pnp_ransac(X[s], x[s], K[s], dist[s], samples[s], pnp_params)
            if not (res["counts"][s] == o["counts"]).all() or int(res["best"][s]) != o["best_h"] or int(res["ninl"][s]) != o["n_inliers"]:
                bad.append(f"stream {s}: PnP counts/winner")
            if not (res["inl"][s] == o["inliers"]).all():
                bad.append(f"stream {s}: PnP inlier mask")
            if np.abs(res["Rt"][s] - o["Rt"]).max() > 1e-9 * max(1.0, np.abs(o["Rt"]).max()):
                bad.append(f"stream {s}: PnP Rt {np.abs(res['Rt'][s] - o['Rt']).max():.3e}")
    return bad
