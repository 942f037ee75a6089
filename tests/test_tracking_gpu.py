"""The configuration bench.py times, checked against the oracle: TrackingBatch at 1920x1080 / 2000 keypoints, two
sub-batches on separate HIP streams with the PnP chain on a second stream each, alternating frame sets (current !=
previous), every stream of the batch compared -- keypoints, descriptors, kNN-2 indices/distances, ratio survivors
(bit-exact), PnP counts / winner / inlier mask (identical), Rt (<= 1e-9).  Stands in for the reference's per-frame tick
FeatureExtractor::process -> CameraPoseEstimator::process (src/Pipeline.h:49-65, src/CameraPoseEstimator.cpp:405-409,472)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

W, H, NF, P, HY = 1920, 1080, 2000, 2000, 4096


def _setup(api, SB, NS, seed):
    import bench
    from monocular_slam_amd.tracking import TrackingBatch
    pp = api.default_pnp_params()
    pp.reproj_thr = 2.0
    tbs, fa, fb, pn, ctxs = [], [], [], [], []
    for i in range(NS):
        c1 = api.Context(0, W, H, 4096, SB)
        c2 = api.Context(0, W, H, 4096, SB)
        tb = TrackingBatch(c1, SB, W, H, NF, cap=NF + 1024, ratio=0.8, pnp_points=P, pnp_hyps=HY, pnp_ctx=c2)
        a = bench.make_frames(SB, seed + 17 * i)
        b = bench.make_next_frames(a, seed + 17 * i)
        tb.upload_frames(a, b)
        q = bench.make_pnp(SB, seed + 100 + 17 * i)
        tb.set_pnp_problems(*q[:5], pp)
        tbs.append(tb); fa.append(a); fb.append(b); pn.append(q); ctxs += [c1, c2]
    return tbs, fa, fb, pn, ctxs


def test_tracking_batch_1080p_two_subbatches_all_streams_vs_oracle(oracle):
    from monocular_slam_amd import api
    from oracle import parity
    SB, NS = 4, 2
    tbs, fa, fb, pn, ctxs = _setup(api, SB, NS, 7007)
    opp = oracle.pnp_default_params()
    opp.reproj_thr = 2.0
    try:
        for _ in range(3):           # enqueue three frame-steps on all four streams without a host sync in between
            for tb in tbs:
                tb.step()
        for i, tb in enumerate(tbs):
            res = tb.results()
            cur, prev = tb.last_frame_sets()
            assert (cur, prev) == (0, 1)   # steps read sets 0, 1, 0
            sets = (fa[i], fb[i])
            bad = parity.check_against_oracle(res, range(SB), sets[cur], sets[prev], NF, 0.8, pn[i], opp)
            assert not bad, bad
            # the match is a real one: current and previous frames differ, most best distances are > 0 and a good part survives
            n = res["n"]
            assert all((res["dist"][s][: n[s], 0] > 0).mean() > 0.5 for s in range(SB))
            assert (res["m_count"] > NF // 8).all()
    finally:
        for c in ctxs:
            c.close()


def test_ingest_leg_matches_resident_frames(oracle):
    """Frames uploaded from pinned host memory on a copy stream inside the loop give the same outputs as resident ones."""
    from monocular_slam_amd import api
    SB = 2
    tbs, fa, fb, pn, ctxs = _setup(api, SB, 1, 9009)
    tb = tbs[0]
    cc = api.Context(0, W, H, 64, 1)
    try:
        for _ in range(2):
            tb.step()
        ref = tb.results()
        # clobber the resident sets, then let the ingest path restore them step by step from host memory
        z = np.zeros_like(fa[0])
        tb.upload_frames(z, z)
        tb.enable_ingest(cc, fa[0], fb[0])
        for _ in range(4):
            tb.step()
        got = tb.results()
        for k in ("n", "idx", "dist", "m_count", "counts", "best"):
            assert (ref[k] == got[k]).all(), k
        n = ref["n"]
        for s in range(SB):
            assert (ref["desc"][s][: n[s]] == got["desc"][s][: n[s]]).all()
    finally:
        cc.close()
        for c in ctxs:
            c.close()
