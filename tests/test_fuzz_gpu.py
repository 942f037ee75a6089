"""Seeded differential fuzzing of the HIP path against the oracle: random frame sizes / contents / ORB parameters, random
descriptor sets, random PnP and two-view problems.  MSLAM_FUZZ_CASES (default 10) cases per family; bit-exact where the
path is integer, identical inlier counts and masks for the RANSAC kernels."""
import os

import numpy as np
import pytest

from monocular_slam_amd import api, synth

pytestmark = pytest.mark.gpu
N = int(os.environ.get("MSLAM_FUZZ_CASES", "10"))
SEED = int(os.environ.get("MSLAM_FUZZ_SEED", "0"))   # shifts every generator: a different family of cases


def _params(oracle, p):
    op = oracle.orb_default_params(p.nfeatures)
    for f, _ in api.OrbParams._fields_:
        setattr(op, f, getattr(p, f))
    return op


def _frame(rng, w, h):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return synth.textured_frame(w, h, int(rng.integers(1, 1 << 30)))
    if kind == 1:  # white noise: very dense corners, many score ties
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:  # low-contrast texture: few corners above threshold
        return (100 + synth.textured_frame(w, h, int(rng.integers(1, 1 << 30))).astype(np.int32) // 6).astype(np.uint8)
    if kind == 3:  # saturated blocks: flat areas with sharp edges
        g = rng.integers(0, 2, (h // 16 + 1, w // 16 + 1), dtype=np.uint8) * 255
        return np.ascontiguousarray(np.kron(g, np.ones((16, 16), np.uint8))[:h, :w])
    a = synth.textured_frame(w, h, int(rng.integers(1, 1 << 30)))  # blurred + gradient
    return ((a.astype(np.int32) + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, 2, 1)) // 4 * np.linspace(0.3, 1.0, w)[None, :]).astype(np.uint8)


@pytest.mark.parametrize("case", range(N))
def test_fuzz_orb(ctx, oracle, case):
    rng = np.random.default_rng(9000 + case + SEED)
    w, h = int(rng.integers(120, 1400)), int(rng.integers(100, 1000))
    img = _frame(rng, w, h)
    p = api.default_orb_params(int(rng.integers(50, 3000)))
    p.scale_factor = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5])))
    p.nlevels = int(rng.integers(1, 9))
    p.fast_threshold = int(rng.choice([5, 10, 20, 20, 40]))
    p.score_type = int(rng.integers(0, 2))
    try:
        kg, dg = ctx.orb_extract(img, p)
    except api.MslamError as e:  # response ties may exceed the capacity: the oracle must agree that they do
        assert e.status == api.ERR_CAPACITY
        ko, _ = oracle.orb_extract(img, _params(oracle, p), cap=1 << 16)
        assert len(ko) > ctx.max_kpts or len(ko) > p.nfeatures
        return
    ko, do = oracle.orb_extract(img, _params(oracle, p), cap=1 << 16)
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        np.testing.assert_array_equal(kg[f], ko[f], err_msg=f)
    np.testing.assert_array_equal(dg, do)


@pytest.mark.parametrize("case", range(N))
def test_fuzz_match(ctx, oracle, case):
    rng = np.random.default_rng(7000 + case + SEED)
    nq, nt = int(rng.integers(1, 3000)), int(rng.integers(2, 3000))
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    k = int(rng.integers(0, min(nq, nt)))
    t[rng.permutation(nt)[:k]] = q[rng.permutation(nq)[:k]]          # exact duplicates: distance 0 and ties
    idx, dist = ctx.match_knn2(q, t)
    rc, oidx, odist = oracle.match_knn2(q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)


@pytest.mark.parametrize("case", range(N))
def test_fuzz_pnp_and_fmat(ctx, oracle, case):
    rng = np.random.default_rng(5000 + case + SEED)
    P, H = int(rng.integers(6, 2500)), int(rng.integers(1, 1500))
    use_dist = bool(rng.integers(0, 2))
    pr = synth.pnp_problem(P, 100 + case + SEED, K=synth.K_TUM1 if use_dist else synth.K_MINITAUR, dist=synth.DIST_TUM1 if use_dist else None,
                           w=640 if use_dist else 1920, h=480 if use_dist else 1080, noise=float(rng.uniform(0, 1.5)),
                           outlier_frac=float(rng.uniform(0, 0.6)))
    S = synth.sample_table(P, H, 4, 200 + case + SEED)
    p = api.default_pnp_params()
    p.reproj_thr, p.confidence, p.min_inliers = float(rng.choice([0.5, 2.0, 3.0, 8.0])), float(rng.choice([0.0, 0.9, 0.99])), 4
    op = oracle.pnp_default_params()
    for f, _ in api.PnpParams._fields_:
        setattr(op, f, getattr(p, f))
    try:
        g = ctx.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, p)
    except api.MslamError as e:
        g = dict(rc=e.status)
    o = oracle.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, op)
    assert g["rc"] == o["rc"]
    if g["rc"] == 0:
        np.testing.assert_array_equal(g["counts"], o["counts"])
        np.testing.assert_array_equal(g["inliers"], o["inliers"])
        assert g["best_h"] == o["best_h"]
        assert np.abs(g["Rt"] - o["Rt"]).max() <= 1e-7 * max(1.0, np.abs(o["Rt"]).max())
    M, Hf = int(rng.integers(8, 2500)), int(rng.integers(1, 800))
    tv = synth.two_view_problem(M, 300 + case + SEED, noise=float(rng.uniform(0, 1.0)), outlier_frac=float(rng.uniform(0, 0.5)))
    S8 = synth.sample_table(M, Hf, 8, 400 + case + SEED)
    fp = api.default_fmat_params()
    fp.metric, fp.refit = int(rng.integers(0, 2)), int(rng.integers(0, 3))
    fp.threshold = float(rng.choice([0.1, 1.0, 3.0])) if fp.metric == 1 else float(rng.choice([0.01, 0.1]))
    fp.confidence = float(rng.choice([0.0, 0.85]))
    ofp = oracle.fmat_default_params()
    for f, _ in api.FmatParams._fields_:
        setattr(ofp, f, getattr(fp, f))
    x1, x2 = tv["x1"], tv["x2"]
    if fp.metric == 0:  # the in-tree measure works on normalised coordinates
        Ki = np.linalg.inv(tv["K"])
        x1 = (np.c_[x1, np.ones(M)] @ Ki.T)[:, :2].copy(); x2 = (np.c_[x2, np.ones(M)] @ Ki.T)[:, :2].copy()
    try:
        g = ctx.fmat_ransac8(x1, x2, S8, fp)
    except api.MslamError as e:
        g = dict(rc=e.status)
    o = oracle.fmat_ransac8(x1, x2, S8, ofp)
    assert g["rc"] == o["rc"]
    if g["rc"] == 0:
        np.testing.assert_array_equal(g["counts"], o["counts"])
        np.testing.assert_array_equal(g["status"], o["status"])
        assert g["best_h"] == o["best_h"]
    if g["rc"] in (0, api.ERR_DEGENERATE):
        # without the per-hypothesis counts the library may stop at the adaptive bound (hypothesis ranges): same outcome
        r = ctx.fmat_ransac8(x1, x2, S8, fp, want_counts=False)
        assert r["rc"] == g["rc"] and r["best_h"] == g["best_h"] and r["n_inliers"] == g["n_inliers"]
        np.testing.assert_array_equal(r["status"], g["status"])
        np.testing.assert_array_equal(r["F"], g["F"])
    # the 7-point estimator on pixel coordinates: sizes around the RANSAC / LMedS switch at 15 pairs as often as large ones
    M7 = int(rng.integers(7, 40)) if rng.random() < 0.5 else int(rng.integers(40, 1500))
    H7 = int(rng.integers(1, 400))
    tv7 = synth.two_view_problem(M7, 500 + case + SEED, noise=float(rng.uniform(0, 1.0)), outlier_frac=float(rng.uniform(0, 0.4)))
    S7 = synth.sample_table(M7, H7, 7, 600 + case + SEED)
    fp7 = api.default_fmat_params()
    fp7.refit, fp7.threshold, fp7.confidence = int(rng.choice([0, 2])), float(rng.choice([0.5, 1.0, 3.0])), float(rng.choice([0.0, 0.85, 0.99]))
    ofp7 = oracle.fmat_default_params()
    for f, _ in api.FmatParams._fields_:
        setattr(ofp7, f, getattr(fp7, f))
    method = int(rng.choice([0, 0, 0, 1] + ([2] if M7 <= 32 else [])))
    g = ctx.fmat_ransac7(tv7["x1"], tv7["x2"], S7, fp7, method=method)
    o = oracle.fmat_ransac7(tv7["x1"], tv7["x2"], S7, ofp7, method)
    assert g["rc"] == o["rc"], (M7, H7, method)
    assert g["best_model"] == o["best_model"] and g["n_inliers"] == o["n_inliers"], (M7, H7, method)
    np.testing.assert_array_equal(g["counts"], o["counts"])
    np.testing.assert_array_equal(g["status"], o["status"])
    if g["rc"] == 0:
        assert np.abs(g["F"] - o["F"]).max() <= 1e-6 * max(1e-300, np.abs(o["F"]).max()), (M7, H7, method)
    r = ctx.fmat_ransac7(tv7["x1"], tv7["x2"], S7, fp7, method=method, want_counts=False)   # hypothesis ranges
    assert r["rc"] == g["rc"] and r["best_model"] == g["best_model"] and r["n_inliers"] == g["n_inliers"], (M7, H7, method)
    np.testing.assert_array_equal(r["status"], g["status"])
    np.testing.assert_array_equal(r["F"], g["F"])


@pytest.mark.parametrize("case", range(N))
def test_fuzz_ba_build_and_schur(ctx, oracle, case):
    rng = np.random.default_rng(3000 + case + SEED)
    n_poses, n_points = int(rng.integers(2, 14)), int(rng.integers(5, 2500))
    pr = synth.ba_problem(n_poses, n_points, 600 + case + SEED, obs_noise=float(rng.uniform(0, 3)), gross_frac=float(rng.uniform(0, 0.2)))
    if rng.integers(0, 3) == 0:
        pr["huber_delta"] = 0.0
    extra_fixed = rng.random(n_poses) < 0.2
    extra_fixed[-1] = False                                  # keep at least one free pose
    pr["pose_fixed"] = np.maximum(pr["pose_fixed"], extra_fixed.astype(np.uint8))
    perm = rng.permutation(pr["n_edges"])
    for k in ("edge_pose", "edge_point", "obs", "info"):
        pr[k] = np.ascontiguousarray(pr[k][perm])
    g, o = ctx.ba_build(pr), oracle.ba_build(pr)
    assert g["n_free_poses"] == o["n_free_poses"] and g["nnzb"] == o["nnzb"]
    np.testing.assert_array_equal(g["Hpl_col"], o["Hpl_col"])
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        assert np.abs(g[k] - o[k]).max() <= 1e-9 * max(1e-300, np.abs(o[k]).max()), k
    assert abs(g["robust_chi2"] - o["robust_chi2"]) <= 1e-9 * abs(o["robust_chi2"])
    # Schur complement of the same system (tiles with padding steps, missing blocks, 1 .. 12 local poses), alone and as one window
    # of a block-diagonal batch (tiles end at the seam)
    lam = float(10.0 ** rng.uniform(-4, 3))
    os_ = oracle.ba_schur(pr, lam)
    if os_["rc"] != 0:
        return
    scale, bscale = np.abs(os_["S"]).max(), max(1e-300, np.abs(os_["bs"]).max())
    plan = api.BaPlan(ctx, pr)
    plan.build_dev(); plan.schur_dev(lam)
    S, bs = plan.download_schur()
    plan.close()
    assert np.abs(S - os_["S"]).max() <= 1e-9 * scale and np.abs(bs - os_["bs"]).max() <= 1e-9 * bscale
    other = synth.ba_problem(int(rng.integers(2, 9)), int(rng.integers(5, 300)), 900 + case + SEED)
    cat = synth.ba_concat([pr, other])   # (the batch takes the first problem's Huber delta)
    plan = api.BaPlan(ctx, cat)
    plan.build_dev(); plan.schur_dev(lam)
    Sc, bsc = plan.download_schur()
    plan.close()
    n6 = S.shape[0]
    assert np.abs(Sc[:n6, :n6] - os_["S"]).max() <= 1e-9 * scale and np.abs(bsc[:n6] - os_["bs"]).max() <= 1e-9 * bscale
    assert not Sc[:n6, n6:].any()
