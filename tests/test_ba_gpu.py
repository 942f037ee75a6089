"""K9 parity: HIP BA linearisation (through the C ABI) vs the oracle -- floats within 1e-9 relative."""
import numpy as np
import pytest

from monocular_slam_amd import api, synth

pytestmark = pytest.mark.gpu


def _cmp(g, o, rtol=1e-9):
    assert g["n_free_poses"] == o["n_free_poses"] and g["nnzb"] == o["nnzb"]
    np.testing.assert_array_equal(g["Hpl_rowptr"], o["Hpl_rowptr"])
    np.testing.assert_array_equal(g["Hpl_col"], o["Hpl_col"])
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        scale = max(1e-300, np.abs(o[k]).max())
        assert np.abs(g[k] - o[k]).max() <= rtol * scale, k
    assert abs(g["robust_chi2"] - o["robust_chi2"]) <= rtol * abs(o["robust_chi2"])


def test_c3_10x5000(ctx, oracle):
    """C3: 10 keyframes x 5000 map points (50 k edges), Huber on, pose 0 fixed."""
    pr = synth.ba_problem(10, 5000, 4004)
    g = ctx.ba_build(pr)
    o = oracle.ba_build(pr)
    _cmp(g, o)
    assert (o["rho1"] < 1).mean() > 0.05   # the Huber branch is exercised
    # structure: symmetric blocks, one Hpl block per edge of a free pose
    assert np.abs(g["Hpp"] - g["Hpp"].transpose(0, 2, 1)).max() == 0
    assert g["nnzb"] == (pr["edge_pose"] != 0).sum()


@pytest.mark.parametrize("n_poses,n_points,seed,huber", [(2, 50, 1, True), (3, 17, 2, False), (25, 300, 3, True)])
def test_small_problems(ctx, oracle, n_poses, n_points, seed, huber):
    pr = synth.ba_problem(n_poses, n_points, seed)
    if not huber:
        pr["huber_delta"] = 0.0
    _cmp(ctx.ba_build(pr), oracle.ba_build(pr))


def test_shuffled_edge_order_and_two_fixed_poses(ctx, oracle):
    """Edges in arbitrary order, several fixed poses, a point seen once, a point never seen."""
    pr = synth.ba_problem(6, 120, 9)
    rng = np.random.default_rng(4)
    perm = rng.permutation(pr["n_edges"])
    for k in ("edge_pose", "edge_point", "obs", "info"):
        pr[k] = np.ascontiguousarray(pr[k][perm])
    keep = ~((pr["edge_point"] == 5) | ((pr["edge_point"] == 7) & (pr["edge_pose"] != 2)))
    for k in ("edge_pose", "edge_point", "obs", "info"):
        pr[k] = np.ascontiguousarray(pr[k][keep])
    pr["n_edges"] = int(keep.sum())
    pr["pose_fixed"][3] = 1
    g, o = ctx.ba_build(pr), oracle.ba_build(pr)
    _cmp(g, o)
    assert np.abs(g["Hll"][5]).max() == 0 and g["n_free_poses"] == 4


def test_plan_relinearise_is_deterministic(ctx, oracle):
    """Persistent plan: same state -> bit-identical system (no float atomics); new state -> matches oracle."""
    pr = synth.ba_problem(8, 800, 12)
    plan = api.BaPlan(ctx, pr)
    plan.build_dev(); a = plan.download()
    plan.build_dev(); b = plan.download()
    for k in ("Hpp", "Hll", "Hpl_val", "b", "err"):
        np.testing.assert_array_equal(a[k], b[k])
    assert a["robust_chi2"] == b["robust_chi2"]
    pr2 = dict(pr)
    pr2["poses"] = pr["poses_gt"].copy(); pr2["points"] = pr["points_gt"].copy()
    plan.set_state(pr2["poses"], pr2["points"])
    plan.build_dev()
    _cmp(plan.download(), oracle.ba_build(pr2))
    plan.close()


def test_invalid_problem(ctx):
    pr = synth.ba_problem(3, 20, 1)
    pr["edge_pose"] = pr["edge_pose"].copy(); pr["edge_pose"][0] = 99
    with pytest.raises(api.MslamError) as e:
        ctx.ba_build(pr)
    assert e.value.status == api.ERR_INVALID


@pytest.mark.parametrize("n_poses,n_points,seed", [(10, 5000, 4004), (3, 40, 2), (25, 300, 3), (40, 200, 5), (7, 130, 6)])
def test_schur_complement_and_back_substitution(ctx, oracle, n_poses, n_points, seed):
    """f1: S, b_s on the device; x_p from a host solve; x_l by device back-substitution -- vs the oracle's BlockSolver::solve.
    The sizes take each instantiation of ba_schur_tiles_k: tiles of <= 6, <= 13, <= 27 and <= 64 local poses (the last two with
    several passes over the pose pairs), and tiles whose point count is no multiple of four (padding steps)."""
    pr = synth.ba_problem(n_poses, n_points, seed)
    lam = 1e-5 * 1e6
    o = oracle.ba_schur(pr, lam)
    assert o["rc"] == 0
    plan = api.BaPlan(ctx, pr)
    plan.build_dev()
    plan.schur_dev(lam)
    S, bs = plan.download_schur()
    scale = np.abs(o["S"]).max()
    assert np.abs(S - o["S"]).max() <= 1e-9 * scale and np.abs(S - S.T).max() <= 1e-12 * scale
    assert np.abs(bs - o["bs"]).max() <= 1e-9 * np.abs(o["bs"]).max()
    xp = np.linalg.solve(S, bs)
    n6 = 6 * plan.view.n_free_poses
    np.testing.assert_allclose(xp, o["x"][:n6], rtol=1e-6, atol=1e-9 * np.abs(o["x"][:n6]).max())
    p0, X0 = plan.get_state()
    plan.points_snapshot_dev(False)
    plan.backsub_dev(o["x"][:n6], apply_to_points=True)
    x, sc = plan.download_x()
    np.testing.assert_allclose(x[n6:], o["x"][n6:], rtol=1e-7, atol=1e-9 * np.abs(o["x"][n6:]).max())
    b_l = plan.download()["b"][n6:]
    assert abs(sc - np.sum(x[n6:] * (lam * x[n6:] + b_l))) <= 1e-9 * abs(sc)
    p1, X1 = plan.get_state()
    np.testing.assert_allclose(X1, X0 + x[n6:].reshape(-1, 3), rtol=0, atol=1e-15 * np.abs(X0).max() + 1e-15)
    plan.points_snapshot_dev(True)
    np.testing.assert_array_equal(plan.get_state()[1], X0)
    plan.close()


def test_block_diagonal_batch_of_windows(ctx, oracle):
    """Several local-BA windows as one block-diagonal problem (one launch per build): the grid form of the final
    reductions (more than 8192 points) gives the oracle's system and every window's blocks equal its stand-alone build."""
    parts = [synth.ba_problem(6, 3000, 70 + i) for i in range(4)]
    pr = synth.ba_concat(parts)
    assert pr["n_points"] == 12000
    g, o = ctx.ba_build(pr), oracle.ba_build(pr)
    _cmp(g, o)
    plan = api.BaPlan(ctx, pr)
    plan.build_dev(); a = plan.download()
    plan.build_dev(); b = plan.download()
    for k in ("Hpp", "Hll", "Hpl_val", "b"):
        assert np.array_equal(a[k], b[k]), k
    assert a["robust_chi2"] == b["robust_chi2"]
    plan.close()
    # window 2 alone: same Hll / chi2 values (its points do not see the other windows)
    alone = ctx.ba_build(parts[2])
    lo = sum(p["n_points"] for p in parts[:2])
    np.testing.assert_array_equal(g["Hll"][lo:lo + parts[2]["n_points"]], alone["Hll"])
    assert abs(sum(ctx.ba_build(p)["robust_chi2"] for p in parts) - g["robust_chi2"]) <= 1e-9 * g["robust_chi2"]


def test_windows_with_their_own_lambda_equal_separate_schur_complements(ctx, oracle):
    """Four local-BA windows in one block-diagonal plan, a different lambda each (mslam_ba_plan_set_windows +
    mslam_ba_plan_schur_windows_dev): every window's diagonal block of S / b_s and its landmark updates equal what the window
    gives alone with its lambda; blocks between windows are zero; per-window chi2 / computeScale sums come back."""
    parts = [synth.ba_problem(5, 700, 90 + i) for i in range(4)]
    pr = synth.ba_concat(parts)
    lams = np.array([1e-3, 2.0, 50.0, 1e3])
    plan = api.BaPlan(ctx, pr)
    pose_w = np.concatenate([np.full(q["n_poses"], i, np.int32) for i, q in enumerate(parts)])
    point_w = np.concatenate([np.full(q["n_points"], i, np.int32) for i, q in enumerate(parts)])
    plan.set_windows(pose_w, point_w, 4)
    plan.build_dev()
    plan.schur_windows_dev(lams)
    S, bs = plan.download_schur()
    chi_w, _ = plan.download_window_sums()
    xp_all = np.zeros(S.shape[0])
    off = 0
    for i, q in enumerate(parts):
        alone = api.BaPlan(ctx, q)
        alone.build_dev(); alone.schur_dev(float(lams[i]))
        Sa, ba_ = alone.download_schur()
        n6 = Sa.shape[0]
        # (Schur tiles end at the seams between windows, so a window's terms are grouped as they are when it is alone)
        np.testing.assert_allclose(S[off:off + n6, off:off + n6], Sa, rtol=1e-12, atol=1e-12 * np.abs(Sa).max())
        np.testing.assert_allclose(bs[off:off + n6], ba_, rtol=1e-12, atol=1e-12 * np.abs(ba_).max())
        assert not S[off:off + n6, :off].any() and not S[off:off + n6, off + n6:].any()
        assert abs(chi_w[i] - alone.download()["robust_chi2"]) <= 1e-12 * chi_w[i]
        xp_all[off:off + n6] = np.linalg.solve(Sa, ba_)
        alone.close()
        off += n6
    plan.backsub_dev(xp_all, apply_to_points=False)
    x, sc = plan.download_x()
    _, sc_w = plan.download_window_sums()
    assert abs(sc_w.sum() - sc) <= 1e-9 * abs(sc)
    n6 = S.shape[0]
    b_l = plan.download()["b"][n6:]
    xl = x[n6:]
    lo = 0
    for i, q in enumerate(parts):
        n3 = 3 * q["n_points"]
        ref = np.sum(xl[lo:lo + n3] * (lams[i] * xl[lo:lo + n3] + b_l[lo:lo + n3]))
        assert abs(sc_w[i] - ref) <= 1e-9 * abs(ref)
        lo += n3
    plan.close()


def test_duplicate_edges_are_refused_and_wide_tiles_fall_back(ctx):
    """Two edges between one pose and one point need one accumulated block (g2o adds them): refused with a status instead of
    a silently wrong system.  A graph whose 64-point tiles see more than 64 free poses has no device Schur path:
    MSLAM_ERR_CAPACITY tells the caller to run the Schur complement on the host."""
    pr = synth.ba_problem(4, 60, 3)
    dup = dict(pr)
    for k in ("edge_pose", "edge_point", "obs", "info"):
        dup[k] = np.concatenate([pr[k], pr[k][:1]])
    dup["n_edges"] = pr["n_edges"] + 1
    with pytest.raises(api.MslamError) as e:
        api.BaPlan(ctx, dup)
    assert e.value.status == api.ERR_INVALID
    wide = synth.ba_problem(80, 64, 5)
    if np.bincount(wide["edge_point"][~wide["pose_fixed"][wide["edge_pose"]].astype(bool)], minlength=64).size:
        plan = api.BaPlan(ctx, wide)
        plan.build_dev()
        nfree_seen = len(set(wide["edge_pose"][~wide["pose_fixed"][wide["edge_pose"]].astype(bool)].tolist()))
        if nfree_seen > 64:
            with pytest.raises(api.MslamError) as e2:
                plan.schur_dev(1.0)
            assert e2.value.status == api.ERR_CAPACITY
        plan.close()
