"""GPU path against the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from monocular_slam_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_orb_golden(ctx):
    g = np.load(os.path.join(GOLD, "orb_320x240_seed4242.npz"))
    k, d = ctx.orb_extract(g["img"], api.default_orb_params(300))
    assert len(k) == len(g["kps"])
    for f in ("x", "y", "size", "angle", "response", "octave"):
        np.testing.assert_array_equal(k[f], g["kps"][f])
    np.testing.assert_array_equal(d, g["desc"])


def test_match_golden(ctx):
    g = np.load(os.path.join(GOLD, "match_400_seed2003.npz"))
    idx, dist = ctx.match_knn2(g["q"], g["t"])
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(dist, g["dist"])


def test_fmat_golden(ctx):
    g = np.load(os.path.join(GOLD, "fmat_300_seed11.npz"))
    r = ctx.fmat_ransac8(g["x1"], g["x2"], g["samples"], api.default_fmat_params())
    np.testing.assert_array_equal(r["counts"], g["counts"])
    assert r["best_h"] == int(g["best_h"]) and r["n_inliers"] == int(g["n_inliers"])
    np.testing.assert_array_equal(r["status"], g["status"])
    np.testing.assert_allclose(r["F"], g["F"], rtol=1e-7, atol=1e-12)


def test_pnp_golden(ctx):
    g = np.load(os.path.join(GOLD, "pnp_400_seed3003.npz"))
    pp = api.default_pnp_params(); pp.reproj_thr = 2.0
    r = ctx.pnp_ransac(g["X"], g["x"], g["K"], g["dist"], g["samples"], pp)
    np.testing.assert_array_equal(r["counts"], g["counts"])
    assert r["best_h"] == int(g["best_h"]) and r["n_inliers"] == int(g["n_inliers"])
    np.testing.assert_array_equal(r["inliers"], g["inliers"])
    np.testing.assert_allclose(r["Rt"], g["Rt"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(r["rvec"], g["rvec"], rtol=1e-9, atol=1e-13)


def test_ba_golden(ctx):
    g = np.load(os.path.join(GOLD, "ba_4x150_seed4004.npz"))
    pr = {k[3:]: (g[k] if g[k].ndim else g[k].item()) for k in g.files if k.startswith("in_")}
    s = ctx.ba_build(pr)
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        np.testing.assert_allclose(s[k], g[k], rtol=1e-9, atol=1e-9 * np.abs(g[k]).max())
    np.testing.assert_array_equal(s["Hpl_col"], g["Hpl_col"])
    np.testing.assert_array_equal(s["Hpl_rowptr"], g["Hpl_rowptr"])
    assert abs(s["robust_chi2"] - float(g["robust_chi2"])) < 1e-9 * float(g["robust_chi2"])
