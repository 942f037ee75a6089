"""CPU suite, part 1: the oracle (CPU restatement) against independent numpy statements of the same mathematics,
against the known-answer crumbs the reference holds, and against the committed golden fixtures.

The reference has no golden vectors for this path and cannot be built here: equality of the
oracle with OpenCV 2.4.13 / g2o+Eigen is "parity unpinned"; what IS pinned here is that the oracle implements the
published definitions."""
import ctypes as C
import os

import numpy as np
import pytest

from monocular_slam_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def P(a, t):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ match
def test_match_against_numpy_unpackbits(oracle):
    a, b, perm = synth.descriptor_sets(300, 1)
    rc, idx, dist = oracle.match_knn2(b, a)
    D = (np.unpackbits(b, axis=1)[:, None, :] != np.unpackbits(a, axis=1)[None, :, :]).sum(2)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]  # stable: ties -> lower train index
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_array_equal(dist, np.take_along_axis(D, order, 1))
    qi, ti, d = oracle.ratio_filter(idx, dist, 0.8)
    keep = dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * np.float32(0.8)
    np.testing.assert_array_equal(qi, np.nonzero(keep)[0])
    assert oracle.match_knn2(b, a[:1])[0] == -4  # < 2 train rows (reference :209 is UB there)


# ------------------------------------------------------------------ ORB pieces
def test_level_geometry_matches_survey_table(oracle):
    """level sizes and per-level budgets for the two configurations."""
    lw, lh, sf, inv, nf = oracle.orb_level_sizes(1920, 1080, oracle.orb_default_params(2000))
    assert list(lw) == [1920, 1600, 1333, 1111, 926, 772, 643, 536]
    assert list(lh) == [1080, 900, 750, 625, 521, 434, 362, 301]
    assert list(nf) == [434, 362, 302, 251, 209, 175, 145, 122] and nf.sum() == 2000
    lw, lh, sf, inv, nf = oracle.orb_level_sizes(640, 480, oracle.orb_default_params(500))
    assert list(nf) == [109, 90, 75, 63, 52, 44, 36, 31]
    k = (C.c_int * 7)()
    oracle.lib().orc_gaussian_kernel7_q8(k)
    assert list(k) == [18, 34, 49, 55, 49, 34, 18]


def _np_resize(src, dw, dh):
    """independent vectorised statement of OpenCV's 8U INTER_LINEAR fixed-point resize"""
    sh, sw = src.shape
    def tab(d, s):
        scale = 1.0 / (np.float64(d) / s)
        f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        return i, f
    sx, fx = tab(dw, sw)
    fx = np.where((sx < 0) | (sx >= sw - 1), np.float32(0), fx); sx = np.clip(sx, 0, sw - 1)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64); a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    sy, fy = tab(dh, sh)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64); b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    y0 = np.clip(sy, 0, sh - 1); y1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int64)
    x1 = np.minimum(sx + 1, sw - 1)
    Hrow = s[:, sx] * a0 + s[:, x1] * a1
    S0, S1 = Hrow[y0], Hrow[y1]
    return (((((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2)).astype(np.uint8)


def _np_blur(img):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    row = sum(k[j] * p[:, j:j + w] for j in range(7))
    col = sum(k[j] * row[j:j + h, :] for j in range(7))
    return np.clip((col + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def test_pyramid_and_blur_against_numpy(oracle):
    img = synth.textured_frame(203, 157, 3)
    p = oracle.orb_default_params(100)
    lw, lh, *_ = oracle.orb_level_sizes(203, 157, p)
    cur = img
    for l in range(p.nlevels):
        if l > 0:
            cur = _np_resize(cur, int(lw[l]), int(lh[l]))
        np.testing.assert_array_equal(oracle.orb_level(img, p, l, False), cur, err_msg=f"level {l}")
        np.testing.assert_array_equal(oracle.orb_level(img, p, l, True), _np_blur(cur), err_msg=f"blur {l}")
    # BGR -> gray fixed point
    bgr = np.random.default_rng(1).integers(0, 256, (157, 203, 3), dtype=np.uint8)
    g = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14)
    np.testing.assert_array_equal(oracle.orb_level(bgr, p, 0, False), g.astype(np.uint8))


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = np.array([int(img[y + dy, x + dx]) for dx, dy in RING])
    for sign in (1, -1):
        m = (sign * (ring - v) > t)
        mm = np.concatenate([m, m])
        run = 0
        for b in mm:
            run = run + 1 if b else 0
            if run >= 9:
                return True
    return False


def test_fast_score_and_nms_by_definition(oracle):
    """score = largest threshold for which the pixel still passes the segment test (by brute force)."""
    rng = np.random.default_rng(7)
    img = (rng.integers(0, 2, (40, 56)) * rng.integers(60, 200, (40, 56)) + rng.integers(0, 30, (40, 56))).astype(np.uint8)
    img = np.kron(img[::4, ::4], np.ones((4, 4), np.uint8))[:40, :56] + rng.integers(0, 12, (40, 56)).astype(np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    oracle.lib().orc_fast_nms_map(P(np.ascontiguousarray(img), oracle.u8p), w, h, w, 20, P(out, oracle.u8p))
    score = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if _is_corner(img, x, y, 20):
                t = 20
                while t < 255 and _is_corner(img, x, y, t + 1):
                    t += 1
                score[y, x] = t
    exp = np.zeros_like(score)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx or dy)):
                exp[y, x] = s
    assert (score > 0).sum() > 20
    np.testing.assert_array_equal(out, exp)


def test_harris_angle_descriptor_against_numpy(oracle):
    img = synth.textured_frame(160, 120, 9)
    L = oracle.lib()
    f32 = np.float32
    for (x, y) in [(40, 40), (77, 61), (100, 33)]:
        win = img[y - 4:y + 5, x - 4:x + 5].astype(np.int64)
        Ix = (win[1:-1, 2:] - win[1:-1, :-2]) * 2 + (win[:-2, 2:] - win[:-2, :-2]) + (win[2:, 2:] - win[2:, :-2])
        Iy = (win[2:, 1:-1] - win[:-2, 1:-1]) * 2 + (win[2:, :-2] - win[:-2, :-2]) + (win[2:, 2:] - win[:-2, 2:])
        a, b, c = int((Ix * Ix).sum()), int((Iy * Iy).sum()), int((Ix * Iy).sum())
        sc = f32(1.0) / (f32(4 * 7) * f32(255.0))
        s4 = f32(f32(f32(sc * sc) * sc) * sc)
        ab = f32(f32(a) * f32(b)); cc = f32(f32(c) * f32(c)); sm = f32(f32(a) + f32(b))
        exp = f32(f32(f32(ab - cc) - f32(f32(f32(0.04) * sm) * sm)) * s4)
        got = L.orc_harris_response(P(img, oracle.u8p), 160, x, y)
        assert f32(got) == exp
        # intensity centroid
        umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                m10 += u * int(img[y + v, x + u]); m01 += v * int(img[y + v, x + u])
        ang = L.orc_ic_angle(P(img, oracle.u8p), 160, x, y)
        assert ang == L.orc_fast_atan2(C.c_float(m01), C.c_float(m10))
        true = np.degrees(np.arctan2(m01, m10)) % 360
        assert min(abs(ang - true), 360 - abs(ang - true)) < 0.35  # fastAtan2 is a ~0.3 degree approximation
        # descriptor
        pat = np.loadtxt(os.path.join(GOLD, "orb_pattern_31.txt"), dtype=int)
        blur = _np_blur(img)
        desc = np.zeros(32, np.uint8)
        L.orc_orb_descriptor(P(blur, oracle.u8p), 160, x, y, C.c_float(ang), P(desc, oracle.u8p))
        ar = f32(f32(ang) * f32(np.pi / 180.0))
        ca, sb = f32(np.cos(np.float64(ar))), f32(np.sin(np.float64(ar)))
        bits = []
        for (x0, y0, x1, y1) in pat:
            def smp(px, py):
                rx = f32(f32(f32(px) * ca) - f32(f32(py) * sb)); ry = f32(f32(f32(px) * sb) + f32(f32(py) * ca))
                return int(blur[y + int(np.rint(ry)), x + int(np.rint(rx))])
            bits.append(smp(x0, y0) < smp(x1, y1))
        np.testing.assert_array_equal(desc, np.packbits(np.array(bits, np.uint8), bitorder="little"))


def test_orb_extract_properties(oracle):
    img = synth.textured_frame(640, 480, 1001)
    p = oracle.orb_default_params(500)
    k, d = oracle.orb_extract(img, p)
    lw, lh, sf, inv, nf = oracle.orb_level_sizes(640, 480, p)
    assert len(k) == 500 and d.shape == (500, 32)
    np.testing.assert_array_equal(np.bincount(k["octave"], minlength=8), nf)
    assert (np.diff(k["octave"]) >= 0).all()                      # level-major
    for l in range(8):
        kk = k[k["octave"] == l]
        key = np.rint(kk["y"] / sf[l]).astype(int) * 4096 + np.rint(kk["x"] / sf[l]).astype(int)
        assert (np.diff(key) > 0).all()                           # row-major inside a level, unique
        assert (kk["x"] / sf[l] >= 31 - 1e-3).all() and (kk["x"] / sf[l] < lw[l] - 31 + 1e-3).all()
        np.testing.assert_array_equal(kk["size"], np.float32(31) * sf[l])
    assert ((k["angle"] >= 0) & (k["angle"] < 360)).all()
    # nfeatures below the number of levels: zero-budget levels are cleared (retainBest(0))
    p2 = oracle.orb_default_params(5)
    k2, _ = oracle.orb_extract(img, p2)
    assert len(k2) == 5 == oracle.orb_level_sizes(640, 480, p2)[4].sum()


# ------------------------------------------------------------------ linear algebra / geometry
def test_jacobi_and_null_vector(oracle):
    rng = np.random.default_rng(0)
    L = oracle.lib()
    for n in (3, 4, 9, 12):
        A = rng.normal(size=(n, n)); A = A @ A.T
        w = np.zeros(n); V = np.zeros((n, n))
        L.orc_jacobi_eig(P(A, oracle.f64p), n, P(w, oracle.f64p), P(V, oracle.f64p))
        assert np.abs(A @ V - V * w).max() < 1e-12 * np.abs(A).max() and (np.diff(w) >= 0).all()
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-13
    for _ in range(20):
        A = rng.normal(size=(8, 9))
        f = np.zeros(9)
        assert L.orc_null_8x9(P(A.copy(), oracle.f64p), P(f, oracle.f64p)) == 1
        assert np.abs(A @ f).max() < 1e-12 and abs(np.linalg.norm(f) - 1) < 1e-14
        assert abs(abs(f @ np.linalg.svd(A)[2][-1]) - 1) < 1e-10   # == last right singular vector up to sign
    A = rng.normal(size=(3, 3))
    U, s, Vt = np.zeros(9), np.zeros(3), np.zeros(9)
    L.orc_svd3(P(A, oracle.f64p), P(U, oracle.f64p), P(s, oracle.f64p), P(Vt, oracle.f64p))
    assert np.abs(U.reshape(3, 3) @ np.diag(s) @ Vt.reshape(3, 3) - A).max() < 1e-13
    np.testing.assert_allclose(s, np.linalg.svd(A)[1], rtol=1e-12)


def test_fmat_geometry(oracle):
    tv = synth.two_view_problem(400, 11, noise=0.0, outlier_frac=0.0)
    S = synth.sample_table(400, 64, 8, 5)
    p = oracle.fmat_default_params(); p.confidence = 0.0
    r = oracle.fmat_ransac8(tv["x1"], tv["x2"], S, p)
    F = r["F"]
    assert r["rc"] == 0 and r["n_inliers"] == 400 and abs(F[2, 2] - 1) < 1e-12
    x1h = np.c_[tv["x1"], np.ones(400)]; x2h = np.c_[tv["x2"], np.ones(400)]
    assert np.abs(np.einsum("ij,jk,ik->i", x2h, F, x1h)).max() < 1e-6 * np.abs(F).max() * 1e3
    assert abs(np.linalg.det(F)) < 1e-12 * np.abs(F).max() ** 3 + 1e-18    # rank 2
    # ground-truth F from R, t, K agrees up to scale
    K = tv["K"]; t = tv["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Fgt = np.linalg.inv(K).T @ tx @ tv["R"] @ np.linalg.inv(K)
    Fgt /= Fgt[2, 2]
    np.testing.assert_allclose(F, Fgt, rtol=1e-5, atol=1e-9)
    best, Rt, X, cnt = oracle.pose_from_F(F, K, K, tv["x1"], tv["x2"])
    assert cnt[best] == 400 and np.abs(Rt[:, :3] - tv["R"]).max() < 1e-6
    assert np.abs(Rt[:, 3] / np.linalg.norm(Rt[:, 3]) - t).max() < 1e-6
    np.testing.assert_allclose(X * np.linalg.norm(t) / np.linalg.norm(Rt[:, 3]), tv["X"], rtol=1e-5)  # unit baseline on both sides
    # adaptive bound: with 25 % outliers and confidence 0.85 the scan stops early and never picks a later hypothesis
    tv2 = synth.two_view_problem(500, 11)
    S2 = synth.sample_table(500, 2000, 8, 5)
    p.confidence = 0.85
    r2 = oracle.fmat_ransac8(tv2["x1"], tv2["x2"], S2, p)
    p.confidence = 0.0
    r3 = oracle.fmat_ransac8(tv2["x1"], tv2["x2"], S2, p)
    assert r2["best_h"] <= r3["best_h"] and r3["n_inliers"] >= r2["n_inliers"] > 300


def test_p3p_pnp_triangulate(oracle):
    rng = np.random.default_rng(2)
    pr = synth.pnp_problem(60, 1, noise=0.0, outlier_frac=0.0)
    X, x, K = pr["X"], pr["x"], pr["K"]
    f = np.c_[(x[:, 0] - K[0, 2]) / K[0, 0], (x[:, 1] - K[1, 2]) / K[1, 1], np.ones(60)]
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    for _ in range(100):
        s = rng.choice(60, 3, replace=False)
        R = np.zeros((4, 9)); t = np.zeros((4, 3))
        n = oracle.lib().orc_p3p(P(np.ascontiguousarray(f[s]), oracle.f64p), P(np.ascontiguousarray(X[s]), oracle.f64p),
                                 P(R, oracle.f64p), P(t, oracle.f64p))
        assert 1 <= n <= 4
        errs = []
        for c in range(n):
            Rc = R[c].reshape(3, 3)
            assert np.abs(Rc @ Rc.T - np.eye(3)).max() < 1e-8 and abs(np.linalg.det(Rc) - 1) < 1e-8
            pc = X[s] @ Rc.T + t[c]
            assert (pc[:, 2] > 0).all() and np.abs(pc / np.linalg.norm(pc, axis=1, keepdims=True) - f[s]).max() < 1e-8
            errs.append(np.abs(Rc - pr["R"]).max() + np.abs(t[c] - pr["tvec"]).max())
        assert min(errs) < 1e-8
    for dist_ in (None, synth.DIST_TUM1):
        kw = dict(K=synth.K_TUM1, dist=dist_, w=640, h=480) if dist_ is not None else {}
        pr = synth.pnp_problem(800, 3003, **kw)
        S = synth.sample_table(800, 512, 4, 77)
        pp = oracle.pnp_default_params(); pp.reproj_thr = 2.0
        r = oracle.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, pp)
        assert r["rc"] == 0 and np.abs(r["Rt"][:, :3] - pr["R"]).max() < 2e-4 and np.abs(r["Rt"][:, 3] - pr["tvec"]).max() < 2e-3
        assert r["inliers"][~pr["outlier"]].mean() > 0.98 and r["inliers"][pr["outlier"]].mean() < 0.02
        np.testing.assert_allclose(synth.rodrigues(r["rvec"]), r["Rt"][:, :3], atol=1e-12)
        # refinement lowers the squared reprojection error of the inliers
        pp.refine_iters = 0
        r0 = oracle.pnp_ransac(pr["X"], pr["x"], pr["K"], pr["dist"], S, pp)
        def cost(Rt):
            pc = pr["X"] @ Rt[:, :3].T + Rt[:, 3]
            xd = synth.distort(pc[:, :2] / pc[:, 2:3], pr["dist"])
            uv = np.c_[pr["K"][0, 0] * xd[:, 0] + pr["K"][0, 2], pr["K"][1, 1] * xd[:, 1] + pr["K"][1, 2]]
            return (((uv - pr["x"]) ** 2).sum(1) * r["inliers"]).sum()
        assert cost(r["Rt"]) < cost(r0["Rt"])
    tv = synth.two_view_problem(100, 3, noise=0.0, outlier_frac=0.0)
    Rt2 = np.c_[tv["R"], tv["t"]]
    Xt, front, nf = oracle.triangulate(tv["x1"], tv["x2"], np.eye(3, 4), Rt2, tv["K"], tv["K"])
    np.testing.assert_allclose(Xt, tv["X"], rtol=1e-7)
    assert nf == 100


# ------------------------------------------------------------------ BA
def test_se3quat_known_answer_from_reference_unit_test(oracle):
    """The only known input the reference holds for this path: R, T of UnitTest/g2o/UnitTestg2o.cpp:14-17."""
    Rt = np.array([[0.9506, 0.2940, -0.0998, 2], [-0.2845, 0.9535, 0.0993, 3], [0.1244, -0.0660, 0.9900, 1]])
    p7 = np.zeros(7)
    oracle.lib().orc_Rt_to_se3quat(P(Rt, oracle.f64p), P(p7, oracle.f64p))
    from scipy.spatial.transform import Rotation
    R = Rt[:, :3]
    # Eigen's Quaterniond(R) for trace > 0 (independent formula)
    w = np.sqrt(1 + np.trace(R)) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    q /= np.linalg.norm(q)
    np.testing.assert_allclose(p7[3:], q, rtol=1e-14)
    np.testing.assert_allclose(p7[:3], [2, 3, 1])
    assert p7[6] > 0 and abs(np.linalg.norm(p7[3:]) - 1) < 1e-15
    np.testing.assert_allclose(p7[3:], Rotation.from_matrix(R).as_quat(), atol=2e-4)  # R is only rounded to 4 digits
    Rt2 = np.zeros(12)
    oracle.lib().orc_se3quat_to_Rt(P(p7, oracle.f64p), P(Rt2, oracle.f64p))
    np.testing.assert_allclose(Rt2.reshape(3, 4)[:, :3], Rotation.from_quat(p7[3:]).as_matrix(), atol=1e-14)
    # exp: first-order agreement with the matrix exponential, and T <- exp(d) T composition
    d = np.array([0.01, -0.02, 0.015, 0.1, 0.2, -0.1])
    p = p7.copy()
    oracle.lib().orc_se3_exp_left(P(d, oracle.f64p), P(p, oracle.f64p))
    from scipy.linalg import expm
    xi = np.zeros((4, 4)); xi[:3, :3] = [[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]]; xi[:3, 3] = d[3:]
    T0 = np.eye(4); T0[:3, :3] = Rotation.from_quat(p7[3:]).as_matrix(); T0[:3, 3] = p7[:3]
    T1 = expm(xi) @ T0
    np.testing.assert_allclose(Rotation.from_quat(p[3:]).as_matrix(), T1[:3, :3], atol=1e-12)
    np.testing.assert_allclose(p[:3], T1[:3, 3], atol=1e-12)


def test_ba_jacobians_by_central_differences(oracle):
    """g2o's own numeric-Jacobian definition (base_binary_edge.hpp:147-198) as the check of the analytic blocks."""
    pr = synth.ba_problem(3, 12, 5, gross_frac=0.0)
    pr["huber_delta"] = 0.0
    s = oracle.ba_build(pr)
    nf = s["n_free_poses"]

    def residuals(poses, points):
        q = dict(pr); q["poses"] = poses; q["points"] = points
        return oracle.ba_build(q)["err"]

    # H = J^T Omega J and b = -J^T Omega e with J from differences on the point / pose-increment manifold
    E = pr["n_edges"]
    J = np.zeros((2 * E, 6 * nf + 3 * pr["n_points"]))
    h = 1e-6
    free = [j for j in range(pr["n_poses"]) if not pr["pose_fixed"][j]]
    for fi, j in enumerate(free):
        for a in range(6):
            ep = []
            for sgn in (1, -1):
                poses = pr["poses"].copy()
                d = np.zeros(6); d[a] = sgn * h
                pj = poses[j].copy()
                oracle.lib().orc_se3_exp_left(P(d, oracle.f64p), P(pj, oracle.f64p))
                poses[j] = pj
                ep.append(residuals(poses, pr["points"]))
            J[:, 6 * fi + a] = ((ep[0] - ep[1]) / (2 * h)).ravel()
    for i in range(pr["n_points"]):
        for a in range(3):
            ep = []
            for sgn in (1, -1):
                pts = pr["points"].copy(); pts[i, a] += sgn * h
                ep.append(residuals(pr["poses"], pts))
            J[:, 6 * nf + 3 * i + a] = ((ep[0] - ep[1]) / (2 * h)).ravel()
    Om = np.repeat(pr["info"], 2)
    Hd = J.T @ (Om[:, None] * J)
    bd = -J.T @ (Om * s["err"].ravel())
    np.testing.assert_allclose(s["b"], bd, rtol=1e-5, atol=1e-5 * np.abs(bd).max())
    for fi in range(nf):
        np.testing.assert_allclose(s["Hpp"][fi], Hd[6 * fi:6 * fi + 6, 6 * fi:6 * fi + 6], rtol=1e-4, atol=1e-6 * np.abs(Hd).max())
    for i in range(pr["n_points"]):
        blk = Hd[6 * nf + 3 * i:6 * nf + 3 * i + 3, 6 * nf + 3 * i:6 * nf + 3 * i + 3]
        np.testing.assert_allclose(s["Hll"][i], blk, rtol=1e-4, atol=1e-6 * np.abs(Hd).max())
    for f in range(nf):
        for k in range(s["Hpl_rowptr"][f], s["Hpl_rowptr"][f + 1]):
            i = s["Hpl_col"][k]
            np.testing.assert_allclose(s["Hpl_val"][k], Hd[6 * f:6 * f + 6, 6 * nf + 3 * i:6 * nf + 3 * i + 3], rtol=1e-4,
                                       atol=1e-6 * np.abs(Hd).max())


def test_ba_lm_converges_and_fixed_pose_stays(oracle):
    pr = synth.ba_problem(6, 400, 4004)
    r = oracle.ba_optimize(pr, 15)
    assert (np.diff(r["chi2"]) <= 1e-9 * r["chi2"][0]).all() and r["chi2"][-1] < 0.99 * oracle.ba_build(pr)["robust_chi2"]
    np.testing.assert_array_equal(r["poses"][0], pr["poses"][0])      # frame id 0 is fixed (src/Util.cpp:72)
    q = dict(pr); q["poses"] = r["poses"]; q["points"] = r["points"]
    e0, e1 = oracle.ba_build(pr)["err"], oracle.ba_build(q)["err"]   # (monocular BA has a free scale: compare residuals, not poses)
    assert np.median(np.abs(e1)) < 0.8 * np.median(np.abs(e0))
    assert np.abs(np.linalg.norm(r["poses"][:, 3:], axis=1) - 1).max() < 1e-14


def test_ba_robust_cost_and_minimum_against_scipy(oracle):
    """Independent statement of the BA objective (numpy: e = obs - pi(R X + t), chi2 = info |e|^2, g2o's Huber rho) and an
    independent minimiser (scipy.optimize.least_squares with loss='huber', whose rho is the same function): the oracle's
    robust chi2 equals the numpy value, and where the oracle's LM stops scipy finds nothing better."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    pr = synth.ba_problem(4, 60, 77)
    delta = pr["huber_delta"]
    dsqr = float(np.float32(delta * delta))   # g2o keeps delta^2 as a float (robust_kernel_impl.cpp:78-91)
    ep, el, obs, info = pr["edge_pose"], pr["edge_point"], pr["obs"], pr["info"]
    fx, fy, cx, cy = (pr["intr"][ep, k] for k in range(4))           # intrinsics per pose: (fx, fy, cx, cy)

    def residuals_px(poses, points):
        R = Rotation.from_quat(poses[:, 3:7]).as_matrix()          # (x, y, z, w)
        Xc = np.einsum("eij,ej->ei", R[ep], points[el]) + poses[ep, :3]
        proj = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], axis=1)
        return obs - proj

    def robust(poses, points):
        e = residuals_px(poses, points)
        chi2 = info * (e ** 2).sum(axis=1)
        return np.where(chi2 <= dsqr, chi2, 2 * np.sqrt(chi2) * delta - dsqr).sum()

    s0 = oracle.ba_build(pr)
    np.testing.assert_allclose(s0["err"], residuals_px(pr["poses"], pr["points"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(s0["robust_chi2"], robust(pr["poses"], pr["points"]), rtol=1e-12)
    r = oracle.ba_optimize(pr, 40)
    c_lm = robust(r["poses"], r["points"])
    np.testing.assert_allclose(r["chi2"][-1], c_lm, rtol=1e-10)

    # scipy from the oracle's end state: free poses as (rotation vector increment, translation), points
    free = [j for j in range(pr["n_poses"]) if not pr["pose_fixed"][j]]
    R_end = Rotation.from_quat(r["poses"][:, 3:7])

    def unpack(x):
        poses = r["poses"].copy()
        for k, j in enumerate(free):
            dR = Rotation.from_rotvec(x[6 * k:6 * k + 3])
            poses[j, 3:7] = (dR * R_end[j]).as_quat()
            poses[j, :3] = r["poses"][j, :3] + x[6 * k + 3:6 * k + 6]
        pts = r["points"] + x[6 * len(free):].reshape(-1, 3)
        return poses, pts

    # least_squares' Huber acts per scalar residual; g2o's acts on the edge's chi2 (two residuals together): compare at an
    # edge-wise statement instead -- minimise the exact edge-wise robust cost through its square root as a residual vector
    def fun_edge(x):
        poses, pts = unpack(x)
        e = residuals_px(poses, pts)
        chi2 = info * (e ** 2).sum(axis=1)
        rho = np.where(chi2 <= dsqr, chi2, 2 * np.sqrt(chi2) * delta - dsqr)
        return np.sqrt(rho)

    x0 = np.zeros(6 * len(free) + 3 * pr["n_points"])
    assert abs((fun_edge(x0) ** 2).sum() - c_lm) <= 1e-9 * c_lm
    sol = least_squares(fun_edge, x0, method="trf", xtol=1e-12, ftol=1e-12, gtol=1e-12, max_nfev=200)
    c_sp = (sol.fun ** 2).sum()
    # the LM of the reference stops on its own rule (relative decrease < 1e-3 three times): scipy may go on a little, not far
    assert c_sp <= c_lm * (1 + 1e-12) and c_sp >= 0.98 * c_lm, (c_sp, c_lm)


# ------------------------------------------------------------------ golden fixtures
def test_oracle_reproduces_golden_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "orb_320x240_seed4242.npz"))
    k, d = oracle.orb_extract(g["img"], oracle.orb_default_params(300))
    np.testing.assert_array_equal(k, g["kps"]); np.testing.assert_array_equal(d, g["desc"])
    g = np.load(os.path.join(GOLD, "match_400_seed2003.npz"))
    rc, idx, dist = oracle.match_knn2(g["q"], g["t"])
    np.testing.assert_array_equal(idx, g["idx"]); np.testing.assert_array_equal(dist, g["dist"])
    g = np.load(os.path.join(GOLD, "fmat_300_seed11.npz"))
    r = oracle.fmat_ransac8(g["x1"], g["x2"], g["samples"], oracle.fmat_default_params())
    np.testing.assert_array_equal(r["counts"], g["counts"]); assert r["best_h"] == int(g["best_h"])
    np.testing.assert_array_equal(r["status"], g["status"]); np.testing.assert_allclose(r["F"], g["F"], rtol=1e-9, atol=1e-15)
    g = np.load(os.path.join(GOLD, "pnp_400_seed3003.npz"))
    pp = oracle.pnp_default_params(); pp.reproj_thr = 2.0
    r = oracle.pnp_ransac(g["X"], g["x"], g["K"], g["dist"], g["samples"], pp)
    np.testing.assert_array_equal(r["counts"], g["counts"]); assert r["best_h"] == int(g["best_h"])
    np.testing.assert_array_equal(r["inliers"], g["inliers"]); np.testing.assert_allclose(r["Rt"], g["Rt"], rtol=1e-10, atol=1e-14)
    g = np.load(os.path.join(GOLD, "ba_4x150_seed4004.npz"))
    pr = {k[3:]: (g[k] if g[k].ndim else g[k].item()) for k in g.files if k.startswith("in_")}
    s = oracle.ba_build(pr)
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        np.testing.assert_allclose(s[k], g[k], rtol=1e-12, atol=1e-12 * np.abs(g[k]).max())
    np.testing.assert_array_equal(s["Hpl_col"], g["Hpl_col"])


def test_descriptor_trig_kernel_equals_libm_after_float_cast(tmp_path):
    """The GPU's quadrant-reduced sin/cos (csrc/sincos_pio4.h, plain C) vs libm, compared as the descriptor uses them:
    (float)f((double)a) for float a in [0, 2 pi].  Every 5th float here (4 s); `trig_check 1` is the exhaustive run
    (1 086 918 620 values, 0 differences, 14 s)."""
    import subprocess
    exe = tmp_path / "trig_check"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trig_check.c")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-o", str(exe), src, "-lm"], check=True)
    r = subprocess.run([str(exe), "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "differences 0" in r.stdout


# ---------------------------------------------------------------- independent cross-check (scikit-image 0.18.3)
def test_oracle_fast_and_hamming_agree_with_scikit_image_vectors(oracle):
    """tests/golden/skimage_crosscheck.npz was written by the one independent implementation in the build image
    (scikit-image 0.18.3, tests/golden/make_skimage_crosscheck.py).  It is NOT the reference and pins nothing about
    OpenCV 2.4.13 -- parity stays unpinned -- but it is evidence that does not come from this repository: the oracle's
    FAST-9/16 segment-test set (before non-max suppression, threshold 20, strict comparisons) and its brute-force
    Hamming nearest neighbour (ties to the lower train index) must agree with it exactly."""
    z = np.load(os.path.join(GOLD, "skimage_crosscheck.npz"))
    for seed in (11, 12):
        img = z[f"img_{seed}"]
        mask = np.unpackbits(z[f"fast_mask_{seed}"])[: img.size].reshape(img.shape).astype(bool)
        mine = oracle.fast_score_map(img, 20) > 0
        assert mask.sum() > 300                      # a meaningful number of corners
        assert (mine == mask).all(), (int(mine.sum()), int(mask.sum()), int((mine != mask).sum()))
        # and the scores the oracle attaches are >= the threshold wherever the segment test passes
        assert (oracle.fast_score_map(img, 20)[mask] >= 20).all()
    for seed in (21, 22):
        a, b, nn = z[f"desc_a_{seed}"], z[f"desc_b_{seed}"], z[f"nn_{seed}"]
        rc, idx, dist = oracle.match_knn2(a, b)
        assert rc == 0 and (idx[:, 0] == nn).all()
        assert (b[5] == b[3]).all() and (idx[:, 0] != 5).all() and (idx[:, 0] != 200).all()   # duplicates: the lower index wins


# ---- 7-point estimator (findFundamentalMat(CV_FM_RANSAC) as the reference calls it, src/CameraPoseEstimator.cpp:563)
def test_cubic_solver_against_numpy_roots_and_opencv_root_order():
    from oracle import binding as ob
    rng = np.random.default_rng(5)
    for _ in range(500):
        c = rng.normal(size=4) * 10.0 ** rng.uniform(-3, 3, size=4)
        r = ob.solve_cubic(*c)
        ref = np.roots(c)
        ref = np.sort(ref[np.abs(ref.imag) < 1e-9 * np.maximum(1, np.abs(ref))].real)
        assert len(r) == len(ref), (c, r, ref)
        np.testing.assert_allclose(np.sort(r), ref, rtol=1e-6, atol=1e-9)
        if len(r) == 3:   # cvSolveCubic's order for three real roots: smallest, largest, middle
            assert r[0] <= r[2] <= r[1]
    assert list(ob.solve_cubic(0, 1, -3, 2)) == [2.0, 1.0] and list(ob.solve_cubic(0, 0, 2, -1)) == [0.5] and len(ob.solve_cubic(0, 0, 0, 1)) == 0
    np.testing.assert_allclose(np.sort(ob.solve_cubic(1, -6, 11, -6)), [1, 2, 3], rtol=1e-12)


def test_seven_point_matrices_are_rank_two_and_one_of_them_is_the_true_one():
    from oracle import binding as ob
    from monocular_slam_amd import synth
    found = 0
    for seed in range(20):
        tv = synth.two_view_problem(7, 300 + seed, noise=0.0, outlier_frac=0.0)
        Fs = ob.fmat_7point(tv["x1"], tv["x2"])
        assert 1 <= len(Fs) <= 3
        x1 = np.c_[tv["x1"], np.ones(7)]; x2 = np.c_[tv["x2"], np.ones(7)]
        K = tv["K"]; t = tv["t"]; R = tv["R"]
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Ft = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
        Ft = Ft / Ft[2, 2]
        for F in Fs:
            assert np.abs(np.einsum("ni,ij,nj->n", x2, F, x1)).max() < 1e-7 * max(1.0, np.abs(F).max())
            assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-10
            assert F[2, 2] == 1.0
        found += min(np.abs(F - Ft).max() for F in Fs) < 1e-6 * np.abs(Ft).max()
    assert found == 20


def test_ransac7_estimator_choice_follows_the_number_of_pairs():
    from oracle import binding as ob
    from monocular_slam_amd import synth
    fp = ob.fmat_default_params()
    fp.refit = 0
    # >= 15 pairs: runRANSAC (counts filled, the bound stops early at confidence 0.85); fewer: runLMeDS (no counts, at most 124 samples)
    tv = synth.two_view_problem(400, 7, outlier_frac=0.3)
    S = synth.sample_table(400, 500, 7, 8)
    r = ob.fmat_ransac7(tv["x1"], tv["x2"], S, fp)
    assert r["rc"] == 0 and r["counts"].max() > 200 and r["n_inliers"] == r["counts"].reshape(-1)[r["best_model"]]
    first_full = int(np.argmax(r["counts"].reshape(-1) >= r["n_inliers"]))
    assert r["best_model"] == first_full or r["counts"].reshape(-1)[:r["best_model"]].max() < r["n_inliers"]
    fp.confidence = 0.0
    r0 = ob.fmat_ransac7(tv["x1"], tv["x2"], S, fp)
    assert r0["n_inliers"] == r0["counts"].max() and r0["best_model"] == int(np.argmax(r0["counts"].reshape(-1)))
    assert (r0["status"].astype(bool) == ~tv["outlier"]).mean() > 0.9
    fp = ob.fmat_default_params(); fp.refit = 0
    tv = synth.two_view_problem(12, 9, noise=0.2, outlier_frac=0.0)
    S = synth.sample_table(12, 400, 7, 10)
    r = ob.fmat_ransac7(tv["x1"], tv["x2"], S, fp)
    assert r["rc"] == 0 and not r["counts"].any() and r["best_model"] < 3 * 124 and 7 <= r["n_inliers"] <= 12
    # the median rule: the winner's median error is the least among all matrices of the first 124 samples
    def med(F):
        x1 = np.c_[tv["x1"], np.ones(12)]; x2 = np.c_[tv["x2"], np.ones(12)]
        l2 = x1 @ F.T; l1 = x2 @ F
        d2 = np.einsum("ni,ni->n", x2, l2) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
        d1 = np.einsum("ni,ni->n", x1, l1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
        return np.median(np.maximum(d1, d2).astype(np.float32))
    meds = [med(F) for h in range(124) for F in ob.fmat_7point(tv["x1"], tv["x2"], S[h])]
    assert abs(med(r["F"]) - min(meds)) <= 1e-6 * max(min(meds), 1e-12)
