"""ctypes binding of liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle.h).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4")])


class FmatParams(C.Structure):
    _fields_ = [("metric", C.c_int), ("threshold", C.c_double), ("confidence", C.c_double), ("refit", C.c_int)]


class PnpParams(C.Structure):
    _fields_ = [("reproj_thr", C.c_double), ("confidence", C.c_double), ("refine_iters", C.c_int),
                ("min_inliers", C.c_int), ("stop_inliers", C.c_int)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", f64p), ("pose_fixed", u8p), ("points", f64p),
                ("edge_pose", i32p), ("edge_point", i32p), ("obs", f64p), ("info", f64p),
                ("intr", f64p), ("huber_delta", C.c_double)]


class BaSystem(C.Structure):
    _fields_ = [("err", f64p), ("chi2", f64p), ("rho1", f64p), ("Hpp", f64p), ("Hll", f64p),
                ("Hpl_rowptr", i32p), ("Hpl_col", i32p), ("Hpl_val", f64p), ("b", f64p),
                ("robust_chi2", C.c_double), ("n_free_poses", C.c_int), ("nnzb", C.c_int)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (the checker; building it is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [f for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in srcs)
    if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_harris_response.restype = C.c_float
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        for name in ("orc_det3", "orc_fmat_error"):
            if hasattr(_LIB, name):
                getattr(_LIB, name).restype = C.c_double
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def orb_default_params(nfeatures: int = 500) -> OrbParams:
    p = OrbParams()
    lib().orc_orb_default_params(C.byref(p))
    p.nfeatures = nfeatures
    return p


def orb_level_sizes(w, h, params):
    L = params.nlevels
    lw = np.zeros(L, np.int32); lh = np.zeros(L, np.int32); nf = np.zeros(L, np.int32)
    sf = np.zeros(L, np.float32); inv = np.zeros(L, np.float32)
    lib().orc_orb_level_sizes(w, h, C.byref(params), _p(lw, i32p), _p(lh, i32p), _p(sf, f32p), _p(inv, f32p), _p(nf, i32p))
    return lw, lh, sf, inv, nf


def orb_extract(img: np.ndarray, params: OrbParams, cap: int | None = None):
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    if cap is None:
        cap = params.nfeatures * 2 + 256
    kps = np.zeros(cap, dtype=KP_DTYPE)
    desc = np.zeros((cap, 32), dtype=np.uint8)
    n = C.c_int(0)
    rc = lib().orc_orb_extract(_p(img, u8p), w, h, w * ch, ch, C.byref(params), kps.ctypes.data_as(C.c_void_p),
                               _p(desc, u8p), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"orc_orb_extract rc={rc}")
    return kps[: n.value].copy(), desc[: n.value].copy()


def orb_level(img: np.ndarray, params: OrbParams, level: int, blurred: bool):
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.zeros(w * h, dtype=np.uint8)
    lw, lh = C.c_int(0), C.c_int(0)
    rc = lib().orc_orb_level(_p(img, u8p), w, h, w * ch, ch, C.byref(params), level, int(blurred), _p(out, u8p),
                             C.byref(lw), C.byref(lh))
    if rc != 0:
        raise RuntimeError(f"orc_orb_level rc={rc}")
    return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()


def fast_score_map(img: np.ndarray, thr: int = 20, nms: bool = False) -> np.ndarray:
    """FAST-9/16 cornerScore map of a gray image, before (default) or after the 3x3 non-max suppression."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    (lib().orc_fast_nms_map if nms else lib().orc_fast_score_map)(_p(img, u8p), w, h, w, thr, _p(out, u8p))
    return out


def match_knn2(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.zeros((nq, 2), np.int32); dist = np.zeros((nq, 2), np.int32)
    rc = lib().orc_match_knn2(_p(q, u8p), nq, _p(t, u8p), nt, _p(idx, i32p), _p(dist, i32p))
    return rc, idx, dist


def ratio_filter(idx, dist, ratio: float):
    nq = idx.shape[0]
    qi = np.zeros(nq, np.int32); ti = np.zeros(nq, np.int32); d = np.zeros(nq, np.float32)
    n = C.c_int(0)
    lib().orc_ratio_filter(_p(idx, i32p), _p(dist, i32p), nq, C.c_float(ratio), _p(qi, i32p), _p(ti, i32p), _p(d, f32p), C.byref(n))
    return qi[: n.value].copy(), ti[: n.value].copy(), d[: n.value].copy()


def fmat_default_params() -> FmatParams:
    p = FmatParams()
    lib().orc_fmat_default_params(C.byref(p))
    return p


def pnp_default_params() -> PnpParams:
    p = PnpParams()
    lib().orc_pnp_default_params(C.byref(p))
    return p


def fmat_ransac8(x1, x2, samples, params: FmatParams):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    samples = np.ascontiguousarray(samples, np.int32)
    M, H = x1.shape[0], samples.shape[0]
    F = np.zeros(9); status = np.zeros(M, np.uint8); counts = np.zeros(H, np.int32)
    best, ninl = C.c_int(-1), C.c_int(0)
    rc = lib().orc_fmat_ransac8(_p(x1, f64p), _p(x2, f64p), M, _p(samples, i32p), H, C.byref(params), _p(F, f64p),
                                _p(status, u8p), _p(counts, i32p), C.byref(best), C.byref(ninl))
    return dict(rc=rc, F=F.reshape(3, 3), status=status, counts=counts, best_h=best.value, n_inliers=ninl.value)


def pnp_ransac(X, x, K, dist, samples, params: PnpParams):
    X = np.ascontiguousarray(X, np.float64); x = np.ascontiguousarray(x, np.float64)
    K = np.ascontiguousarray(K, np.float64); dist = np.ascontiguousarray(dist, np.float64)
    samples = np.ascontiguousarray(samples, np.int32)
    P, H = X.shape[0], samples.shape[0]
    rvec, tvec, Rt = np.zeros(3), np.zeros(3), np.zeros(12)
    inl = np.zeros(P, np.uint8); counts = np.zeros(H, np.int32)
    best, ninl = C.c_int(-1), C.c_int(0)
    rc = lib().orc_pnp_ransac(_p(X, f64p), _p(x, f64p), P, _p(K, f64p), _p(dist, f64p), _p(samples, i32p), H,
                              C.byref(params), _p(rvec, f64p), _p(tvec, f64p), _p(Rt, f64p), _p(inl, u8p),
                              _p(counts, i32p), C.byref(best), C.byref(ninl))
    return dict(rc=rc, rvec=rvec, tvec=tvec, Rt=Rt.reshape(3, 4), inliers=inl, counts=counts, best_h=best.value,
                n_inliers=ninl.value)


def triangulate(x1, x2, Rt1, Rt2, K1, K2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    n = x1.shape[0]
    X = np.zeros((n, 3)); front = np.zeros(n, np.uint8); nf = C.c_int(0)
    a = [np.ascontiguousarray(m, np.float64) for m in (Rt1, Rt2, K1, K2)]
    lib().orc_triangulate(_p(x1, f64p), _p(x2, f64p), n, _p(a[0], f64p), _p(a[1], f64p), _p(a[2], f64p), _p(a[3], f64p),
                          _p(X, f64p), _p(front, u8p), C.byref(nf))
    return X, front, nf.value


def pose_from_F(F, K1, K2, x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    n = x1.shape[0]
    a = [np.ascontiguousarray(m, np.float64) for m in (F, K1, K2)]
    Rt = np.zeros(12); X = np.zeros((n, 3)); counts = np.zeros(4, np.int32)
    best = lib().orc_pose_from_F(_p(a[0], f64p), _p(a[1], f64p), _p(a[2], f64p), _p(x1, f64p), _p(x2, f64p), n,
                                 _p(Rt, f64p), _p(X, f64p), _p(counts, i32p))
    return best, Rt.reshape(3, 4), X, counts


def _ba_problem_struct(pr, cls=None):
    """pr: dict from synth.ba_problem -> (struct, keepalive list)"""
    cls = cls or BaProblem
    keep = dict(poses=np.ascontiguousarray(pr["poses"], np.float64), pose_fixed=np.ascontiguousarray(pr["pose_fixed"], np.uint8),
                points=np.ascontiguousarray(pr["points"], np.float64), edge_pose=np.ascontiguousarray(pr["edge_pose"], np.int32),
                edge_point=np.ascontiguousarray(pr["edge_point"], np.int32), obs=np.ascontiguousarray(pr["obs"], np.float64),
                info=np.ascontiguousarray(pr["info"], np.float64), intr=np.ascontiguousarray(pr["intr"], np.float64))
    s = cls()
    s.n_poses, s.n_points, s.n_edges = pr["n_poses"], pr["n_points"], pr["n_edges"]
    s.poses = _p(keep["poses"], f64p); s.pose_fixed = _p(keep["pose_fixed"], u8p); s.points = _p(keep["points"], f64p)
    s.edge_pose = _p(keep["edge_pose"], i32p); s.edge_point = _p(keep["edge_point"], i32p)
    s.obs = _p(keep["obs"], f64p); s.info = _p(keep["info"], f64p); s.intr = _p(keep["intr"], f64p)
    s.huber_delta = pr["huber_delta"]
    return s, keep


def _ba_system_alloc(n_edges, n_points, nf, nnzb, cls=None):
    cls = cls or BaSystem
    a = dict(err=np.zeros((n_edges, 2)), chi2=np.zeros(n_edges), rho1=np.zeros(n_edges), Hpp=np.zeros((nf, 6, 6)),
             Hll=np.zeros((n_points, 3, 3)), Hpl_rowptr=np.zeros(nf + 1, np.int32), Hpl_col=np.zeros(max(nnzb, 1), np.int32),
             Hpl_val=np.zeros((max(nnzb, 1), 6, 3)), b=np.zeros(6 * nf + 3 * n_points))
    s = cls()
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        setattr(s, k, _p(a[k], f64p))
    s.Hpl_rowptr = _p(a["Hpl_rowptr"], i32p); s.Hpl_col = _p(a["Hpl_col"], i32p)
    return s, a


def ba_build(pr):
    ps, keep = _ba_problem_struct(pr)
    nf, nnzb = C.c_int(0), C.c_int(0)
    lib().orc_ba_sizes(C.byref(ps), C.byref(nf), C.byref(nnzb))
    ss, a = _ba_system_alloc(pr["n_edges"], pr["n_points"], nf.value, nnzb.value)
    rc = lib().orc_ba_build(C.byref(ps), C.byref(ss))
    assert rc == 0
    a["Hpl_col"] = a["Hpl_col"][: nnzb.value]; a["Hpl_val"] = a["Hpl_val"][: nnzb.value]
    a.update(robust_chi2=ss.robust_chi2, n_free_poses=ss.n_free_poses, nnzb=ss.nnzb)
    return a


def ba_optimize(pr, iterations: int, points_fixed: bool = False):
    ps, keep = _ba_problem_struct(pr)
    poses = keep["poses"].copy(); points = keep["points"].copy()
    hist = np.zeros(iterations + 1); nh = C.c_int(0)
    lib().orc_ba_optimize_ex.restype = C.c_int
    done = lib().orc_ba_optimize_ex(C.byref(ps), _p(poses, f64p), _p(points, f64p), iterations, int(points_fixed), _p(hist, f64p), C.byref(nh))
    return dict(poses=poses, points=points, chi2=hist[: nh.value].copy(), iterations=done)


def ba_schur(pr, lam: float):
    ps, keep = _ba_problem_struct(pr)
    nf, nnzb = C.c_int(0), C.c_int(0)
    lib().orc_ba_sizes(C.byref(ps), C.byref(nf), C.byref(nnzb))
    n6 = 6 * nf.value
    S = np.zeros((n6, n6)); bs = np.zeros(n6); x = np.zeros(n6 + 3 * pr["n_points"])
    rc = lib().orc_ba_schur(C.byref(ps), C.c_double(lam), _p(S, f64p), _p(bs, f64p), _p(x, f64p))
    return dict(rc=rc, S=S, bs=bs, x=x)


def solve_cubic(c3, c2, c1, c0):
    r = np.zeros(3)
    lib().orc_solve_cubic.argtypes = [C.c_double] * 4 + [f64p]
    n = lib().orc_solve_cubic(c3, c2, c1, c0, _p(r, f64p))
    return r[:n]


def fmat_7point(x1, x2, sel=None):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    F = np.zeros(27)
    selp = None if sel is None else _p(np.ascontiguousarray(sel, np.int32), i32p)
    n = lib().orc_fmat_7point(_p(x1, f64p), _p(x2, f64p), selp, _p(F, f64p))
    return F.reshape(3, 3, 3)[:n]


def fmat_ransac7(x1, x2, samples, params: FmatParams, method: int = 0):
    """findFundamentalMat(CV_FM_RANSAC) with 7-point samples (H x 7); method 0 = OpenCV's choice by the number of pairs."""
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    samples = np.ascontiguousarray(samples, np.int32)
    M, H = x1.shape[0], samples.shape[0]
    F = np.zeros(9); status = np.zeros(M, np.uint8); counts = np.zeros(3 * H, np.int32)
    best, ninl = C.c_int(-1), C.c_int(0)
    rc = lib().orc_fmat_ransac7(_p(x1, f64p), _p(x2, f64p), M, _p(samples, i32p), H, C.byref(params), method, _p(F, f64p),
                                _p(status, u8p), _p(counts, i32p), C.byref(best), C.byref(ninl))
    return dict(rc=rc, F=F.reshape(3, 3), status=status, counts=counts.reshape(H, 3), best_model=best.value, n_inliers=ninl.value)
