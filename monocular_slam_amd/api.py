"""ctypes binding of libmslam_hip.so (C ABI: include/mslam.h).

This is the only way Python reaches the compute path, and there is no fallback: if the
library is not built, or no gfx950 device is present, the calls raise.  The class names
mirror the reference's stage vocabulary (src/FeatureExtractor.cpp, src/CameraPoseEstimator.cpp,
src/Optimiser.cpp); the C++ stage classes under monocular_slam_amd/host bind the same
entry points.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmslam_hip.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p

OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_TOO_FEW, ERR_NO_DEVICE, ERR_DEGENERATE = 0, -1, -2, -3, -4, -5, -6


class MslamError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: status {status}" + (f" ({detail})" if detail else ""))


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


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


class BaDevView(C.Structure):
    _fields_ = [("poses", vp), ("points", vp), ("err", vp), ("chi2", vp), ("rho1", vp),
                ("Hpp", vp), ("Hll", vp), ("Hpl_val", vp), ("b", vp), ("Hpl_rowptr", vp), ("Hpl_col", vp),
                ("robust_chi2", vp), ("n_free_poses", C.c_int), ("nnzb", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4")])

_LIB = None


def lib() -> C.CDLL:
    """Load libmslam_hip.so; raises if it has not been built (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MslamError(ERR_NO_DEVICE, "libmslam_hip.so is not built",
                             "run `python -c 'import __graft_entry__ as g; g.build()'` or make -C monocular_slam_amd/csrc")
        L = C.CDLL(LIB_PATH)
        L.mslam_strerror.restype = C.c_char_p
        L.mslam_last_hip_error.restype = C.c_char_p
        L.mslam_last_hip_error.argtypes = [vp]
        L.mslam_stream.restype = vp
        L.mslam_stream.argtypes = [vp]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def default_fmat_params() -> FmatParams:
    p = FmatParams()
    lib().mslam_fmat_default_params(C.byref(p))
    return p


def default_pnp_params() -> PnpParams:
    p = PnpParams()
    lib().mslam_pnp_default_params(C.byref(p))
    return p


def default_orb_params(nfeatures: int = 500) -> OrbParams:
    p = OrbParams()
    lib().mslam_orb_default_params(C.byref(p))
    p.nfeatures = nfeatures
    return p


class Context:
    """One context per GPU (mslam_create)."""

    def __init__(self, device: int = 0, max_w: int = 1920, max_h: int = 1080, max_kpts: int = 4096, batch: int = 1):
        self._h = vp()
        self.max_kpts = max_kpts
        self.batch = batch
        rc = lib().mslam_create(C.byref(self._h), device, max_w, max_h, max_kpts, batch)
        if rc != OK:
            self._h = None
            raise MslamError(rc, "mslam_create", lib().mslam_strerror(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            for child in list(getattr(self, "_children", [])):   # plans / buffers must go before their context
                try:
                    child.close() if hasattr(child, "close") else child.free()
                except Exception:
                    pass
            lib().mslam_destroy(self._h)
            self._h = None

    def _adopt(self, child):
        import weakref
        if not hasattr(self, "_children"):
            self._children = weakref.WeakSet()
        self._children.add(child)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, what: str):
        if rc != OK:
            detail = lib().mslam_strerror(rc).decode()
            if rc == ERR_HIP:
                detail += ": " + lib().mslam_last_hip_error(self._h).decode()
            raise MslamError(rc, what, detail)

    @property
    def handle(self):
        return self._h

    def sync(self):
        self._check(lib().mslam_sync(self._h), "mslam_sync")

    # ---- timing on the context's stream
    def event(self):
        e = vp()
        self._check(lib().mslam_event_create(self._h, C.byref(e)), "mslam_event_create")
        return e

    def record(self, e):
        self._check(lib().mslam_event_record(self._h, e), "mslam_event_record")

    def wait_event(self, e):
        """This context's stream waits for event `e` (recorded on another context of the same device)."""
        self._check(lib().mslam_stream_wait_event(self._h, e), "mslam_stream_wait_event")

    def memcpy_h2d_async(self, dst_ptr: int, src_ptr: int, nbytes: int):
        self._check(lib().mslam_memcpy_h2d_async(self._h, vp(dst_ptr), vp(src_ptr), C.c_uint64(nbytes)), "mslam_memcpy_h2d_async")

    def memcpy_d2h_async(self, dst_ptr: int, src_ptr: int, nbytes: int):
        self._check(lib().mslam_memcpy_d2h_async(self._h, vp(dst_ptr), vp(src_ptr), C.c_uint64(nbytes)), "mslam_memcpy_d2h_async")

    def memcpy_d2d_async(self, dst_ptr: int, src_ptr: int, nbytes: int):
        self._check(lib().mslam_memcpy_d2d_async(self._h, vp(dst_ptr), vp(src_ptr), C.c_uint64(nbytes)), "mslam_memcpy_d2d_async")

    def elapsed_ms(self, a, b) -> float:
        ms = C.c_float(0)
        self._check(lib().mslam_event_elapsed_ms(self._h, a, b, C.byref(ms)), "mslam_event_elapsed_ms")
        return float(ms.value)

    # ---- per-kernel HIP-event timing
    def set_graphs(self, enable: bool = True):
        """Replay the extract chain as a captured HIP graph (launch-bound small batches)."""
        self._check(lib().mslam_set_graphs(self._h, int(bool(enable))), "mslam_set_graphs")

    def profile_enable(self, max_launches: int):
        self._check(lib().mslam_profile_enable(self._h, max_launches), "mslam_profile_enable")

    def profile_read(self) -> dict:
        """{kernel name: (total ms, launches)} since profile_enable (syncs)."""
        n = lib().mslam_profile_kernel_count()
        ms = np.zeros(n, np.float64); cnt = np.zeros(n, np.int32)
        self._check(lib().mslam_profile_read(self._h, _p(ms, f64p), _p(cnt, i32p)), "mslam_profile_read")
        lib().mslam_profile_kernel_name.restype = C.c_char_p
        return {lib().mslam_profile_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i] > 0}

    # ---- R1-R6
    def orb_extract(self, img: np.ndarray, params: OrbParams, cap: int | None = None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        if cap is None:
            cap = self.max_kpts
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        rc = lib().mslam_orb_extract(self._h, _p(img, u8p), w, h, w * ch, ch, C.byref(params),
                                     kps.ctypes.data_as(vp), _p(desc, u8p), cap, C.byref(n))
        self._check(rc, "mslam_orb_extract")
        return kps[: n.value].copy(), desc[: n.value].copy()

    def orb_debug_level(self, level: int, blurred: bool, max_w: int, max_h: int):
        out = np.zeros(max_w * max_h, dtype=np.uint8)
        lw, lh = C.c_int(0), C.c_int(0)
        self._check(lib().mslam_orb_debug_level(self._h, level, int(blurred), _p(out, u8p), C.byref(lw), C.byref(lh)),
                    "mslam_orb_debug_level")
        return out[: lw.value * lh.value].reshape(lh.value, lw.value).copy()

    def orb_extract_batch_dev(self, imgs_ptr: int, B: int, w: int, h: int, stride: int, frame_stride: int, channels: int,
                              params: OrbParams, kps_ptr: int, desc_ptr: int, cap: int, n_out_ptr: int, status_ptr: int = 0):
        rc = lib().mslam_orb_extract_batch_dev(self._h, vp(imgs_ptr), B, w, h, stride, C.c_int64(frame_stride), channels,
                                               C.byref(params), vp(kps_ptr), vp(desc_ptr), cap, vp(n_out_ptr),
                                               vp(status_ptr) if status_ptr else None)
        self._check(rc, "mslam_orb_extract_batch_dev")

    # ---- R7
    def match_knn2(self, q: np.ndarray, t: np.ndarray):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.int32)
        rc = lib().mslam_match_knn2(self._h, _p(q, u8p), nq, _p(t, u8p), nt, _p(idx, i32p), _p(dist, i32p))
        self._check(rc, "mslam_match_knn2")
        return idx, dist

    def match_knn2_batch_dev(self, q_ptr, nq_ptr, t_ptr, nt_ptr, cap, B, idx_ptr, dist_ptr, ratio=0.0,
                             m_query_ptr=0, m_train_ptr=0, m_dist_ptr=0, m_count_ptr=0):
        def o(p):
            return vp(p) if p else None
        rc = lib().mslam_match_knn2_batch_dev(self._h, vp(q_ptr), vp(nq_ptr), vp(t_ptr), vp(nt_ptr), cap, B,
                                              vp(idx_ptr), vp(dist_ptr), C.c_float(ratio), o(m_query_ptr), o(m_train_ptr),
                                              o(m_dist_ptr), o(m_count_ptr))
        self._check(rc, "mslam_match_knn2_batch_dev")


def ba_problem_struct(pr: dict):
    """dict (as synth.ba_problem) -> (mslam_ba_problem, keepalive arrays)"""
    keep = dict(poses=np.ascontiguousarray(pr["poses"], np.float64), pose_fixed=np.ascontiguousarray(pr["pose_fixed"], np.uint8),
                points=np.ascontiguousarray(pr["points"], np.float64), edge_pose=np.ascontiguousarray(pr["edge_pose"], np.int32),
                edge_point=np.ascontiguousarray(pr["edge_point"], np.int32), obs=np.ascontiguousarray(pr["obs"], np.float64),
                info=np.ascontiguousarray(pr["info"], np.float64), intr=np.ascontiguousarray(pr["intr"], np.float64))
    s = BaProblem()
    s.n_poses, s.n_points, s.n_edges = pr["n_poses"], pr["n_points"], pr["n_edges"]
    s.poses = _p(keep["poses"], f64p); s.pose_fixed = _p(keep["pose_fixed"], u8p); s.points = _p(keep["points"], f64p)
    s.edge_pose = _p(keep["edge_pose"], i32p); s.edge_point = _p(keep["edge_point"], i32p)
    s.obs = _p(keep["obs"], f64p); s.info = _p(keep["info"], f64p); s.intr = _p(keep["intr"], f64p)
    s.huber_delta = pr["huber_delta"]
    return s, keep


def ba_system_alloc(n_edges: int, n_points: int, nf: int, nnzb: int):
    a = dict(err=np.zeros((n_edges, 2)), chi2=np.zeros(n_edges), rho1=np.zeros(n_edges), Hpp=np.zeros((nf, 6, 6)),
             Hll=np.zeros((n_points, 3, 3)), Hpl_rowptr=np.zeros(nf + 1, np.int32), Hpl_col=np.zeros(max(nnzb, 1), np.int32),
             Hpl_val=np.zeros((max(nnzb, 1), 6, 3)), b=np.zeros(6 * nf + 3 * n_points))
    s = BaSystem()
    for k in ("err", "chi2", "rho1", "Hpp", "Hll", "Hpl_val", "b"):
        setattr(s, k, _p(a[k], f64p))
    s.Hpl_rowptr = _p(a["Hpl_rowptr"], i32p); s.Hpl_col = _p(a["Hpl_col"], i32p)
    return s, a


class BaPlan:
    """Device-resident BA structure + state for the LM loop (mslam_ba_plan_*)."""

    def __init__(self, ctx: "Context", pr: dict):
        self.ctx = ctx
        self.pr = pr
        ps, self._keep = ba_problem_struct(pr)
        self._h = vp()
        ctx._check(lib().mslam_ba_plan_create(ctx.handle, C.byref(ps), C.byref(self._h)), "mslam_ba_plan_create")
        self.view = BaDevView()
        ctx._check(lib().mslam_ba_plan_view(self._h, C.byref(self.view)), "mslam_ba_plan_view")
        ctx._adopt(self)

    def set_state(self, poses=None, points=None):
        a = np.ascontiguousarray(poses, np.float64) if poses is not None else None
        b = np.ascontiguousarray(points, np.float64) if points is not None else None
        self.ctx._check(lib().mslam_ba_plan_set_state(self._h, _p(a, f64p) if a is not None else None,
                                                      _p(b, f64p) if b is not None else None), "mslam_ba_plan_set_state")

    def set_robust(self, info=None, huber_delta=0.0):
        a = np.ascontiguousarray(info, np.float64) if info is not None else None
        self.ctx._check(lib().mslam_ba_plan_set_robust(self._h, _p(a, f64p) if a is not None else None,
                                                       C.c_double(huber_delta)), "mslam_ba_plan_set_robust")

    def build_dev(self):
        self.ctx._check(lib().mslam_ba_plan_build_dev(self._h), "mslam_ba_plan_build_dev")

    def download(self):
        nf, nnzb = self.view.n_free_poses, self.view.nnzb
        ss, a = ba_system_alloc(self.pr["n_edges"], self.pr["n_points"], nf, nnzb)
        self.ctx._check(lib().mslam_ba_plan_download(self._h, C.byref(ss)), "mslam_ba_plan_download")
        a["Hpl_col"] = a["Hpl_col"][:nnzb]; a["Hpl_val"] = a["Hpl_val"][:nnzb]
        a.update(robust_chi2=ss.robust_chi2, n_free_poses=ss.n_free_poses, nnzb=ss.nnzb)
        return a

    # ---- Schur complement on the device (f1)
    def schur_dev(self, lam: float):
        self.ctx._check(lib().mslam_ba_plan_schur_dev(self._h, C.c_double(lam)), "mslam_ba_plan_schur_dev")

    def set_windows(self, pose_window, point_window, n_windows: int):
        a = np.ascontiguousarray(pose_window, np.int32); b = np.ascontiguousarray(point_window, np.int32)
        self.n_windows = n_windows
        self.ctx._check(lib().mslam_ba_plan_set_windows(self._h, _p(a, i32p), _p(b, i32p), n_windows), "mslam_ba_plan_set_windows")

    def schur_windows_dev(self, lam):
        a = np.ascontiguousarray(lam, np.float64)
        assert a.size == self.n_windows
        self.ctx._check(lib().mslam_ba_plan_schur_windows_dev(self._h, _p(a, f64p)), "mslam_ba_plan_schur_windows_dev")

    def download_window_sums(self):
        chi = np.zeros(self.n_windows); sc = np.zeros(self.n_windows)
        self.ctx._check(lib().mslam_ba_plan_download_window_sums(self._h, _p(chi, f64p), _p(sc, f64p)), "mslam_ba_plan_download_window_sums")
        return chi, sc

    def download_schur(self):
        n6 = 6 * self.view.n_free_poses
        S = np.zeros((n6, n6)); bs = np.zeros(n6)
        self.ctx._check(lib().mslam_ba_plan_download_schur(self._h, _p(S, f64p), _p(bs, f64p)), "mslam_ba_plan_download_schur")
        return S, bs

    def backsub_dev(self, x_pose: np.ndarray, apply_to_points: bool = False):
        xp = np.ascontiguousarray(x_pose, np.float64)
        self.ctx._check(lib().mslam_ba_plan_backsub_dev(self._h, _p(xp, f64p), int(apply_to_points)), "mslam_ba_plan_backsub_dev")

    def download_x(self):
        x = np.zeros(6 * self.view.n_free_poses + 3 * self.pr["n_points"]); sc = C.c_double(0)
        self.ctx._check(lib().mslam_ba_plan_download_x(self._h, _p(x, f64p), C.byref(sc)), "mslam_ba_plan_download_x")
        return x, sc.value

    def points_snapshot_dev(self, restore: bool):
        self.ctx._check(lib().mslam_ba_plan_points_snapshot_dev(self._h, int(restore)), "mslam_ba_plan_points_snapshot_dev")

    def get_state(self):
        poses = np.zeros((self.pr["n_poses"], 7)); points = np.zeros((self.pr["n_points"], 3))
        self.ctx._check(lib().mslam_ba_plan_get_state(self._h, _p(poses, f64p), _p(points, f64p)), "mslam_ba_plan_get_state")
        return poses, points

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mslam_ba_plan_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ctx_methods():
    def fmat_ransac8(self, x1, x2, samples, params: FmatParams, want_counts: bool = True):
        """R8/R9: eight-point F-matrix RANSAC over an explicit H x 8 sample table.  want_counts=False leaves out the
        per-hypothesis counts, which lets the library stop at the adaptive iteration bound (same winner, same F)."""
        x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
        samples = np.ascontiguousarray(samples, np.int32)
        M, H = x1.shape[0], samples.shape[0]
        F = np.zeros(9); status = np.zeros(M, np.uint8); counts = np.zeros(H, np.int32) if want_counts else None
        best, ninl = C.c_int(-1), C.c_int(0)
        rc = lib().mslam_fmat_ransac8(self._h, _p(x1, f64p), _p(x2, f64p), M, _p(samples, i32p), H, C.byref(params),
                                      _p(F, f64p), _p(status, u8p), _p(counts, i32p) if want_counts else None, C.byref(best), C.byref(ninl))
        if rc not in (OK, ERR_DEGENERATE):
            self._check(rc, "mslam_fmat_ransac8")
        return dict(rc=rc, F=F.reshape(3, 3), status=status, counts=counts, best_h=best.value, n_inliers=ninl.value)

    def fmat_ransac7(self, x1, x2, samples, params: FmatParams, method: int = 0, want_counts: bool = True):
        """R8 as the reference's live path runs it (findFundamentalMat(CV_FM_RANSAC), src/CameraPoseEstimator.cpp:563): 7-point
        samples (H x 7), up to three matrices each; method 0 = OpenCV's choice by the number of pairs (RANSAC / LMedS / direct)."""
        x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
        samples = np.ascontiguousarray(samples, np.int32)
        M, H = x1.shape[0], samples.shape[0]
        assert samples.shape[1] == 7
        F = np.zeros(9); status = np.zeros(M, np.uint8); counts = np.zeros(3 * H, np.int32) if want_counts else None
        best, ninl = C.c_int(-1), C.c_int(0)
        rc = lib().mslam_fmat_ransac7(self._h, _p(x1, f64p), _p(x2, f64p), M, _p(samples, i32p), H, C.byref(params), int(method),
                                      _p(F, f64p), _p(status, u8p), _p(counts, i32p) if want_counts else None, C.byref(best), C.byref(ninl))
        if rc not in (OK, ERR_DEGENERATE):
            self._check(rc, "mslam_fmat_ransac7")
        return dict(rc=rc, F=F.reshape(3, 3), status=status, counts=counts.reshape(H, 3) if want_counts else None,
                    best_model=best.value, n_inliers=ninl.value)

    def pnp_ransac(self, X, x, K, dist, samples, params: PnpParams):
        """R12: PnP RANSAC (P3P hypotheses over an explicit H x 4 sample table)."""
        X = np.ascontiguousarray(X, np.float64); x = np.ascontiguousarray(x, np.float64)
        K = np.ascontiguousarray(K, np.float64); dist = np.ascontiguousarray(dist, np.float64)
        samples = np.ascontiguousarray(samples, np.int32)
        P, H = X.shape[0], samples.shape[0]
        rvec, tvec, Rt = np.zeros(3), np.zeros(3), np.zeros(12)
        inl = np.zeros(P, np.uint8); counts = np.zeros(H, np.int32)
        best, ninl = C.c_int(-1), C.c_int(0)
        rc = lib().mslam_pnp_ransac(self._h, _p(X, f64p), _p(x, f64p), P, _p(K, f64p), _p(dist, f64p), _p(samples, i32p), H,
                                    C.byref(params), _p(rvec, f64p), _p(tvec, f64p), _p(Rt, f64p), _p(inl, u8p),
                                    _p(counts, i32p), C.byref(best), C.byref(ninl))
        if rc not in (OK, ERR_DEGENERATE):
            self._check(rc, "mslam_pnp_ransac")
        return dict(rc=rc, rvec=rvec, tvec=tvec, Rt=Rt.reshape(3, 4), inliers=inl, counts=counts, best_h=best.value,
                    n_inliers=ninl.value)

    def triangulate(self, x1, x2, Rt1, Rt2, K1, K2):
        """R11: two-view DLT triangulation + cheirality flags."""
        x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
        n = x1.shape[0]
        X = np.zeros((n, 3)); front = np.zeros(n, np.uint8); nf = C.c_int(0)
        a = [np.ascontiguousarray(m, np.float64) for m in (Rt1, Rt2, K1, K2)]
        self._check(lib().mslam_triangulate(self._h, _p(x1, f64p), _p(x2, f64p), n, _p(a[0], f64p), _p(a[1], f64p),
                                            _p(a[2], f64p), _p(a[3], f64p), _p(X, f64p), _p(front, u8p), C.byref(nf)),
                    "mslam_triangulate")
        return X, front, nf.value

    def triangulate_batch(self, x1, x2, ofs, Rt1, Rt2, K1, K2):
        """mslam_triangulate_batch: G groups (points ofs[g]..ofs[g+1]) with their own cameras in one launch."""
        x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
        ofs = np.ascontiguousarray(ofs, np.int32)
        G, N = ofs.size - 1, int(ofs[-1])
        a = [np.ascontiguousarray(m, np.float64) for m in (Rt1, Rt2, K1, K2)]
        X = np.zeros((N, 3)); front = np.zeros(max(N, 1), np.uint8); nf = np.zeros(G, np.int32)
        self._check(lib().mslam_triangulate_batch(self._h, _p(x1, f64p), _p(x2, f64p), _p(ofs, i32p), G, _p(a[0], f64p), _p(a[1], f64p),
                                                  _p(a[2], f64p), _p(a[3], f64p), _p(X, f64p), _p(front, u8p), _p(nf, i32p)),
                    "mslam_triangulate_batch")
        return X, front[:N], nf

    def sample_tables_dev(self, n_ptr, seeds_ptr, H, k, B, tables_ptr):
        """mslam_sample_tables_dev: B tables of H x k distinct indices, one row per thread (device pointers)."""
        self._check(lib().mslam_sample_tables_dev(self._h, vp(n_ptr), vp(seeds_ptr), int(H), int(k), int(B), vp(tables_ptr)),
                    "mslam_sample_tables_dev")

    def sample_tables(self, n, seeds, H, k):
        """Array form of sample_tables_dev: n (B,) int32, seeds (B,) uint64 -> (B, H, k) int32."""
        n = np.ascontiguousarray(n, np.int32); seeds = np.ascontiguousarray(seeds, np.uint64)
        B = n.size
        dn, ds = DevBuf.from_array(self, n), DevBuf.from_array(self, seeds)
        dt = DevBuf(self, 4 * B * H * k)
        self.sample_tables_dev(dn.ptr, ds.ptr, H, k, B, dt.ptr)
        out = dt.download(np.int32, (B, H, k))
        for d in (dn, ds, dt):
            d.free()
        return out

    def gather_matches_dev(self, kq_ptr, kt_ptr, cap, query_ptr, train_ptr, count_ptr, B, min_count, x1_ptr, x2_ptr, M_ptr):
        """mslam_gather_matches_dev: matched pixel coordinates (doubles) from resident keypoints + match lists."""
        self._check(lib().mslam_gather_matches_dev(self._h, vp(kq_ptr), vp(kt_ptr), int(cap), vp(query_ptr), vp(train_ptr), vp(count_ptr),
                                                   int(B), int(min_count), vp(x1_ptr), vp(x2_ptr), vp(M_ptr) if M_ptr else None),
                    "mslam_gather_matches_dev")

    def fmat_ransac8_batch_dev(self, x1_ptr, x2_ptr, M_ptr, capM, samples_ptr, H, B, params, F_ptr, status_ptr,
                               counts_ptr, best_ptr, ninl_ptr):
        self._check(lib().mslam_fmat_ransac8_batch_dev(self._h, vp(x1_ptr), vp(x2_ptr), vp(M_ptr), capM, vp(samples_ptr), H, B,
                                                       C.byref(params), vp(F_ptr), vp(status_ptr),
                                                       vp(counts_ptr) if counts_ptr else None, vp(best_ptr), vp(ninl_ptr)),
                    "mslam_fmat_ransac8_batch_dev")

    def fmat_ransac7_batch_dev(self, x1_ptr, x2_ptr, M_ptr, capM, samples_ptr, H, B, params, method, F_ptr, status_ptr,
                               counts_ptr, best_ptr, ninl_ptr):
        self._check(lib().mslam_fmat_ransac7_batch_dev(self._h, vp(x1_ptr), vp(x2_ptr), vp(M_ptr), capM, vp(samples_ptr), H, B,
                                                       C.byref(params), int(method), vp(F_ptr), vp(status_ptr),
                                                       vp(counts_ptr) if counts_ptr else None, vp(best_ptr), vp(ninl_ptr)),
                    "mslam_fmat_ransac7_batch_dev")

    def pnp_ransac_batch_dev(self, X_ptr, x_ptr, P_ptr, capP, K_ptr, dist_ptr, samples_ptr, H, B, params, Rt_ptr,
                             rvec_ptr, tvec_ptr, inl_ptr, counts_ptr, best_ptr, ninl_ptr):
        def o(p):
            return vp(p) if p else None
        self._check(lib().mslam_pnp_ransac_batch_dev(self._h, vp(X_ptr), vp(x_ptr), vp(P_ptr), capP, vp(K_ptr), vp(dist_ptr),
                                                     vp(samples_ptr), H, B, C.byref(params), vp(Rt_ptr), o(rvec_ptr),
                                                     o(tvec_ptr), vp(inl_ptr), o(counts_ptr), vp(best_ptr), vp(ninl_ptr)),
                    "mslam_pnp_ransac_batch_dev")

    def ba_build(self, pr: dict):
        """R15-R17: one linearisation of the BA problem `pr` (dict as synth.ba_problem) -> dict of system blocks."""
        ps, keep = ba_problem_struct(pr)
        nf, nnzb = C.c_int(0), C.c_int(0)
        self._check(lib().mslam_ba_sizes(C.byref(ps), C.byref(nf), C.byref(nnzb)), "mslam_ba_sizes")
        ss, a = ba_system_alloc(pr["n_edges"], pr["n_points"], nf.value, nnzb.value)
        self._check(lib().mslam_ba_build(self._h, C.byref(ps), C.byref(ss)), "mslam_ba_build")
        a["Hpl_col"] = a["Hpl_col"][: nnzb.value]; a["Hpl_val"] = a["Hpl_val"][: nnzb.value]
        a.update(robust_chi2=ss.robust_chi2, n_free_poses=ss.n_free_poses, nnzb=ss.nnzb)
        return a

    for f in (fmat_ransac8, fmat_ransac7, fmat_ransac7_batch_dev, pnp_ransac, triangulate, triangulate_batch, sample_tables_dev, sample_tables, gather_matches_dev,
              fmat_ransac8_batch_dev, pnp_ransac_batch_dev, ba_build):
        setattr(Context, f.__name__, f)


_ctx_methods()


class DevBuf:
    """A device buffer owned through the C ABI (mslam_dev_alloc / mslam_memcpy_*).

    The library is linked against the system HIP runtime; PyTorch wheels bundle their own
    copy, so device pointers are never shared between the two -- torch is used for
    torch.distributed only.
    """

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self._p = vp()
        ctx._check(lib().mslam_dev_alloc(ctx.handle, C.byref(self._p), C.c_uint64(max(1, self.nbytes))), "mslam_dev_alloc")
        ctx._adopt(self)

    @classmethod
    def from_array(cls, ctx: Context, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(ctx, a.nbytes)
        b.upload(a)
        return b

    @property
    def ptr(self) -> int:
        return int(self._p.value or 0)

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        self.ctx._check(lib().mslam_memcpy_h2d(self.ctx.handle, vp(self.ptr + offset), a.ctypes.data_as(vp), C.c_uint64(a.nbytes)),
                        "mslam_memcpy_h2d")

    def download(self, dtype, shape, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        self.ctx._check(lib().mslam_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(vp), vp(self.ptr + offset), C.c_uint64(out.nbytes)),
                        "mslam_memcpy_d2h")
        return out

    def free(self):
        if self._p and self._p.value and self.ctx.handle:
            lib().mslam_dev_free(self.ctx.handle, self._p)
            self._p = vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuf:
    """Page-locked host memory (mslam_host_alloc) viewed as a numpy array; source/target of the async copies."""

    def __init__(self, ctx: Context, shape, dtype=np.uint8):
        self.ctx = ctx
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = vp()
        ctx._check(lib().mslam_host_alloc(ctx.handle, C.byref(self._p), C.c_uint64(max(1, self.nbytes))), "mslam_host_alloc")
        buf = (C.c_uint8 * max(1, self.nbytes)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)
        ctx._adopt(self)

    @property
    def ptr(self) -> int:
        return int(self._p.value or 0)

    def free(self):
        if self._p and self._p.value and self.ctx.handle:
            self.array = None
            lib().mslam_host_free(self.ctx.handle, self._p)
            self._p = vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_preload() -> str:
    """Bind this library's RCCL (the one beside its HIP runtime) now; call before importing torch.  Returns the path."""
    buf = C.create_string_buffer(512)
    rc = lib().mslam_comm_preload(buf, 512)
    if rc != OK:
        raise MslamError(rc, "mslam_comm_preload", lib().mslam_strerror(rc).decode())
    return buf.value.decode()


class Comm:
    """One rank of the RCCL communicator behind mslam_comm_* (one process per GPU)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = lib().mslam_comm_unique_id(buf)
        if rc != OK:
            raise MslamError(rc, "mslam_comm_unique_id", lib().mslam_strerror(rc).decode())
        return bytes(buf)

    def __init__(self, ctx: Context, uid: bytes, rank: int, nranks: int):
        assert len(uid) == 128
        self.ctx, self.rank, self.nranks = ctx, rank, nranks
        self._h = vp()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        ctx._check(lib().mslam_comm_create(ctx.handle, buf, rank, nranks, C.byref(self._h)), "mslam_comm_create")
        ctx._adopt(self)

    def info(self) -> dict:
        r, n, v = C.c_int(0), C.c_int(0), C.c_int(0)
        self.ctx._check(lib().mslam_comm_info(self._h, C.byref(r), C.byref(n), C.byref(v)), "mslam_comm_info")
        return dict(rank=r.value, nranks=n.value, rccl_version=v.value)

    def allgather_i32_dev(self, send_ptr: int, n: int, recv_ptr: int):
        self.ctx._check(lib().mslam_comm_allgather_i32_dev(self._h, vp(send_ptr), n, vp(recv_ptr)), "mslam_comm_allgather_i32_dev")

    def allgather_i32(self, send: np.ndarray) -> np.ndarray:
        send = np.ascontiguousarray(send, np.int32)
        recv = np.zeros(self.nranks * send.size, np.int32)
        self.ctx._check(lib().mslam_comm_allgather_i32(self._h, _p(send, i32p), int(send.size), _p(recv, i32p)), "mslam_comm_allgather_i32")
        return recv

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mslam_comm_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_i32_single_process(ctxs, send: np.ndarray) -> np.ndarray:
    """mslam_allgather_i32: one process, len(ctxs) GPUs; send is (ngpu, n)."""
    send = np.ascontiguousarray(send, np.int32)
    ngpu, n = send.shape
    arr = (vp * ngpu)(*[c.handle for c in ctxs])
    recv = np.zeros(ngpu * n, np.int32)
    rc = lib().mslam_allgather_i32(arr, ngpu, _p(send, i32p), n, _p(recv, i32p))
    ctxs[0]._check(rc, "mslam_allgather_i32")
    return recv


def loaded_runtime_libs() -> dict:
    """Which libamdhip64 / librccl files this process has mapped (two HIP runtimes in one process do not share memory)."""
    out = {"libamdhip64": [], "librccl": []}
    try:
        for line in open("/proc/self/maps"):
            if "r-xp" not in line:
                continue
            path = line.split()[-1]
            for k in out:
                if k in path and path not in out[k]:
                    out[k].append(path)
    except OSError:
        pass
    return out


def ratio_filter(idx: np.ndarray, dist: np.ndarray, ratio: float):
    nq = idx.shape[0]
    qi = np.zeros(nq, np.int32); ti = np.zeros(nq, np.int32); d = np.zeros(nq, np.float32)
    n = C.c_int(0)
    rc = lib().mslam_ratio_filter(_p(np.ascontiguousarray(idx, np.int32), i32p), _p(np.ascontiguousarray(dist, np.int32), i32p),
                                  nq, C.c_float(ratio), _p(qi, i32p), _p(ti, i32p), _p(d, f32p), C.byref(n))
    if rc != OK:
        raise MslamError(rc, "mslam_ratio_filter")
    return qi[: n.value].copy(), ti[: n.value].copy(), d[: n.value].copy()
