"""Device-resident batched tracking front end: B independent streams per GPU.

One `step()` is one pass of the hot path over one batch of frames already resident in
HBM: ORB extract (K1-K6) -> Hamming kNN-2 + ratio test against the previous frame of the
same stream (K7) -> PnP RANSAC on that frame's 3D-2D correspondences (K8b).  It is the
batched equivalent of one ProcessingPipeline::process(dm, i) tick of the reference
(src/Pipeline.h:49-65) restricted to FeatureExtractor + CameraPoseEstimator (i >= 2 path,
src/CameraPoseEstimator.cpp:384-515), for B streams at once.  Everything is enqueued on
the context's stream through the C ABI; nothing here computes.
"""
from __future__ import annotations

import numpy as np

from . import api


class TrackingBatch:
    def __init__(self, ctx: api.Context, B: int, w: int, h: int, nfeatures: int, cap: int | None = None,
                 ratio: float = 0.8, pnp_points: int = 0, pnp_hyps: int = 0, pnp_ctx: api.Context | None = None):
        self.ctx, self.B, self.w, self.h = ctx, B, w, h
        # optional second context (= second HIP stream on the same GPU) for the PnP chain, so that it overlaps the
        # extract+match chain of the same step (software pipelining across stages; outputs are unchanged)
        self.pnp_ctx = pnp_ctx or ctx
        self.params = api.default_orb_params(nfeatures)
        self.cap = cap or min(ctx.max_kpts, nfeatures + 1024)
        self.ratio = ratio
        cap = self.cap
        D = api.DevBuf
        self.frames = D(ctx, B * w * h)
        # double-buffered per-stream features: cur / prev
        self.kps = [D(ctx, B * cap * 24), D(ctx, B * cap * 24)]
        self.desc = [D(ctx, B * cap * 32), D(ctx, B * cap * 32)]
        self.n = [D.from_array(ctx, np.zeros(B, np.int32)), D.from_array(ctx, np.zeros(B, np.int32))]
        self.status = D(ctx, B * 4)
        self.idx = D(ctx, B * cap * 2 * 4)
        self.dist = D(ctx, B * cap * 2 * 4)
        self.m_query, self.m_train, self.m_dist = D(ctx, B * cap * 4), D(ctx, B * cap * 4), D(ctx, B * cap * 4)
        self.m_count = D(ctx, B * 4)
        self.cur = 0
        self.pnp = None
        if pnp_points and pnp_hyps:
            self.pnp = dict(P=pnp_points, H=pnp_hyps)

    def upload_frames(self, frames: np.ndarray):
        assert frames.shape == (self.B, self.h, self.w) and frames.dtype == np.uint8
        self.frames.upload(frames)

    def set_pnp_problems(self, X, x, K, dist, samples, params):
        """B synthetic 3D-2D problems resident on the device (config C2 shape)."""
        B, P, H = self.B, self.pnp["P"], self.pnp["H"]
        D = api.DevBuf.from_array
        ctx = self.pnp_ctx
        self.pnp.update(
            X=D(ctx, X), x=D(ctx, x), K=D(ctx, K), dist=D(ctx, dist), samples=D(ctx, samples),
            Pn=D(ctx, np.full(B, P, np.int32)), params=params,
            Rt=api.DevBuf(ctx, B * 12 * 8), rvec=api.DevBuf(ctx, B * 3 * 8), tvec=api.DevBuf(ctx, B * 3 * 8),
            inl=api.DevBuf(ctx, B * P), counts=api.DevBuf(ctx, B * H * 4), best=api.DevBuf(ctx, B * 4),
            ninl=api.DevBuf(ctx, B * 4))

    def extract(self):
        c = self.cur
        self.ctx.orb_extract_batch_dev(self.frames.ptr, self.B, self.w, self.h, self.w, self.w * self.h, 1, self.params,
                                       self.kps[c].ptr, self.desc[c].ptr, self.cap, self.n[c].ptr, self.status.ptr)

    def match_prev(self):
        c, p = self.cur, 1 - self.cur
        self.ctx.match_knn2_batch_dev(self.desc[c].ptr, self.n[c].ptr, self.desc[p].ptr, self.n[p].ptr, self.cap, self.B,
                                      self.idx.ptr, self.dist.ptr, self.ratio, self.m_query.ptr, self.m_train.ptr,
                                      self.m_dist.ptr, self.m_count.ptr)

    def pnp_ransac(self):
        q = self.pnp
        self.pnp_ctx.pnp_ransac_batch_dev(q["X"].ptr, q["x"].ptr, q["Pn"].ptr, q["P"], q["K"].ptr, q["dist"].ptr,
                                      q["samples"].ptr, q["H"], self.B, q["params"], q["Rt"].ptr, q["rvec"].ptr,
                                      q["tvec"].ptr, q["inl"].ptr, q["counts"].ptr, q["best"].ptr, q["ninl"].ptr)

    def step(self):
        """extract + match(prev) [+ PnP]; then the current features become the previous ones."""
        self.extract()
        self.match_prev()
        if self.pnp is not None and "X" in self.pnp:
            self.pnp_ransac()
        self.cur = 1 - self.cur

    def sync(self):
        self.ctx.sync()
        if self.pnp_ctx is not self.ctx:
            self.pnp_ctx.sync()

    def results(self):
        """Host copies of the last step's outputs (syncs)."""
        self.sync()
        p = 1 - self.cur  # buffers written by the last step
        B, cap = self.B, self.cap
        out = dict(n=self.n[p].download(np.int32, (B,)), status=self.status.download(np.int32, (B,)),
                   kps=self.kps[p].download(api.KP_DTYPE, (B, cap)), desc=self.desc[p].download(np.uint8, (B, cap, 32)),
                   idx=self.idx.download(np.int32, (B, cap, 2)), dist=self.dist.download(np.int32, (B, cap, 2)),
                   m_count=self.m_count.download(np.int32, (B,)), m_query=self.m_query.download(np.int32, (B, cap)),
                   m_train=self.m_train.download(np.int32, (B, cap)))
        if self.pnp is not None and "X" in self.pnp:
            q = self.pnp
            out.update(Rt=q["Rt"].download(np.float64, (B, 3, 4)), best=q["best"].download(np.int32, (B,)),
                       ninl=q["ninl"].download(np.int32, (B,)), counts=q["counts"].download(np.int32, (B, q["H"])),
                       inl=q["inl"].download(np.uint8, (B, q["P"])))
        return out
