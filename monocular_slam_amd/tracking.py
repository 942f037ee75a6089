"""Device-resident batched tracking front end: B independent streams per GPU.

One `step()` is one pass of the hot path over one batch of frames already resident in
HBM: ORB extract (K1-K6) -> Hamming kNN-2 + ratio test against the previous frame of the
same stream (K7) -> PnP RANSAC on that frame's 3D-2D correspondences (K8b).  It is the
batched equivalent of one ProcessingPipeline::process(dm, i) tick of the reference
(src/Pipeline.h:49-65) restricted to FeatureExtractor + CameraPoseEstimator (i >= 2 path,
src/CameraPoseEstimator.cpp:384-515), for B streams at once.  Everything is enqueued on
the context's stream through the C ABI; nothing here computes.

Frames: two device-resident frame sets per stream that alternate step by step (frame k of a
stream is set k % 2), so "previous frame" (src/CameraPoseEstimator.cpp:405-409) is a different
image from the current one and the match is a real one.  With `enable_ingest` every step first
uploads its frame set from pinned host memory on a second (copy) context/stream, double-buffered
against the kernels of the step before (what a live camera feed would do; the reference loads
all frames from disk before its loop, src/main.cpp:35-37).
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
        self.frames = [D(ctx, B * w * h), None]     # frame sets; [1] allocated by upload_frames(a, b)
        self.fset = 0                               # frame set the next step reads
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
        self.steps_done = 0
        self.pnp = None
        if pnp_points and pnp_hyps:
            self.pnp = dict(P=pnp_points, H=pnp_hyps)
        self.ingest = None
        self.comm = None

    # ---- inputs
    def upload_frames(self, frames: np.ndarray, frames_b: np.ndarray | None = None):
        """frames: set 0 (B, h, w) u8; frames_b: set 1 (the steps alternate between them).  Without frames_b every step
        re-reads set 0 (current == previous frame: only for callers that do not look at the matches)."""
        assert frames.shape == (self.B, self.h, self.w) and frames.dtype == np.uint8
        self.frames[0].upload(frames)
        if frames_b is not None:
            assert frames_b.shape == frames.shape and frames_b.dtype == np.uint8
            if self.frames[1] is None:
                self.frames[1] = api.DevBuf(self.ctx, self.B * self.w * self.h)
            self.frames[1].upload(frames_b)
        self.fset = 0

    def enable_ingest(self, copy_ctx: api.Context, frames: np.ndarray, frames_b: np.ndarray):
        """Every step uploads its frame set from pinned host memory on `copy_ctx`'s stream before extracting it."""
        if self.frames[1] is None:
            self.frames[1] = api.DevBuf(self.ctx, self.B * self.w * self.h)
        host = [api.PinnedBuf(copy_ctx, frames.shape), api.PinnedBuf(copy_ctx, frames_b.shape)]
        host[0].array[...] = frames
        host[1].array[...] = frames_b
        self.ingest = dict(ctx=copy_ctx, host=host, up=[copy_ctx.event(), copy_ctx.event()],
                           read=[self.ctx.event(), self.ctx.event()], read_valid=[False, False])

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

    def set_comm(self, comm: api.Comm):
        """BASELINE configs[4]: after every PnP RANSAC the int32[H] inlier counts of this rank's stream 0 are all-gathered
        over RCCL on the PnP context's stream (no host sync); `gathered` then holds nranks x H counts on every rank."""
        assert self.pnp is not None and comm.ctx is self.pnp_ctx
        self.comm = comm
        self.gathered = api.DevBuf(self.pnp_ctx, comm.nranks * self.pnp["H"] * 4)

    # ---- stages
    def extract(self):
        c, f = self.cur, self.fset if self.frames[1] is not None else 0
        g = self.ingest
        if g is not None:
            cc = g["ctx"]
            if g["read_valid"][f]:
                cc.wait_event(g["read"][f])     # the extract that last read this buffer must be done before it is overwritten
            cc.memcpy_h2d_async(self.frames[f].ptr, g["host"][f].ptr, self.B * self.w * self.h)
            cc.record(g["up"][f])
            self.ctx.wait_event(g["up"][f])
        self.ctx.orb_extract_batch_dev(self.frames[f].ptr, self.B, self.w, self.h, self.w, self.w * self.h, 1, self.params,
                                       self.kps[c].ptr, self.desc[c].ptr, self.cap, self.n[c].ptr, self.status.ptr)
        if g is not None:
            self.ctx.record(g["read"][f])
            g["read_valid"][f] = True

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
        if self.comm is not None:
            self.comm.allgather_i32_dev(q["counts"].ptr, q["H"], self.gathered.ptr)

    def step(self):
        """extract + match(prev) [+ PnP]; then the current features become the previous ones."""
        self.extract()
        self.match_prev()
        if self.pnp is not None and "X" in self.pnp:
            self.pnp_ransac()
        self.advance()

    def advance(self):
        self.cur = 1 - self.cur
        self.fset = 1 - self.fset
        self.steps_done += 1

    def sync(self):
        self.ctx.sync()
        if self.pnp_ctx is not self.ctx:
            self.pnp_ctx.sync()
        if self.ingest is not None:
            self.ingest["ctx"].sync()

    def last_frame_sets(self):
        """(frame set extracted by the last step, frame set of the step before it)."""
        if self.frames[1] is None:
            return 0, 0
        return 1 - self.fset, self.fset

    def results(self):
        """Host copies of the last step's outputs (syncs)."""
        self.sync()
        p = 1 - self.cur  # buffers written by the last step
        B, cap = self.B, self.cap
        out = dict(n=self.n[p].download(np.int32, (B,)), status=self.status.download(np.int32, (B,)),
                   kps=self.kps[p].download(api.KP_DTYPE, (B, cap)), desc=self.desc[p].download(np.uint8, (B, cap, 32)),
                   n_prev=self.n[1 - p].download(np.int32, (B,)), desc_prev=self.desc[1 - p].download(np.uint8, (B, cap, 32)),
                   idx=self.idx.download(np.int32, (B, cap, 2)), dist=self.dist.download(np.int32, (B, cap, 2)),
                   m_count=self.m_count.download(np.int32, (B,)), m_query=self.m_query.download(np.int32, (B, cap)),
                   m_train=self.m_train.download(np.int32, (B, cap)), m_dist=self.m_dist.download(np.float32, (B, cap)))
        if self.pnp is not None and "X" in self.pnp:
            q = self.pnp
            out.update(Rt=q["Rt"].download(np.float64, (B, 3, 4)), best=q["best"].download(np.int32, (B,)),
                       ninl=q["ninl"].download(np.int32, (B,)), counts=q["counts"].download(np.int32, (B, q["H"])),
                       inl=q["inl"].download(np.uint8, (B, q["P"])))
            if self.comm is not None:
                out["gathered"] = self.gathered.download(np.int32, (self.comm.nranks, q["H"]))
        return out
