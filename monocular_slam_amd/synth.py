"""Seeded synthetic inputs for the hot path.

There is no dataset on the GPU box, so every test and bench input is generated here.
All generators are pure numpy, seeded with PCG64, and cheap enough for 1920x1080.
"""
from __future__ import annotations

import numpy as np

# camera intrinsics the reference hard-codes (src/FrameLoader.cpp:173-176, :196-199)
K_TUM1 = np.array([[517.306408, 0.0, 318.643040], [0.0, 516.469215, 255.313989], [0.0, 0.0, 1.0]])
K_MINITAUR = np.array([[1060.0, 0.0, 913.5], [0.0, 1060.0, 550.2], [0.0, 0.0, 1.0]])
# TUM freiburg1 distortion the reference hard-codes in pnpPoseEstimation (src/CameraPoseEstimator.cpp:462-469)
DIST_TUM1 = np.array([0.2624, -0.9531, -0.0054, 0.0026, 1.1633, 0.0, 0.0, 0.0])


def textured_frame(w: int, h: int, seed: int, n_blobs: int | None = None, n_rects: int | None = None) -> np.ndarray:
    """Corner-rich u8 gray frame: Gaussian blobs + axis-aligned rectangles on mid-gray."""
    rng = np.random.Generator(np.random.PCG64(seed))
    area = w * h
    if n_blobs is None:
        n_blobs = max(40, int(400 * area / (640 * 480)))
    if n_rects is None:
        n_rects = max(20, int(200 * area / (640 * 480)))
    img = np.full((h, w), 128.0, dtype=np.float32)
    # rectangles first (sharp corners), then blobs on top (smooth texture)
    for _ in range(n_rects):
        rw, rh = rng.integers(6, 48, size=2)
        x0 = int(rng.integers(0, max(1, w - rw)))
        y0 = int(rng.integers(0, max(1, h - rh)))
        img[y0:y0 + rh, x0:x0 + rw] = float(rng.integers(0, 256))
    for _ in range(n_blobs):
        s = float(rng.uniform(1.5, 6.0))
        r = int(3 * s) + 1
        cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
        amp = float(rng.uniform(-110, 110))
        x0, x1 = max(0, cx - r), min(w, cx + r + 1)
        y0, y1 = max(0, cy - r), min(h, cy + r + 1)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s)).astype(np.float32)
    img += rng.normal(0.0, 2.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def shifted_frame(img: np.ndarray, dx: int, dy: int, seed: int) -> np.ndarray:
    """The same scene translated by (dx, dy) pixels with fresh sensor noise (next frame of a stream)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.roll(np.roll(img, dy, axis=0), dx, axis=1).astype(np.float32)
    out += rng.normal(0.0, 1.5, size=out.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def descriptor_sets(n: int, seed: int, flip_p: float = 0.08, junk_frac: float = 0.2):
    """C1m: set A random 256-bit rows; set B = permuted A with bit flips, junk_frac of rows random."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    perm = rng.permutation(n)
    bits = np.unpackbits(a[perm], axis=1)
    flips = rng.random(bits.shape) < flip_p
    b = np.packbits(bits ^ flips.astype(np.uint8), axis=1)
    junk = rng.random(n) < junk_frac
    b[junk] = rng.integers(0, 256, size=(int(junk.sum()), 32), dtype=np.uint8)
    return a, b, perm


def rodrigues(r: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def distort(xn: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """OpenCV radial/tangential model k1,k2,p1,p2,k3 on normalised coordinates (n x 2)."""
    k1, k2, p1, p2, k3 = dist[:5]
    x, y = xn[:, 0], xn[:, 1]
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd, yd], axis=1)


def pnp_problem(P: int, seed: int, K: np.ndarray = K_MINITAUR, dist: np.ndarray | None = None,
                w: int = 1920, h: int = 1080, noise: float = 0.5, outlier_frac: float = 0.3):
    """C2: P world points in the frustum z in [2,10], small random pose, pixel noise, uniform outliers."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if dist is None:
        dist = np.zeros(8)
    rvec = rng.normal(0, 0.05, size=3)
    tvec = rng.normal(0, 0.2, size=3)
    R = rodrigues(rvec)
    # sample camera-frame points inside the image, then move to world
    z = rng.uniform(2.0, 10.0, size=P)
    u = rng.uniform(40, w - 40, size=P)
    v = rng.uniform(40, h - 40, size=P)
    xc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    X = (xc - tvec) @ R  # R^T (xc - t)
    xn = xc[:, :2] / xc[:, 2:3]
    xd = distort(xn, dist)
    x = np.stack([K[0, 0] * xd[:, 0] + K[0, 2], K[1, 1] * xd[:, 1] + K[1, 2]], axis=1)
    x += rng.normal(0, noise, size=x.shape)
    out = rng.random(P) < outlier_frac
    x[out] = np.stack([rng.uniform(0, w, size=int(out.sum())), rng.uniform(0, h, size=int(out.sum()))], axis=1)
    return dict(X=np.ascontiguousarray(X), x=np.ascontiguousarray(x), K=K.copy(), dist=dist.copy(),
                rvec=rvec, tvec=tvec, R=R, outlier=out)


def sample_table(n_items: int, H: int, s: int, seed: int) -> np.ndarray:
    """H rows of s distinct indices in [0, n_items) -- the explicit RANSAC sample table."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((H, s), dtype=np.int32)
    for hh in range(H):
        out[hh] = rng.choice(n_items, size=s, replace=False)
    return out


def two_view_problem(M: int, seed: int, K: np.ndarray = K_TUM1, w: int = 640, h: int = 480,
                     noise: float = 0.4, outlier_frac: float = 0.25):
    """Matched pixel pairs of a rigid scene seen from two poses (+ mismatches), with ground truth."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rvec = np.array([0.01, 0.05, -0.01]) + rng.normal(0, 0.005, size=3)
    t = np.array([1.0, 0.1, 0.05]); t /= np.linalg.norm(t)
    R = rodrigues(rvec)
    z = rng.uniform(3.0, 12.0, size=M)
    u = rng.uniform(20, w - 20, size=M)
    v = rng.uniform(20, h - 20, size=M)
    X1 = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    X2 = X1 @ R.T + t
    x1 = np.stack([u, v], axis=1)
    x2 = np.stack([K[0, 0] * X2[:, 0] / X2[:, 2] + K[0, 2], K[1, 1] * X2[:, 1] / X2[:, 2] + K[1, 2]], axis=1)
    x1 = x1 + rng.normal(0, noise, size=x1.shape)
    x2 = x2 + rng.normal(0, noise, size=x2.shape)
    out = rng.random(M) < outlier_frac
    x2[out] = np.stack([rng.uniform(0, w, size=int(out.sum())), rng.uniform(0, h, size=int(out.sum()))], axis=1)
    return dict(x1=np.ascontiguousarray(x1), x2=np.ascontiguousarray(x2), K=K.copy(), R=R, t=t, X=X1, outlier=out)


def quat_from_R(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x,y,z,w), w >= 0."""
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def ba_problem(n_poses: int, n_points: int, seed: int, K: np.ndarray = K_MINITAUR, w: int = 1920, h: int = 1080,
               obs_noise: float = 1.0, gross_frac: float = 0.05, pose_noise: float = 0.01, point_noise: float = 0.02):
    """C3: poses on an arc looking at a point cloud; edges where a point projects inside the image."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = np.stack([rng.uniform(-4, 4, n_points), rng.uniform(-2.5, 2.5, n_points), rng.uniform(6, 14, n_points)], axis=1)
    poses_gt = np.zeros((n_poses, 7))
    Rs, ts = [], []
    for j in range(n_poses):
        ang = (j - (n_poses - 1) / 2) * 0.04
        R = rodrigues(np.array([0.0, ang, 0.0]))
        c = np.array([8.0 * np.sin(ang) * 0.5, 0.0, 0.0])  # camera centre
        t = -R @ c
        Rs.append(R); ts.append(t)
        poses_gt[j, :3] = t
        poses_gt[j, 3:] = quat_from_R(R)
    scales = 31.0 * (np.float32(1.2) ** np.arange(8)).astype(np.float64)
    e_pose, e_point, obs, info = [], [], [], []
    for i in range(n_points):  # point-major, pose-minor: the order Util::BundleAdjustment creates edges in
        for j in range(n_poses):
            pc = Rs[j] @ pts[i] + ts[j]
            if pc[2] <= 0.5:
                continue
            u = K[0, 0] * pc[0] / pc[2] + K[0, 2]
            v = K[1, 1] * pc[1] / pc[2] + K[1, 2]
            if not (0 <= u < w and 0 <= v < h):
                continue
            o = np.array([u, v]) + rng.normal(0, obs_noise, 2)
            if rng.random() < gross_frac:
                o += rng.uniform(-50, 50, 2)
            # the reference truncates observations to float (src/Util.cpp:136) and uses 1.0f/scale (:151-153)
            o = o.astype(np.float32).astype(np.float64)
            sc = np.float32(scales[int(rng.integers(0, 8))])
            e_pose.append(j); e_point.append(i); obs.append(o); info.append(float(np.float32(1.0) / sc))
    poses = poses_gt.copy()
    for j in range(1, n_poses):
        dR = rodrigues(rng.normal(0, pose_noise, 3))
        R = dR @ Rs[j]
        poses[j, :3] = ts[j] + rng.normal(0, pose_noise, 3)
        poses[j, 3:] = quat_from_R(R)
    points = pts + rng.normal(0, point_noise, pts.shape)
    fixed = np.zeros(n_poses, dtype=np.uint8); fixed[0] = 1
    intr = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (n_poses, 1))
    return dict(n_poses=n_poses, n_points=n_points, n_edges=len(e_pose),
                poses=np.ascontiguousarray(poses), poses_gt=poses_gt, pose_fixed=fixed,
                points=np.ascontiguousarray(points), points_gt=pts,
                edge_pose=np.array(e_pose, dtype=np.int32), edge_point=np.array(e_point, dtype=np.int32),
                obs=np.ascontiguousarray(np.array(obs)), info=np.array(info), intr=np.ascontiguousarray(intr),
                huber_delta=float(np.float32(np.sqrt(5.99))))


def smooth_depth(w: int, h: int, base: float = 6.0) -> np.ndarray:
    """A smooth, clearly non-planar depth field for frame A (metres along the optical axis)."""
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    return base + 2.2 * np.sin(u / (0.11 * w)) * np.cos(v / (0.13 * h)) + 2.0 * (u / w) - 1.5 * (v / h)


def render_view(img_a: np.ndarray, depth_a: np.ndarray, K: np.ndarray, R: np.ndarray, t: np.ndarray, iters: int = 10) -> np.ndarray:
    """Frame B of the rigid scene (texture img_a at depth depth_a in camera A) seen from pose (R, t): x_B = R x_A + t.

    Backward warping: for every pixel of B the source pixel in A is found by fixed-point iteration on the forward
    projection (small motion), then sampled bilinearly."""
    h, w = img_a.shape
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    vb, ub = np.mgrid[0:h, 0:w].astype(np.float64)
    ua, va = ub.copy(), vb.copy()

    def bil(a, u, v):
        u = np.clip(u, 0, w - 1.001); v = np.clip(v, 0, h - 1.001)
        u0 = np.floor(u).astype(int); v0 = np.floor(v).astype(int)
        fu, fv = u - u0, v - v0
        return (a[v0, u0] * (1 - fu) * (1 - fv) + a[v0, u0 + 1] * fu * (1 - fv) + a[v0 + 1, u0] * (1 - fu) * fv + a[v0 + 1, u0 + 1] * fu * fv)

    for _ in range(iters):
        z = bil(depth_a, ua, va)
        xa = np.stack([(ua - cx) / fx * z, (va - cy) / fy * z, z], axis=-1)
        xb = xa @ R.T + t
        pu = fx * xb[..., 0] / xb[..., 2] + cx
        pv = fy * xb[..., 1] / xb[..., 2] + cy
        ua -= (pu - ub); va -= (pv - vb)
    out = bil(img_a.astype(np.float64), ua, va)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def rigid_sequence(w: int, h: int, seed: int, n_frames: int = 3, K: np.ndarray = K_TUM1):
    """C0: frame 0 = textured frame; frames 1.. = the same scene from small known motions. Returns frames, poses (R, t)."""
    a = textured_frame(w, h, seed)
    depth = smooth_depth(w, h)
    frames, poses = [a], [(np.eye(3), np.zeros(3))]
    for i in range(1, n_frames):
        R = rodrigues(np.array([0.012 * i, 0.03 * i, -0.02 * i]))
        t = np.array([0.22 * i, -0.10 * i, 0.16 * i])
        frames.append(render_view(a, depth, K, R, t))
        poses.append((R, t))
    return np.stack(frames), poses, depth


def ba_concat(problems):
    """Several independent BA problems as one block-diagonal problem (pose / point indices offset): what a batched
    local BA over many streams hands to one mslam_ba_plan."""
    out = {}
    po = pt = 0
    ep, el = [], []
    for q in problems:
        ep.append(q["edge_pose"] + po); el.append(q["edge_point"] + pt)
        po += q["n_poses"]; pt += q["n_points"]
    for k in ("poses", "pose_fixed", "points", "obs", "info", "intr"):
        out[k] = np.ascontiguousarray(np.concatenate([q[k] for q in problems]))
    for k in ("poses_gt", "points_gt"):
        if all(k in q for q in problems):
            out[k] = np.concatenate([q[k] for q in problems])
    out.update(edge_pose=np.concatenate(ep).astype(np.int32), edge_point=np.concatenate(el).astype(np.int32),
               n_poses=po, n_points=pt, n_edges=int(sum(q["n_edges"] for q in problems)), huber_delta=problems[0]["huber_delta"])
    return out
