"""The reference's CameraPoseEstimator control flow (src/CameraPoseEstimator.cpp:264-530) orchestrated in Python
over the ORACLE primitives -- the checker for the C++ stage classes (which run the same flow over the C ABI).
Test infrastructure only."""
import numpy as np

M64 = (1 << 64) - 1


class Rng:
    """xorshift64* -- must equal geom::Rng (monocular_slam_amd/host/Geometry.h)."""

    def __init__(self, seed):
        self.s = seed if seed else 0x9E3779B97F4A7C15

    def next(self):
        s = self.s
        s ^= s >> 12
        s ^= (s << 25) & M64
        s ^= s >> 27
        self.s = s
        return (s * 0x2545F4914F6CDD1D) & M64

    def below(self, n):
        return (self.next() >> 32) % n


def splitmix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def sample_table(n, H, k, seed):
    """geom::sampleTable / mslam_sample_tables_dev: row h has its own generator, Rng(splitmix64(seed + (h+1) * golden))."""
    out = np.zeros((H, k), np.int32)
    if n < k:
        return out
    for h in range(H):
        rng = Rng(splitmix64(seed + (h + 1) * 0x9E3779B97F4A7C15))
        for i in range(k):
            while True:
                v = rng.below(n)
                if v not in out[h, :i]:
                    break
            out[h, i] = v
    return out


def match_features(ob, d1, d2, ratio):
    if len(d1) == 0 or len(d2) < 2:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    rc, idx, dist = ob.match_knn2(d1, d2)
    qi, ti, _ = ob.ratio_filter(idx, dist, ratio)
    return qi, ti


def fundamental(ob, p1, p2, qi, ti, seed, H=2000, k=8):
    fp = ob.fmat_default_params()   # metric 1, 3.0, 0.85, refit 2 == MAX_DISTANCE / CONFIDENCE / CV_FM_8POINT
    a, b = p1[qi], p2[ti]
    if len(qi) < k:
        return None, np.zeros(len(qi), np.uint8), a, b
    if k == 7:   # CameraPoseEstimator::fmatSamplePoints = 7: OpenCV's own estimator (7-point samples, LMedS below 15 matches)
        r = ob.fmat_ransac7(a, b, sample_table(len(qi), H, 7, seed), fp)
    else:
        r = ob.fmat_ransac8(a, b, sample_table(len(qi), H, 8, seed), fp)
    if r["rc"] != 0:
        return None, np.zeros(len(qi), np.uint8), a, b
    return r["F"], r["status"], a, b


def run(ob, frames, K, nfeatures=500, seed=5, pnp_hyps=100, pnp_thr=8.0, dist=None, fmat_points=8, pnp_stop=100):   # seed = CameraPoseEstimator::seed's default (Stages.h)
    """Returns per-frame dict(kps, desc, Rt, map_idx) and the map (positions, observers)."""
    p = ob.orb_default_params(nfeatures)
    dist = np.zeros(8) if dist is None else dist
    F = []
    mappts, observers = [], []
    for i, img in enumerate(frames):
        kps, desc = ob.orb_extract(img, p, cap=nfeatures + 1024)
        pos = np.stack([kps["x"].astype(np.float64), kps["y"].astype(np.float64)], axis=1)
        f = dict(kps=kps, desc=desc, pos=pos, map_idx=np.full(len(kps), -1, np.int64), Rt=None)
        F.append(f)
        if i == 0:
            f["Rt"] = np.eye(3, 4)
            continue
        if i == 1:
            pre = F[0]
            qi, ti = match_features(ob, pre["desc"], f["desc"], np.float32(0.85))
            Fm, status, a, b = fundamental(ob, pre["pos"], f["pos"], qi, ti, seed + 1000 * i, k=fmat_points)
            f["Rt"] = pre["Rt"].copy()
            if Fm is None:
                continue
            g1, g2 = a[status == 1], b[status == 1]
            best, Rt, X, cnt = ob.pose_from_F(Fm, K, K, g1, g2)
            f["Rt"] = np.c_[pre["Rt"][:, :3] @ Rt[:, :3], pre["Rt"][:, :3] @ Rt[:, 3]]   # concatenateRts as written
            c = 0
            for k in range(len(qi)):
                if status[k] == 1:
                    mid = len(mappts)
                    mappts.append(X[c]); observers.append({0: int(qi[k]), 1: int(ti[k])})
                    pre["map_idx"][qi[k]] = mid; f["map_idx"][ti[k]] = mid
                    c += 1
            continue
        matched = np.zeros(len(kps), bool)
        Xs, xs, cached = [], [], []
        for j in range(i - 1, max(-1, i - 6), -1):
            pre = F[j]
            qi, ti = match_features(ob, f["desc"], pre["desc"], np.float32(0.8))
            Fm, status, a, b = fundamental(ob, f["pos"], pre["pos"], qi, ti, seed + 1000 * i + (i - j), k=fmat_points)
            qi, ti = qi[status == 1], ti[status == 1]
            cached.append((j, qi, ti))
            for cf, pf in zip(qi, ti):
                if pre["map_idx"][pf] != -1 and not matched[cf]:
                    matched[cf] = True
                    mid = int(pre["map_idx"][pf])
                    f["map_idx"][cf] = mid; observers[mid][i] = int(cf)
                    Xs.append(mappts[mid]); xs.append(f["pos"][cf])
        f["Rt"] = F[i - 1]["Rt"].copy()
        if len(Xs) >= 4:
            pp = ob.pnp_default_params(); pp.reproj_thr = pnp_thr
            pp.stop_inliers = pnp_stop   # the stage's default: solvePnPRansac's minInliersCount (100)
            r = ob.pnp_ransac(np.array(Xs), np.array(xs), K, dist, sample_table(len(Xs), pnp_hyps, 4, seed + 1000 * i + 99), pp)
            if r["rc"] == 0:
                f["Rt"] = r["Rt"]
        for (j, qi, ti) in cached:
            pre = F[j]
            pairs = []
            for cf, pf in zip(qi, ti):
                if pre["map_idx"][pf] == -1 and f["map_idx"][cf] == -1 and all(pf != a_ and cf != b_ for a_, b_ in pairs):
                    pairs.append((pf, cf))
            if not pairs:
                continue
            x1 = np.array([pre["pos"][a_] for a_, _ in pairs]); x2 = np.array([f["pos"][b_] for _, b_ in pairs])
            X, _, _ = ob.triangulate(x1, x2, pre["Rt"], f["Rt"], K, K)
            for k, (pf, cf) in enumerate(pairs):
                mid = len(mappts)
                mappts.append(X[k]); observers.append({j: int(pf), i: int(cf)})
                pre["map_idx"][pf] = mid; f["map_idx"][cf] = mid
    return F, np.array(mappts), observers
