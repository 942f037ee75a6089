"""Config C0 and beyond: the re-stated ProcessingPipeline (C++ stages over the C ABI, monocular_slam_amd/host) on a
synthetic rigid sequence, against (a) ground truth and (b) the same control flow run over the oracle primitives."""
import os
import struct
import subprocess

import numpy as np
import pytest

from monocular_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_pipeline_demo")


def _run_demo(tmp_path, frames, K, *flags):
    n, h, w = frames.shape
    inp, out = tmp_path / "seq.bin", tmp_path / "out.txt"
    with open(inp, "wb") as f:
        f.write(struct.pack("<4i", n, w, h, 1))
        f.write(np.ascontiguousarray(K, np.float64).tobytes())
        f.write(np.ascontiguousarray(frames).tobytes())
    r = subprocess.run([DEMO, str(inp), str(out), *flags], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    poses, nfeat, pts, chi = [], [], [], None
    for line in open(out):
        tok = line.split()
        if tok[0] == "frame":
            nfeat.append(int(tok[2])); poses.append(np.array(tok[3:], float).reshape(3, 4))
        elif tok[0] == "mappoints":
            pass
        elif tok[0] == "ba_chi2":
            chi = np.array(tok[1:], float)
        else:
            pts.append([float(v) for v in tok[1:4]] + [int(tok[4])])
    return poses, nfeat, np.array(pts), chi


def _rot_angle(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


@pytest.fixture(scope="module")
def sequence():
    frames, poses, depth = synth.rigid_sequence(640, 480, 1003, n_frames=4)
    return frames, poses


def test_c0_two_frame_initialisation(tmp_path, sequence, oracle):
    """C0: two 640x480 frames through FeatureExtractor -> CameraPoseEstimator; R, t (up to scale) vs truth and oracle flow.
    The thresholds against ground truth are a sanity check for the stages' default table seed (Stages.h): with the
    reference's RANSAC confidence of 0.85 the initialisation varies a lot from table to table (translation direction 1-27
    degrees off over seeds 1..15 in the oracle flow).  Parity with the oracle-orchestrated flow is the test proper."""
    frames, gt = sequence
    poses, nfeat, pts, _ = _run_demo(tmp_path, frames[:2], synth.K_TUM1, "--no-dist")
    assert nfeat == [500, 500]
    np.testing.assert_array_equal(poses[0], np.eye(3, 4))
    R, t = gt[1]
    assert _rot_angle(poses[1][:, :3], R) < 0.5
    td = poses[1][:, 3] / np.linalg.norm(poses[1][:, 3])
    assert np.degrees(np.arccos(np.clip(td @ (t / np.linalg.norm(t)), -1, 1))) < 5.0
    assert len(pts) > 100 and (pts[:, 2] > 0).mean() > 0.95 and (pts[:, 3] == 2).all()
    # parity with the oracle-orchestrated flow: same poses, same map
    import ref_pipeline
    F, mp, obs = ref_pipeline.run(oracle, frames[:2], synth.K_TUM1)
    np.testing.assert_allclose(poses[1], F[1]["Rt"], rtol=1e-6, atol=1e-9)
    assert len(pts) == len(mp)
    np.testing.assert_allclose(pts[:, :3], mp, rtol=1e-6, atol=1e-8)


def test_tracking_with_pnp_and_map_growth(tmp_path, sequence, oracle):
    """Frames >= 2: match against up to 5 previous frames, F filter, PnP RANSAC, triangulation of new points."""
    frames, gt = sequence
    poses, nfeat, pts, _ = _run_demo(tmp_path, frames, synth.K_TUM1, "--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0")
    s = np.linalg.norm(gt[1][1]) / np.linalg.norm(poses[1][:, 3])   # monocular scale from the first baseline
    for i in (2, 3):
        R, t = gt[i]
        assert _rot_angle(poses[i][:, :3], R) < 1.0, i
        assert np.linalg.norm(poses[i][:, 3] * s - t) < 0.15 * np.linalg.norm(t), i
    import ref_pipeline
    F, mp, obs = ref_pipeline.run(oracle, frames, synth.K_TUM1, pnp_hyps=256, pnp_thr=3.0)
    for i in range(4):
        np.testing.assert_allclose(poses[i], F[i]["Rt"], rtol=1e-6, atol=1e-8)
    assert len(pts) == len(mp) and (pts[:, 3] == [len(o) for o in obs]).all()
    np.testing.assert_allclose(pts[:, :3], mp, rtol=1e-5, atol=1e-7)


def test_optimiser_stage_reduces_robust_chi2(tmp_path, sequence):
    """Optimiser::localBA after every frame (full BA, LM on the host over GPU linearisations): chi2 decreases, frame 0 fixed."""
    frames, gt = sequence
    poses, nfeat, pts, chi = _run_demo(tmp_path, frames[:3], synth.K_TUM1, "--no-dist", "--ba", "--pnp-hyps", "256", "--pnp-thr", "3.0")
    assert chi is not None and len(chi) >= 2 and (np.diff(chi) <= 1e-9 * chi[0]).all()
    np.testing.assert_allclose(poses[0], np.eye(3, 4), atol=1e-12)
    assert _rot_angle(poses[2][:, :3], gt[2][0]) < 1.5      # sanity against truth (table-seed dependent, see test_c0); 1.02 for the default seed


def test_local_ba_window_keeps_older_frames_fixed(tmp_path, sequence):
    """Optimiser::localBA(data, n, window): window = 1 frees only the newest posed frame, so the pose frame 1 got while it was
    the newest one survives every later BA bit for bit; with window <= 0 (the reference's global BA, src/Optimiser.cpp:6-18)
    later BAs move it.  chi2 decreases either way and frame 0 stays the gauge."""
    frames, gt = sequence
    flags = ("--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0")
    p2, _, _, _ = _run_demo(tmp_path, frames[:2], synth.K_TUM1, *flags, "--ba-window", "1")
    p4, _, pts, chi = _run_demo(tmp_path, frames, synth.K_TUM1, *flags, "--ba-window", "1")
    np.testing.assert_array_equal(p4[0], np.eye(3, 4))
    np.testing.assert_array_equal(p4[1], p2[1])
    assert chi is not None and len(chi) >= 2 and (np.diff(chi) <= 1e-9 * chi[0]).all()
    g2, _, _, _ = _run_demo(tmp_path, frames[:2], synth.K_TUM1, *flags, "--ba")
    g4, _, _, _ = _run_demo(tmp_path, frames, synth.K_TUM1, *flags, "--ba")
    np.testing.assert_array_equal(g2[1], p2[1])          # with two frames the window and the global problem coincide
    assert np.abs(g4[1] - g2[1]).max() > 1e-9            # global BA keeps refining frame 1
    for i in (2, 3):
        assert _rot_angle(p4[i][:, :3], gt[i][0]) < 1.0
    # the sequence demo exposes the same switch (src/main.cpp:43's commented-out stage, parameterised)
    r = subprocess.run([os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_sequence_demo")], capture_output=True, text=True)
    assert "--ba-window" in r.stderr


def test_frame_loader_sequence_matches_raw_pipeline(tmp_path, sequence):
    """f2: FrameLoader (TUM-style tree: rgb/<timestamp>.png, groundtruth.txt) -> ProcessingPipeline gives what the raw-frame
    demo gives on the same pixels and intrinsics; ground truth is attached to the frames as the reference does."""
    from test_frameloader_cpu import write_png
    frames, gt = sequence
    root = tmp_path / "synthetic_f1_sequence"
    (root / "rgb").mkdir(parents=True)
    stamps = [1305031102.175304 + 0.0333 * i for i in range(len(frames))]
    for i in np.random.default_rng(0).permutation(len(frames)):          # directory order must not matter
        write_png(root / "rgb" / f"{stamps[i]:.6f}.png", frames[i].astype(np.uint16))
    recs = []
    rng = np.random.default_rng(3)
    for i, ts in enumerate(stamps):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        recs.append((ts - 0.004, *rng.uniform(-1, 1, 3), *q))
    with open(root / "groundtruth.txt", "w") as f:
        f.write("# ground truth trajectory\n# file: synthetic\n# timestamp tx ty tz qx qy qz qw\n")
        for r in recs:
            f.write(" ".join(f"{v:.9f}" if k else f"{v:.6f}" for k, v in enumerate(r)) + "\n")
    out = tmp_path / "seq_out.txt"
    demo = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_sequence_demo")
    r = subprocess.run([demo, str(root), "0", "100", "1", str(out), "--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    poses, gts = [], []
    for line in open(out):
        tok = line.split()
        if tok[0] == "frame":
            poses.append(np.array(tok[3:], float).reshape(3, 4))
        elif tok[0] == "gt":
            gts.append(np.array(tok[2:], float).reshape(3, 4))
    K = np.array([[517.3, 0, 318.6], [0, 516.5, 255.3], [0, 0, 1.0]])   # loadCameraIntrinsics_kinect
    ref, _, _, _ = _run_demo(tmp_path, frames, K, "--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0")
    assert len(poses) == len(frames) == len(gts)
    for a, b in zip(poses, ref):
        np.testing.assert_array_equal(a, b)
    for g, rec in zip(gts, recs):
        tx, ty, tz, qx, qy, qz, qw = [float(f"{v:.9f}") for v in rec[1:]]
        R = np.array([[1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw],
                      [2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw],
                      [2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy]])
        np.testing.assert_allclose(g, np.c_[R, -R @ np.array([tx, ty, tz])], rtol=1e-12, atol=1e-12)  # R kept, t' = -R t (as written)


def _write_seq(path, frames, K):
    n, h, w = frames.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", n, w, h, 1))
        f.write(np.ascontiguousarray(K, np.float64).tobytes())
        f.write(np.ascontiguousarray(frames).tobytes())


def test_batched_pipeline_equals_single_stream_pipelines(tmp_path, sequence):
    """BatchedPipeline (B sequences per launch, features resident on the device between the stages) gives every stream what
    the single-stream ProcessingPipeline gives it alone: identical feature counts, poses and map, bit for bit."""
    frames, gt = sequence
    seqs = [frames, synth.rigid_sequence(640, 480, 2007, n_frames=4)[0], frames[:, ::-1, :].copy()]
    flags = ("--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0")
    paths = []
    for i, fr in enumerate(seqs):
        p = tmp_path / f"s{i}.bin"
        _write_seq(p, fr, synth.K_TUM1)
        paths.append(str(p))
    out = tmp_path / "batched.txt"
    demo = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_batched_demo")
    r = subprocess.run([demo, str(out), *paths, *flags, "--check-descriptors"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = []
    for line in open(out):
        tok = line.split()
        if tok[0] == "stream":
            got.append(dict(poses=[], nfeat=[], pts=[], desc=None))
        elif tok[0] == "frame":
            got[-1]["nfeat"].append(int(tok[2])); got[-1]["poses"].append(np.array(tok[3:], float).reshape(3, 4))
        elif tok[0] == "descsum":
            got[-1]["desc"] = (int(tok[1]), int(tok[2]))
        elif tok[0] != "mappoints":
            got[-1]["pts"].append([float(v) for v in tok[1:4]] + [int(tok[4])])
    assert len(got) == 3
    for i, fr in enumerate(seqs):
        poses, nfeat, pts, _ = _run_demo(tmp_path, fr, synth.K_TUM1, *flags)
        assert got[i]["nfeat"] == nfeat
        for a, b in zip(got[i]["poses"], poses):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.array(got[i]["pts"]), pts)
        assert got[i]["desc"][0] == nfeat[-1] and got[i]["desc"][1] != 0
    # the streams really are different problems
    assert not np.array_equal(got[0]["poses"][2], got[1]["poses"][2])


def test_pipeline_with_the_seven_point_estimator_of_the_live_reference_call(tmp_path, sequence, oracle):
    """CameraPoseEstimator::fmatSamplePoints = 7 (--fmat7): findFundamentalMat(CV_FM_RANSAC) as OpenCV runs it
    (src/CameraPoseEstimator.cpp:563) -- single-stream stages against the oracle-orchestrated flow, and the batched
    pipeline against the single-stream one."""
    frames, gt = sequence
    flags = ("--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0", "--fmat7")
    poses, nfeat, pts, _ = _run_demo(tmp_path, frames, synth.K_TUM1, *flags)
    import ref_pipeline
    F, mp, obs = ref_pipeline.run(oracle, frames, synth.K_TUM1, pnp_hyps=256, pnp_thr=3.0, fmat_points=7)
    for i in range(4):
        np.testing.assert_allclose(poses[i], F[i]["Rt"], rtol=1e-6, atol=1e-8)
    assert len(pts) == len(mp) and len(pts) > 100 and (pts[:, 3] == [len(o) for o in obs]).all()
    np.testing.assert_allclose(pts[:, :3], mp, rtol=1e-5, atol=1e-7)
    poses8, _, _, _ = _run_demo(tmp_path, frames, synth.K_TUM1, *flags[:-1])
    assert not np.array_equal(poses8[1], poses[1])          # a different estimator, a different two-view initialisation
    assert _rot_angle(poses[1][:, :3], gt[1][0]) < 1.0
    p = tmp_path / "s7.bin"
    _write_seq(p, frames, synth.K_TUM1)
    out = tmp_path / "batched7.txt"
    demo = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_batched_demo")
    r = subprocess.run([demo, str(out), str(p), str(p), *flags], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [np.array(line.split()[3:], float).reshape(3, 4) for line in open(out) if line.startswith("frame")]
    assert len(got) == 8
    for i in range(4):
        np.testing.assert_array_equal(got[i], poses[i])
        np.testing.assert_array_equal(got[4 + i], poses[i])


def test_cpp_bench_binary_runs_the_batched_path():
    """mslam_bench (C++ host, MultiStreamRunner over the C ABI) at a small configuration: valid outputs, one JSON line."""
    import json
    exe = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_bench")
    r = subprocess.run([exe, "--steps", "2", "--warmup", "1", "--batch", "4", "--sub-batches", "2", "--seq-len", "3", "--w", "640", "--h", "480",
                        "--nfeatures", "500", "--pnp-points", "400", "--pnp-hyps", "256"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["valid_outputs"] is True and d["value"] > 0 and d["mean_keypoints"] > 400 and d["mean_ratio_survivors"] > 60


def test_device_twin_is_not_trusted_after_an_in_place_edit(tmp_path, sequence):
    """Features::descriptors keeps a device twin so that matching does not upload it again; cv::Mat hands out mutable
    access, so an in-place edit of the host rows must make matchFeatures fall back to them (content hash in the twin)."""
    frames, _ = sequence
    n, h, w = frames[:2].shape
    inp, out = tmp_path / "seq.bin", tmp_path / "out.txt"
    with open(inp, "wb") as f:
        f.write(struct.pack("<4i", n, w, h, 1))
        f.write(np.ascontiguousarray(synth.K_TUM1, np.float64).tobytes())
        f.write(np.ascontiguousarray(frames[:2]).tobytes())
    r = subprocess.run([DEMO, str(inp), str(out), "--twin-check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + open(out).read() if os.path.exists(out) else r.stderr
    tok = open(out).read().split()
    assert tok[:2] == ["twin_check", "ok"] and int(tok[2]) > 50
