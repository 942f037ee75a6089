#!/usr/bin/env python3
"""Rate of the coupled C++ pipeline (BatchedPipeline: extract -> match vs <= 5 previous frames -> F filter -> association ->
PnP RANSAC -> triangulation, map kept per stream) for B streams per launch, next to the single-stream ProcessingPipeline.
Rendered rigid sequences (640x480, 6 frames, 3 distinct scenes replicated)."""
import json, os, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from monocular_slam_amd import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 11
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = tempfile.mkdtemp()
paths = []
for i, seed in enumerate((1003, 2007, 3011)):
    frames = synth.rigid_sequence(640, 480, seed, n_frames=NF)[0]
    p = os.path.join(tmp, f"s{i}.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<4i", NF, 640, 480, 1)); f.write(np.ascontiguousarray(synth.K_TUM1, np.float64).tobytes()); f.write(frames.tobytes())
    paths.append(p)
host = os.path.join(ROOT, "monocular_slam_amd", "host")
flags = ["--no-dist", "--pnp-hyps", "256", "--pnp-thr", "3.0"] + sys.argv[3:]   # e.g. --fmat7: the 7-point F estimator
r = subprocess.run([os.path.join(host, "mslam_batched_demo"), os.path.join(tmp, "b.txt"), *paths, *flags, "--replicate", str(R), "--time"],
                   capture_output=True, text=True, timeout=900)
print("batched:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
t0 = time.perf_counter()
r1 = subprocess.run([os.path.join(host, "mslam_pipeline_demo"), paths[0], os.path.join(tmp, "o.txt"), *flags], capture_output=True, text=True, timeout=900)
dt = time.perf_counter() - t0
print(json.dumps({"single_stream_demo_wall_s": dt, "frames": NF, "note": "includes process start + context creation"}))
