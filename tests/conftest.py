import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def ctx():
    """One context for the whole GPU session (1080p, 4096 keypoints, batch 8)."""
    from monocular_slam_amd import api
    c = api.Context(device=0, max_w=1920, max_h=1080, max_kpts=4096, batch=8)
    yield c
    c.close()


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts(request):
    """The in-tree build normally travels with the snapshot; if a piece is missing (fresh checkout), build it once.
    Compiling is not computing: the HIP library is still the only thing the GPU tests call."""
    import subprocess
    lib = os.path.join(ROOT, "monocular_slam_amd", "libmslam_hip.so")
    host = os.path.join(ROOT, "monocular_slam_amd", "host")
    demos = [os.path.join(host, n) for n in ("libmslam_host.so", "mslam_bench", "mslam_batched_demo", "mslam_pipeline_demo", "mslam_ba_csv_demo", "mslam_sequence_demo", "mslam_imageio_check", "mslam_geometry_check", "mslam_util_check")]
    try:
        if not os.path.exists(lib):
            subprocess.run(["make", "-C", os.path.join(ROOT, "monocular_slam_amd", "csrc"), "-j4"], check=True, capture_output=True)
        if not all(os.path.exists(d) for d in demos):
            subprocess.run(["make", "-C", host], check=True, capture_output=True)
    except (subprocess.CalledProcessError, FileNotFoundError) as e:  # tests that need the piece will say so themselves
        print("build on demand failed:", e)
