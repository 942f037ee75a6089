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
