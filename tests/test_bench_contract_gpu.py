"""bench.py's output contract on the GPU box: exactly one line on stdout, a JSON object with the driver's keys, the roofline and
cpu_baseline objects, the BA leg and the parity check of the timed configuration -- at a reduced step count so that the test
stays within seconds (the workload per frame is the headline one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--seq-len", "4",
                        "--cpu-frames", "2", "--force-collective"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]   # native libraries (gloo, RCCL) print banners: they must not reach stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["vs_baseline"] is None
    assert d["valid_outputs"] is True and d["parity_checked_streams"] == 4 and not d["parity"]["mismatches"]
    assert abs(d["value"] - 64 * 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "fast_nms_k" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    ba = d["roofline_ba"]
    assert ba["kernel"] == "ba_edges_k" and 0.05 < ba["frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert d["collective"]["verified"] is True and d["collective"]["backend"].startswith("rccl")
