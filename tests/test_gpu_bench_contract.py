"""The bench line keeps its contract: one JSON object with the agreed fields, the roofline and the
CPU baseline objects (run on a small matrix file so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_fields(golden_dir):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "mtx",
           "--matrix", os.path.join(golden_dir, "ref_first", "A.mtx")]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "GFLOP/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    assert d["config"]["nnz_c"] == 3067                      # matrices/first: the reference's C_expected.mtx
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    # the fraction is a bound (compulsory bytes / time / peak); the no-reuse traffic model sits beside it
    assert 0 < r["frac"] <= 1.0 and r["frac_model_no_reuse"] >= r["frac"]
    assert all(0 < f <= 1.0 for f in r["bounds"]["frac_of_each"].values())
    assert r["bounds"]["frac_of_tightest"] == max(r["bounds"]["frac_of_each"].values())
    # counters of a committed profile are used only beside the sources they were measured on
    assert (r["traffic"] is None) == (r["traffic_stale"] is not None)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "GFLOP/s" and c["sample"]
