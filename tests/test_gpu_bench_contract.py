"""bench.py prints ONE JSON line with the driver's contract fields plus `roofline` and `cpu_baseline`
(small workload here; the full-size line is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--rows", "300000", "--nq", "1024", "--n-to-score", "3000", "--cpu-traversals", "16"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    j = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[k], typ), (k, j[k])
    assert j["vs_baseline"] is None and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["data"] == "synthetic" and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0 and "traffic" in r
    assert r["kernel"] in ("trav_kernel", "trav4_kernel") and r["launches"] == 2 and r["avg_launch_ms"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
