"""bench.py's one-line JSON contract (driver-facing): keys, types, and the two extra objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_device_or_with_wrong_world():
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr     # must be launched through torch.distributed.run


@pytest.mark.gpu
def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "3", "--no-extras",
                        "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    b = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(b[k], t), k
    assert b["vs_baseline"] is None and b["n_gpus"] == 1 and b["steps"] == 8 and b["warmup"] == 3
    assert b["scaling"] == "weak" and b["higher_is_better"] is True and b["data"] == "synthetic"
    assert "workload" in b["config"] and "model" not in b["config"]
    rf = b["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = b["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(b["value"] - b["config"]["features_active"] / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]
    assert b["px_err_vs_cpu"] == {"max": 0.0, "status_mismatches": 0}
