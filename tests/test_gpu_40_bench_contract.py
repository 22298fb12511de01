"""bench.py's one-line JSON contract (driver-facing): run it as the driver does, at a small batch, and check every field the contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "4", *extra]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run("--no-cpu-baseline")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "images/sec" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "PATCH-16" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]          # whole-job images/s = batch / step time
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["achieved"] > 0 and r["launches"] > 0 and "traffic" in r and "tfc_igemm2_kernel" in r["kernel"]
    assert d["cpu_baseline"] is None                                                   # --no-cpu-baseline


def test_bench_glo16_and_fp32_modes_run():
    g = _run("--no-cpu-baseline", "--config", "glo16")
    assert "GLO-16" in g["config"]["workload"] and g["value"] > 0
    f = _run("--no-cpu-baseline", "--dtype", "fp32")
    assert f["dtype"] == "fp32" and f["roofline"]["peak"] == 157.3 and 0 < f["roofline"]["frac"] < 1
