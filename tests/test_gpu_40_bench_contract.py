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


def test_bench_stn21_config_and_one_rank_rccl_mode_run():
    """`--config stn21` (BASELINE.json configs[4] on the same line format) and the one-GPU rehearsal of the RCCL path (TFC_FORCE_COLLECTIVES=1: a one-rank
    "nccl" group created by bench.py itself; the collectives are issued and `exposed_allreduce_ms` is measured)"""
    s = _run("--no-cpu-baseline", "--config", "stn21", "--batch", "2")
    assert "STN21" in s["config"]["workload"] and s["value"] > 0 and s["cpu_baseline"] is None and s["roofline"]["launches"] > 0
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"]
    env = dict(os.environ, TFC_FORCE_COLLECTIVES="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    if out.returncode == 77:
        pytest.skip("a one-rank RCCL process group cannot be created on this box (bench.py exit code 77)")
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["allreduce_backend"] == "nccl" and d["exposed_allreduce_ms"] > 0.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (no outer launcher) must start its N ranks itself and print ONE line with n_gpus = N. Rehearsed on
    the one card with gloo (TFC_BENCH_BACKEND=gloo: both ranks share cuda:0); on a multi-GPU node the same command runs over RCCL."""
    env = dict(os.environ, TFC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["allreduce_backend"] == "gloo" and d["exposed_allreduce_ms"] >= 0.0
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_bench_failing_rank_gives_nonzero_exit():
    env = dict(os.environ, TFC_BENCH_BACKEND="no-such-backend")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
