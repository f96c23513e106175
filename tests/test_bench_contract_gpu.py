"""bench.py's one-line JSON contract, on the GPU box."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    env = dict(os.environ, UPKIE_CPU_BASELINE_BUDGET_S="2")
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "10"],
                            capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-2000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "env-steps/s" and out["n_gpus"] == 1 and out["steps"] == 60 and out["warmup"] == 10
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] == pytest.approx(out["config"]["total_envs"] * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"]), rel=1e-6)
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and roof["achieved"] > 1.0
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu and cpu["unit"] == "env-steps/s"
    assert out["value"] > 100 * cpu["value"] / cpu["cores"]  # sanity: a GPU is not slower than a CPU core
