"""The evidence committed under profiles/ is what bench.py's roofline objects
quote: the files must exist, parse, and describe the launch shape bench.py runs."""

import csv
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def test_traffic_and_pmc_describe_the_default_launch():
    with open(os.path.join(P, "traffic.json")) as f:
        traffic = json.load(f)
    assert traffic["launch_envs"] == bench.ENVS_PER_GPU and traffic["steps_per_launch"] == bench.STEPS_PER_LAUNCH
    assert traffic["hbm_bytes_per_launch"] == round((traffic["fetch_kb"] + traffic["write_kb"]) * 1024)
    # below the algorithmic figure: the state is not re-read between the steps of a launch
    algorithmic = bench.ALGORITHMIC_BYTES_PER_ENV_STEP * traffic["launch_envs"] * traffic["steps_per_launch"]
    assert 0.05 * algorithmic < traffic["hbm_bytes_per_launch"] < algorithmic
    floor = bench.issue_floor(22.2, bench.STEPS_PER_LAUNCH)
    assert floor is not None and 0.9 < floor["frac"] < 1.02 and 1.0e4 < floor["instructions_per_wave_per_step"] < 1.4e4
    assert bench.issue_floor(26.2, 1) is None  # the committed counters are those of the 32-step launch


def test_kernel_stats_and_bench_line_agree():
    with open(os.path.join(P, "r01_bench_n1_final.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    assert line["metric"].startswith("env-steps/sec") and line["n_gpus"] == 1 and line["config"]["envs_per_gpu"] == bench.ENVS_PER_GPU
    per_step_bench = line["roofline"]["avg_step_us"]
    with open(os.path.join(P, "r01_kernel_stats_b4096_final.csv")) as f:
        rows = [r for r in csv.DictReader(f) if "step_kernel_pair<6, false, false>" in r["Name"] or "step_kernel_pair<2, false, false>" in r["Name"]]
    assert rows, "the dominant kernel is in the rocprofv3 summary"
    row = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    steps_profiled = 2200  # --steps 2000 --warmup 200
    per_step_rocprof = float(row["TotalDurationNs"]) / 1e3 / steps_profiled
    assert abs(per_step_rocprof - per_step_bench) < 0.03 * per_step_bench, (per_step_rocprof, per_step_bench)
    assert line["roofline"]["frac"] == line["roofline"]["achieved"] / line["roofline"]["peak"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
