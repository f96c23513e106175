"""The evidence committed under profiles/ is what bench.py's roofline objects
quote: the files must exist, parse, and describe the launch shape bench.py times."""

import csv
import glob
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
CURRENT_ROUND = "r06"  # the round whose bench line and rocprofv3 summary this tree claims (the test read round 3's files)


def test_pmc_file_describes_the_timed_launch_shape():
    """`roofline.traffic` and `roofline.valu` are only filled from counters
    collected on the launch shape bench.py times by default (4096 envs, ONE
    env.step() per launch); any other shape yields None instead of a number
    that belongs to a different launch."""
    assert os.path.exists(bench.PMC_FILE)
    with open(bench.PMC_FILE) as f:
        pmc = json.load(f)
    assert pmc["launch_envs"] == bench.ENVS_PER_GPU and pmc["steps_per_launch"] == 1
    # collected on THIS version of the headline kernel (sources without comments + compiler flags): a kernel change without
    # a fresh `bash tools/pmc_pass.sh rNN; python tools/pmc_summary.py rNN --out profiles/pmc_step_b4096.json` fails here
    assert pmc.get("kernel_fingerprint") == bench.kernel_fingerprint(), "profiles/pmc_step_b4096.json is older than the headline kernel"
    assert bench.pmc_of_launch_shape(bench.ENVS_PER_GPU, 1) is not None
    assert bench.pmc_of_launch_shape(bench.ENVS_PER_GPU, 32) is None and bench.pmc_of_launch_shape(8192, 1) is None
    c = pmc["counters"]
    assert pmc["hbm_bytes_per_launch"] == round((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    algorithmic = bench.ALGORITHMIC_BYTES_PER_ENV_STEP * pmc["launch_envs"]
    # no wasted re-reads (round 3: the workgroup -> env map is XCD aware, a 128-byte line of a state row lives in ONE L2;
    # round 2 fetched every line into four of them: 2.2 MB per launch against 1.06 MB algorithmic). What is left above the
    # algorithmic figure: the last torques and the contact flag the kernel stores for the lazy spine observation.
    assert 0.8 * algorithmic < pmc["hbm_bytes_per_launch"] < 1.25 * algorithmic
    valu = bench.valu_roofline(pmc, pmc["avg_launch_us"])
    assert 0.0 < valu["issue_utilisation"] < 1.0 and valu["tflops_upper_bound"] < bench.FP32_VALU_PEAK_TFLOPS
    assert valu["lone_wave_floor_us"] < 1.05 * pmc["avg_launch_us"]  # a launch cannot beat one wave's own instruction stream


def test_kernel_stats_and_bench_line_agree():
    """The committed bench line of this round and the rocprofv3 --stats summary
    of the same command: the dominant kernel's average duration agrees."""
    # the latest round that committed both
    rounds = sorted({os.path.basename(p)[:3] for p in glob.glob(os.path.join(P, "r[0-9][0-9]_kernel_stats_b4096*.csv"))})
    assert rounds and rounds[-1] == CURRENT_ROUND, f"no rocprofv3 summary of round {CURRENT_ROUND} under profiles/ (latest: {rounds[-1:]})"
    lines = sorted(glob.glob(os.path.join(P, f"{rounds[-1]}_bench_n1.json")))
    stats = sorted(glob.glob(os.path.join(P, f"{rounds[-1]}_kernel_stats_b4096*.csv")))
    assert lines and stats
    with open(lines[-1]) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    assert line["metric"].startswith("env-steps/sec") and line["n_gpus"] == 1 and line["config"]["envs_per_gpu"] == bench.ENVS_PER_GPU
    assert line["config"]["steps_per_launch"] == 1
    with open(stats[-1]) as f:
        rows = [r for r in csv.DictReader(f) if "step_kernel" in r["Name"]]
    assert rows, "the dominant kernel is in the rocprofv3 summary"
    row = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    per_launch_rocprof = float(row["AverageNs"]) / 1e3
    per_launch_bench = line["roofline"]["avg_launch_us"]
    # two runs of the same command (the tracer's own is a few percent slower: 18.97 vs 18.31 us in round 2); HIP events
    # bracket the launch gaps too
    assert per_launch_rocprof <= per_launch_bench * 1.06 and per_launch_rocprof > 0.6 * per_launch_bench, (per_launch_rocprof, per_launch_bench)
    assert line["roofline"]["frac"] == line["roofline"]["achieved"] / line["roofline"]["peak"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
