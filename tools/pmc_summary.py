"""Summarise the rocprofv3 --pmc passes of tools/pmc_pass.sh: mean counter value
per launch of the dominant step kernel, in the shape bench.py reads
(profiles/pmc_step_b4096.json: launch shape, counters, HBM bytes per launch).

    python tools/pmc_summary.py <tag> [--envs 4096] [--steps-per-launch 1] [--kernel step_kernel] [--out profiles/pmc_step_b4096.json]

HBM bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests of wide coalesced streaming reads at 64 bytes
(MI355X_MICROARCH.md, HBM section) -- this kernel's reads are 4-byte-per-lane
SoA words (one 256-byte run per wave and state word), not 16-byte-per-lane
streams, so the figure is kept uncorrected and the corrected one (x 2 on the
fetch side) is stored beside it as an upper bound."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

parser = argparse.ArgumentParser()
parser.add_argument("tag", nargs="?", default="r02")
parser.add_argument("--envs", type=int, default=4096)
parser.add_argument("--steps-per-launch", type=int, default=1)
parser.add_argument("--kernel", default="step_kernel", help="substring of the kernel name; the matching kernel with the most launches is summarised")
parser.add_argument("--out", default=None)
args = parser.parse_args()

root = os.path.join("gpurun_out", f"pmc_{args.tag}")
sums, counts, durations = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if args.kernel not in name:
                continue
            sums[name][row["Counter_Name"]] += float(row["Counter_Value"])
            counts[name][row["Counter_Name"]] += 1
for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*kernel_trace.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            if args.kernel in row["Kernel_Name"]:
                durations[row["Kernel_Name"]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
if not sums:
    raise SystemExit(f"no counters of a kernel matching {args.kernel!r} under {root}")
# the timed kernel = the one launched most often
kernel = max(sums, key=lambda n: max(counts[n].values()))
counters = {k: sums[kernel][k] / counts[kernel][k] for k in sorted(sums[kernel])}
out = {
    "kernel": kernel.split("(")[0],
    "launch_envs": args.envs,
    "steps_per_launch": args.steps_per_launch,
    "launches_counted": max(counts[kernel].values()),
    "counters": counters,
    "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_INSTS_* wave-level instructions",
}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    out["hbm_bytes_per_launch"] = round((counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024)
    out["hbm_bytes_per_launch_fetch_doubled"] = round((2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024)
if args.envs == 4096 and args.steps_per_launch == 1 and "octet" in kernel:  # (the Bullet-like kernel is built from the same sources)
    # the version of the kernel these counters belong to (bench.py refuses them for any other: tests/test_profiles_consistency.py)
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench

    out["kernel_fingerprint"] = bench.kernel_fingerprint()
if durations.get(kernel):
    d = durations[kernel]
    out["avg_launch_us"] = sum(d) / len(d) / 1e3  # under counter collection (slightly slower than an unprofiled launch)
text = json.dumps(out, indent=1)
print(text)
if args.out:
    with open(args.out, "w") as f:
        f.write(text + "\n")
