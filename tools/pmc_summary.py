"""Summarise rocprofv3 --pmc passes: mean counter value per launch of the step kernel."""
import csv, glob, json, os, sys
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join("gpurun_out", f"pmc_{tag}")
out = {}
for path in sorted(glob.glob(os.path.join(root, "pass*", "pmc_counter_collection.csv"))):
    sums, counts = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            # the bench kernel: fused agent, several steps per launch (<6>) or one (<2>)
            if not any(k in name for k in ("step_kernel_pair<6, false", "step_kernel_pair<2, false", "step_kernel<2, false")):
                continue
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            counts[row["Counter_Name"]] += 1
    for k in sums:
        out[k] = {"mean_per_launch": sums[k] / counts[k], "launches": counts[k]}
print(json.dumps(out, indent=1))
