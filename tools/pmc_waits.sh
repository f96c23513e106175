#!/bin/bash
# What a wavefront of the bench kernel waits for: average latency of instruction fetches, scalar loads and vector
# loads (rocprofv3's derived counters: in-flight level integrated over time / requests), branch and scalar-unit time.
# One counter group per pass, --kernel-trace only. Usage (GPU box): bash tools/pmc_waits.sh <tag> [extra bench.py flags]
set -u
TAG=${1:-waits}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "InstrFetchLatency" "SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SmemLatency" \
  "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "VmemLatency" "SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
    python $R/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-fused --no-steady-state --no-secondary "$@" > $OUT/pass$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "step_kernel_octet" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]
    print(f"{k:24s} per launch {sum(v)/len(v):14.1f}   (n={len(v)})")
PY
