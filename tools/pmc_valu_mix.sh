#!/bin/bash
# Dynamic instruction mix of the bench kernel's VALU stream (per launch of step_kernel_octet<2,false>, 512 wavefronts):
# fp32 fma / mul / add / transcendental, integer, conversions against all VALU instructions. --kernel-trace only.
# Usage (GPU box): bash tools/pmc_valu_mix.sh <tag>
set -u
TAG=${1:-valumix}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
  "SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SENDMSG SQ_INSTS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
    python $R/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-fused --no-steady-state --no-secondary > $OUT/pass$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "step_kernel_octet" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]
    print(f"{k:28s} per wavefront and launch {sum(v)/len(v)/512:10.1f}")
PY
