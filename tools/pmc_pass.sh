#!/bin/bash
# Collect PMC counters for the default bench workload, one counter group per
# pass (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots).
# Usage (on the GPU box): bash tools/pmc_pass.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
    python $R/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-single-step > $OUT/pass$i.log 2>&1
done
ls -R $OUT | head -40
