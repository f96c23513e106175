#!/bin/bash
# Collect PMC counters for the default bench workload, one counter group per
# pass (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots), with
# --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
# The timed launch shape is bench.py's contract shape: 4096 envs, ONE
# env.step() per kernel launch.
# Usage (on the GPU box): bash tools/pmc_pass.sh <tag> [extra bench.py flags]
#   then: python tools/pmc_summary.py <tag>        -> profiles/pmc_step_b4096.json
set -u
TAG=${1:-r02}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
  "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
    python $R/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-fused --no-steady-state --no-secondary "$@" > $OUT/pass$i.log 2>&1
done
ls -R $OUT | head -40
