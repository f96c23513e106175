#!/bin/bash
# MFMA counters of mpc_step_kernel (BASELINE C3: 16384 envs, N = 16, 15 over-relaxed ADMM
# iterations by default since round 4): instruction count, MFMA busy cycles, wave cycles; one kernel
# trace + stats pass beside the counter passes. Usage: bash tools/pmc_mpc.sh <tag> [horizon, default 16]
# (round 6: horizons > 16 profile mpc_step_h_kernel, the fp16 matrix path; with UPKIE_MPC_FP32=1 in the environment the fp32 kernels: mpc_step_tail_kernel at
# N = 50, with UPKIE_MPC_FOUR_TILES=1 as well round 5's mpc_step_kernel<4, 13>)
set -u
TAG=${1:-r02}
N=${2:-16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_mpc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python $R/tools/mpc_loop.py 16384 200 $N > $OUT/pass$i.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ks -- python $R/tools/mpc_loop.py 16384 300 $N > $OUT/stats.log 2>&1
ls -R $OUT | head -30
