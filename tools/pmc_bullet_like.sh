#!/bin/bash
# VALU / wait counters of the C2 workload under the Bullet-like contact model (the mode closest to
# pybullet_backend.py:306 reported like the headline): the eight-lane kernel step_kernel_octet<2, false, false, false, true>,
# 4096 envs, one env.step() per launch (tools/profile_bullet_like.py: 1100 launches), one counter group per pass, --kernel-trace only.
# Usage (GPU box): bash tools/pmc_bullet_like.sh <tag>; then
#   python tools/pmc_summary.py bullet_like_<tag> --kernel "step_kernel_octet<2, false, false, false, true>" --out profiles/pmc_bullet_like_b4096.json
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_bullet_like_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python $R/tools/profile_bullet_like.py c2 > $OUT/pass$i.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls $OUT
