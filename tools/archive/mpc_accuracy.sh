#!/bin/bash
# One-off (round 6): the evidence of profiles/r06_mpc_f16_split.txt -- accuracy against the fp64 checker and launch time of the
# balancer's kernels on the fp16 matrix path (default) and on the fp32 one (UPKIE_MPC_FP32=1), the hardware checks behind the former.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== tools/archive/mpc_debug.py: |first input - fp64 checker| [m/s2], 500 envs, four steps from an unrelated warm start (the last two saturate the bounds)"
echo "-- default (horizons > 16: fp16 matrix path, two terms per operand, u_q = Minv q out of the loop)"
timeout 300 python tools/archive/mpc_debug.py 2>&1 | grep "first_input" | sed 's/; z error.*//'
echo "-- UPKIE_MPC_FP32=1 (the fp32 MFMA kernels of rounds 2-6)"
UPKIE_MPC_FP32=1 timeout 300 python tools/archive/mpc_debug.py 2>&1 | grep "first_input" | sed 's/; z error.*//'
echo "== tools/mpc_time.py"
timeout 120 python tools/mpc_time.py 2>&1 | grep -v amdgpu.ids
UPKIE_MPC_FP32=1 timeout 120 python tools/mpc_time.py 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/mpc_time.py 2>&1 | grep -v amdgpu.ids
echo "== tools/microbench/f16_split_check.hip"
hipcc --offload-arch=gfx950 -O2 tools/microbench/f16_split_check.hip -o /tmp/f16_split_check 2>/dev/null && /tmp/f16_split_check
echo "== tools/microbench/mfma_shadow.hip"
hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_shadow.hip -o /tmp/mfma_shadow 2>/dev/null && /tmp/mfma_shadow
