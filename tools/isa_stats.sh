#!/bin/bash
# Static ISA statistics of step_kernel<MODE_PENDULUM_AGENT,false>: registers, spills,
# instruction histogram. Usage: tools/isa_stats.sh [extra hipcc flags]
set -e
D=$(mktemp -d)
cd $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -save-temps -Rpass-analysis=kernel-resource-usage "$@" /root/repo/upkie_amd/csrc/upkie_hip.hip -o x.o 2> remarks.txt || { tail -20 remarks.txt; exit 1; }
grep -A10 "step_kernelILi2ELb0" remarks.txt | grep -E "VGPRs:|AGPRs|Scratch|Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo
python3 - <<'PY'
import re, collections
txt = open('upkie_hip-hip-amdgcn-amd-amdhsa-gfx950.s').read()
m = re.search(r'^_ZN5upkie11step_kernelILi2ELb0ELi1EEE.*?:\n(.*?)\.Lfunc_end', txt, re.S | re.M)
ops = collections.Counter()
for line in m.group(1).split('\n'):
    line = line.strip()
    if not line or line[0] in ';.' or line.endswith(':'): continue
    ops[line.split()[0]] += 1
tot = sum(ops.values())
grp = collections.Counter()
for k, v in ops.items():
    if k.startswith(('v_fma', 'v_pk_fma', 'v_fmac', 'v_mul_f32', 'v_pk_mul', 'v_add_f32', 'v_sub_f32', 'v_pk_add', 'v_mac', 'v_mad_f32')): grp['math'] += v
    elif k.startswith(('v_mov', 'v_pk_mov')): grp['v_mov'] += v
    elif 'accvgpr' in k: grp['accvgpr'] += v
    elif 'readlane' in k or 'writelane' in k: grp['sgpr_spill'] += v
    elif k.startswith('scratch'): grp['scratch'] += v
    elif k.startswith('s_'): grp['salu'] += v
    else: grp['other_valu'] += v
print("static total", tot, dict(grp))
PY
rm -rf $D
