#!/bin/bash
# Static ISA statistics of the step kernels: registers, scratch, spills per kernel, and the instruction
# histogram of one of them (default: the bench's eight-lane kernel, step_kernel_octet<MODE_PENDULUM_AGENT, false, true, false>).
# Usage: tools/isa_stats.sh [mangled-kernel-prefix] [extra hipcc flags]
set -e
K=${1:-_ZN5upkie17step_kernel_octetILi2ELb0ELb1ELb0ELb0EEE}
shift || true
D=$(mktemp -d)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $D
# the library's flags (upkie_amd/lib.py): SLP-packing scalar fp32 chains into v_pk_* costs registers and moves; the step
# kernels are compiled by groups (csrc/step_instances.hpp), side by side
for g in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-use-amdgpu-trackers=1 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" \
    -DUPKIE_INSTANCE_GROUP=$g $R/upkie_amd/csrc/step_instances.hip -o k$g.s 2> remarks$g.txt &
done
wait
cat remarks*.txt > remarks.txt
cat k[0-9]*.s > k.s
grep -q "error:" remarks.txt && { grep -A5 "error:" remarks.txt | head -40; exit 1; }
python3 - "$K" <<'PY'
import collections, re, sys
remarks = open('remarks.txt').read()
print("kernel                               VGPR AGPR scratch  SGPR-spill VGPR-spill")
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)', remarks, re.S):
    name = m.group(1)
    if 'step_kernel' in name:
        short = re.sub(r'EEvPK.*', '', name.replace('_ZN5upkie', ''))
        print(f"{short:36s} {m.group(2):>4s} {m.group(3):>4s} {m.group(4):>7s} {m.group(5):>11s} {m.group(6):>10s}")
txt = open('k.s').read()
m = re.search(r'^' + re.escape(sys.argv[1]) + r'[^\n]*:[^\n]*\n(.*?)\.Lfunc_end', txt, re.S | re.M)
ops = collections.Counter()
for line in m.group(1).split('\n'):
    line = line.strip()
    if not line or line[0] in ';.' or line.endswith(':'): continue
    ops[line.split()[0]] += 1
grp = collections.Counter()
for k, v in ops.items():
    if 'dpp' in k: grp['dpp'] += v
    elif k.startswith(('v_fma', 'v_pk_fma', 'v_fmac', 'v_mul_f32', 'v_pk_mul', 'v_add_f32', 'v_sub_f32', 'v_pk_add', 'v_mac', 'v_mad_f32')): grp['math'] += v
    elif k.startswith(('v_mov', 'v_pk_mov')): grp['v_mov'] += v
    elif 'accvgpr' in k: grp['accvgpr'] += v
    elif 'readlane' in k or 'writelane' in k: grp['sgpr_spill'] += v
    elif k.startswith('scratch'): grp['scratch'] += v
    elif k.startswith('s_'): grp['salu'] += v
    else: grp['other_valu'] += v
print(sys.argv[1], "static total", sum(ops.values()), dict(grp), "(whole kernel, rare paths included; tools/isa_blocks.py lists the blocks)")
PY
rm -rf $D
