#!/bin/bash
# ISA of ONE eight-lane step kernel in ~15 s (tools/isa_stats.sh compiles all instantiations: minutes).
# Usage: tools/isa_probe.sh <mode 0..6> [true|false (RAND)] [extra hipcc flags, e.g. -DUPKIE_PROBE_DEFAULT_SCALARS=true -DUPKIE_PROBE_IN_PLACE=true -DUPKIE_PROBE_OCTET_WAVES=1 -DUPKIE_OCTET_BUFFERED_STATE=1]; leaves k.s / remarks.txt in $OUT (default /tmp/isa_probe)
set -e
MODE=${1:-2}; RAND=${2:-false}; shift || true; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/isa_probe}; mkdir -p $OUT; cd $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-use-amdgpu-trackers=1 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
  -DUPKIE_PROBE_OCTET_MODE=$MODE -DUPKIE_PROBE_RAND=$RAND "$@" $R/upkie_amd/csrc/step_instances.hip -o k.s 2> remarks.txt || { tail -30 remarks.txt; exit 1; }
python3 - <<'PY'
import re
remarks = open('remarks.txt').read()
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)', remarks, re.S):
    if 'octet' in m.group(1):
        print(f"{m.group(1)[:60]} VGPR {m.group(2)} AGPR {m.group(3)} scratch {m.group(4)} B/lane, waves/SIMD {m.group(5)}, SGPR spills {m.group(6)}, VGPR spills {m.group(7)}, LDS {m.group(8)} B")
txt = open('k.s').read()
m = re.search(r'^(_ZN5upkie17step_kernel_octet[^\n]*):[^\n]*\n(.*?)\.Lfunc_end', txt, re.S | re.M)
n = sc = 0
for line in m.group(2).split('\n'):
    line = line.strip()
    if not line or line[0] in ';.' or line.endswith(':'): continue
    n += 1
    sc += line.startswith('scratch_')
print("static instructions", n, "scratch ops", sc)
PY
