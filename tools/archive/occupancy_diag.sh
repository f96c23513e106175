#!/bin/bash
# Round 3, VERDICT r2 item 1: why a second wavefront per SIMD of the eight-lane kernel bought nothing.
# A = the round-2 library (2112 B of private scratch per lane in a never-taken path), B = this tree's library
# (that workspace in LDS). Usage (GPU box): bash tools/occupancy_diag.sh <libA.so> ; writes gpurun_out/occ/*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=${1:-$R/ab/libupkie_hip_r02.so}
OUT=$R/gpurun_out/occ
mkdir -p $OUT
cd $R
export UPKIE_LANES_PER_ENV=8
echo "# B: this tree (limit-path workspace in LDS)" > $OUT/lanes8_sweep.txt
python tools/quick_bench.py 4096 8192 16384 32768 >> $OUT/lanes8_sweep.txt 2>&1
echo "# A: round-2 library (2112 B/lane private scratch), default scratch limit" >> $OUT/lanes8_sweep.txt
UPKIE_HIP_LIBRARY=$A python tools/quick_bench.py 4096 8192 16384 32768 >> $OUT/lanes8_sweep.txt 2>&1
echo "# A with HSA_SCRATCH_SINGLE_LIMIT=1 GiB" >> $OUT/lanes8_sweep.txt
HSA_SCRATCH_SINGLE_LIMIT=1073741824 UPKIE_HIP_LIBRARY=$A python tools/quick_bench.py 8192 16384 32768 >> $OUT/lanes8_sweep.txt 2>&1
echo "# A with HSA_SCRATCH_SINGLE_LIMIT=1 GiB and HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4 GiB" >> $OUT/lanes8_sweep.txt
HSA_SCRATCH_SINGLE_LIMIT=1073741824 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4294967296 UPKIE_HIP_LIBRARY=$A python tools/quick_bench.py 16384 >> $OUT/lanes8_sweep.txt 2>&1
echo "# A with HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" >> $OUT/lanes8_sweep.txt
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 UPKIE_HIP_LIBRARY=$A python tools/quick_bench.py 16384 >> $OUT/lanes8_sweep.txt 2>&1
unset UPKIE_LANES_PER_ENV
echo "# B, automatic mapping" >> $OUT/lanes8_sweep.txt
python tools/quick_bench.py 4096 8192 16384 32768 65536 >> $OUT/lanes8_sweep.txt 2>&1
cat $OUT/lanes8_sweep.txt
# residency counters of the 16384-env launch, both libraries (one counter group per pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for WHICH in A B; do
  if [ $WHICH = A ]; then export UPKIE_HIP_LIBRARY=$A; else unset UPKIE_HIP_LIBRARY; fi
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    UPKIE_LANES_PER_ENV=8 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WHICH}_pass$i -o pmc -- \
      python $R/bench.py --envs-per-gpu 16384 --steps 128 --warmup 64 --no-cpu-baseline --no-fused > $OUT/pmc_${WHICH}_pass$i.log 2>&1
  done
done
unset UPKIE_HIP_LIBRARY
cd $R
python - <<'PY'
import csv, glob, os, collections
root = "gpurun_out/occ"
for which in "AB":
    sums, counts, dur = collections.defaultdict(float), collections.defaultdict(int), []
    for path in glob.glob(f"{root}/pmc_{which}_pass*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "step_kernel_octet" in row["Kernel_Name"]:
                sums[row["Counter_Name"]] += float(row["Counter_Value"]); counts[row["Counter_Name"]] += 1
    for path in glob.glob(f"{root}/pmc_{which}_pass*/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "step_kernel_octet" in row["Kernel_Name"]:
                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    mean = {k: sums[k] / counts[k] for k in sums}
    d = sum(dur) / max(len(dur), 1)
    line = {"library": which, "launch_ns_under_profiler": d, **mean}
    if "SQ_WAVE_CYCLES" in mean and d:
        # SQ_WAVE_CYCLES: quad-cycles summed over waves; resident waves per SIMD averaged over the launch
        line["mean_resident_waves_per_simd"] = mean["SQ_WAVE_CYCLES"] * 4 / (d * 1e-9 * 2.4e9 * 1024)
    print(line)
    open(f"{root}/residency_{which}.json", "w").write(__import__("json").dumps(line, indent=1))
PY
