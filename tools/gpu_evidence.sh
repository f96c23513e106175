#!/bin/bash
# On the GPU box: the round's measured evidence in one call (every step under `timeout`):
#   bench.py default and with the driver's flags -> gpurun_out/<tag>_bench_n1.json, <tag>_bench_driver_window.json
#   rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<tag>_kernel_stats/
#   the PMC passes of tools/pmc_pass.sh -> gpurun_out/pmc_<tag>/
# Usage: bash tools/gpu_evidence.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc $?"
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_window.json 2> /dev/null; echo "driver-window bench rc $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_kernel_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kernel_stats -o ks -- \
  python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_kernel_stats.log 2>&1; echo "rocprofv3 stats rc $?"
find $R/gpurun_out/${TAG}_kernel_stats -name "*kernel_trace.csv" -delete  # (tens of MB; the stats summary is what is kept)
cd $R
timeout 600 bash tools/pmc_pass.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1; echo "pmc rc $?"
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -size +8M -delete
f=$(find gpurun_out/${TAG}_kernel_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-160
du -sh gpurun_out | tail -1
# the Bullet-like kernel's counters (bench's c2_bullet_like_contact_model.bullet_like.roofline reads profiles/pmc_bullet_like_b4096.json)
timeout 900 bash tools/pmc_bullet_like.sh $TAG > gpurun_out/pmc_bullet_like_${TAG}.log 2>&1; echo "pmc bullet-like rc $?"

