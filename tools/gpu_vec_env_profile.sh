#!/bin/bash
# On the GPU box: per-kernel times of the PUBLIC loop `env.step(policy(obs))` (tools/bench_vec_env.py, NEXT_STEP, no graphs)
# from rocprofv3's kernel trace -> gpurun_out/prof_vec_env/. Every step under `timeout`: a profiler that does not exit
# must not eat the call's limit (round 4 lost 15 GPU-minutes to one that was run without --output-format csv).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_vec_env
rm -rf $OUT; mkdir -p $OUT
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o vec -- python $R/tools/bench_vec_env.py 4096 1500 next_step --no-graph > $OUT/run.log 2>&1
echo "rocprofv3 rc $?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-220
# the same loop with the policy as ONE launch / inside the step's launch
for P in one_launch in_launch; do
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$P -o vec -- python $R/tools/bench_vec_env.py 4096 1500 next_step --no-graph --policy $P > $OUT/run_$P.log 2>&1
  echo "$P rocprofv3 rc $?"; grep "us per env.step" $OUT/run_$P.log
  f=$(find $OUT/$P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -5 "$f" | cut -c1-220
  find $OUT/$P -name "*kernel_trace.csv" -delete
done
