cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_vec_env.py 4096 2000 next_step,same_step > gpurun_out/r04_vec_env_b.txt 2>&1
cat gpurun_out/r04_vec_env_b.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_vec_env -o vec -- python $R/tools/bench_vec_env.py 4096 2000 next_step --no-graph > $R/gpurun_out/r04_vec_env_prof.log 2>&1
cd $R; ls gpurun_out/prof_vec_env | head; f=$(ls gpurun_out/prof_vec_env/*kernel_stats.csv | head -1); head -12 $f
