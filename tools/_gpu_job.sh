R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_g.log 2>&1
tail -4 gpurun_out/r02_gpu_tests_g.log
timeout 240 python bench.py > gpurun_out/r02_bench_strip1024x6_v6.json 2> gpurun_out/r02_bench_strip1024x6_v6.err
cat gpurun_out/r02_bench_strip1024x6_v6.json | cut -c1-400
timeout 200 python bench.py --workload ref512x6 --no-cpu-baseline > gpurun_out/r02_bench_ref512x6_v5.json 2>/dev/null
cat gpurun_out/r02_bench_ref512x6_v5.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_v4.log 2>&1
find $R/gpurun_out/prof_v4 -name "*kernel_stats.csv" | head -2
