cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_e.log 2>&1
grep -E "passed|failed" gpurun_out/r02_gpu_tests_e.log | tail -2
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_strip1024x6_v4.json.log 2>&1
python bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_ref512x6_v3.json.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_rocprofv3_bench_strip1024x6_v3.log 2>&1)
cp gpurun_out/prof_r02/*kernel_stats.csv gpurun_out/r02_rocprofv3_kernel_stats_strip1024x6_v3.csv 2>/dev/null
rm -rf gpurun_out/prof_r02
for f in gpurun_out/r02_bench_strip1024x6_v4.json.log gpurun_out/r02_bench_ref512x6_v3.json.log; do grep '^{' $f | cut -c1-190; done
head -6 gpurun_out/r02_rocprofv3_kernel_stats_strip1024x6_v3.csv | cut -c1-150
