cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8 -o fp8 -- python $R/bench.py --fp8 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03_rocprof_bench_fp8.log 2>&1; echo "rc=$?"
cd $R
tail -1 gpurun_out/r03_rocprof_bench_fp8.log | cut -c1-600
find gpurun_out/prof_fp8 -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_fp8 -name "*kernel_stats.csv" | head -1)
head -25 "$f"
cp "$f" gpurun_out/r03_rocprofv3_kernel_stats_strip1024x6_fp8.csv
find gpurun_out/prof_fp8 -name "*kernel_trace*" -delete
