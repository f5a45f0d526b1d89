cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref -o ref -- python $R/bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03_rocprof_bench_ref512x6.log 2>&1; echo "rc=$?"
cd $R
grep '^{' gpurun_out/r03_rocprof_bench_ref512x6.log | cut -c1-300
f=$(find gpurun_out/prof_ref -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
head -16 "$f" | cut -c1-220
cp "$f" gpurun_out/r03_rocprofv3_kernel_stats_ref512x6.csv
rm -rf gpurun_out/prof_ref
