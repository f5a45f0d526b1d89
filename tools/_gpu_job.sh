R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref512 -- python $R/bench.py --workload ref512x6 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_ref512.log 2>&1
find $R/gpurun_out/prof_ref512 -name "*kernel_stats.csv" | head -2
