cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 560 python bench.py --workload strip2048x8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_strip2048x8_v0.json.log 2>&1; echo "rc=$?"
grep '^{' gpurun_out/r03_bench_strip2048x8_v0.json.log | cut -c1-600
tail -4 gpurun_out/r03_bench_strip2048x8_v0.json.log | cut -c1-300
