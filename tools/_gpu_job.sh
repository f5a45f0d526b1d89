cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_f.log 2>&1
grep -E "passed|failed" gpurun_out/r02_gpu_tests_f.log | tail -2
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_strip1024x6_v5.json.log 2>&1
python bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_ref512x6_v4.json.log 2>&1
for f in gpurun_out/r02_bench_strip1024x6_v5.json.log gpurun_out/r02_bench_ref512x6_v4.json.log; do grep '^{' $f | cut -c1-190; done
