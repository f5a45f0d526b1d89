cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 1: the new parity tests, the MX fp8 one-wave-per-SIMD kernel (correctness, then speed), a first bench line of the round
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -s -m gpu > gpurun_out/r03_fp8_tests_a.log 2>&1; echo "fp8 tests rc=$?"
tail -n 25 gpurun_out/r03_fp8_tests_a.log | grep -v amdgpu
timeout 300 python tools/perf_fp8.py > gpurun_out/r03_perf_fp8_v0.log 2>&1; echo "perf_fp8 rc=$?"
grep -v amdgpu gpurun_out/r03_perf_fp8_v0.log
timeout 900 python -m pytest tests/test_e2e_tolerance_gpu.py -x -q -s -m gpu > gpurun_out/r03_e2e_tol_a.log 2>&1; echo "e2e rc=$?"
grep -v amdgpu gpurun_out/r03_e2e_tol_a.log | tail -n 30
timeout 600 python -m pytest tests/test_geometry_gpu.py -x -q -m gpu -k "baseline_config_sizes" > gpurun_out/r03_geom_cfg5_a.log 2>&1; echo "geom rc=$?"
tail -n 8 gpurun_out/r03_geom_cfg5_a.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_strip1024x6_v0.json.log 2>&1; echo "bench rc=$?"
tail -n 3 gpurun_out/r03_bench_strip1024x6_v0.json.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8 > gpurun_out/r03_bench_strip1024x6_fp8_v0.json.log 2>&1; echo "bench fp8 rc=$?"
tail -n 3 gpurun_out/r03_bench_strip1024x6_fp8_v0.json.log
