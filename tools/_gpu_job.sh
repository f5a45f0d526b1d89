cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 4: fused fp8 quantisation + fp8 pruning, chart unwrap (exact overlap check), bf16 / fp8 bench pair on one box, rocprofv3 kernel stats of the bench
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -s -m gpu > gpurun_out/r03_fp8_tests_d.log 2>&1; echo "fp8 tests rc=$?"
grep -v amdgpu gpurun_out/r03_fp8_tests_d.log | tail -n 14
timeout 900 python -m pytest tests/test_geometry_gpu.py tests/test_pipeline_gpu.py tests/test_multigpu_gpu.py -x -q -m gpu > gpurun_out/r03_geom_pipe_tests_d.log 2>&1; echo "geom/pipe/multigpu rc=$?"
tail -n 12 gpurun_out/r03_geom_pipe_tests_d.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_strip1024x6_v2.json.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/r03_bench_strip1024x6_v2.json.log | cut -c 1-220
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8 > gpurun_out/r03_bench_strip1024x6_fp8_v1.json.log 2>&1; echo "bench fp8 rc=$?"
tail -n 1 gpurun_out/r03_bench_strip1024x6_fp8_v1.json.log | cut -c 1-220
UTX_FP8_FUSE_QUANT=0 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8 > gpurun_out/r03_bench_strip1024x6_fp8_v1_unfused.json.log 2>&1; echo "bench fp8 unfused rc=$?"
tail -n 1 gpurun_out/r03_bench_strip1024x6_fp8_v1_unfused.json.log | cut -c 1-220
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof_bench.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 14 $f | cut -c 1-200
# keep only the small stats file (the trace itself is tens of MB)
[ -n "$f" ] && cp $f gpurun_out/r03_rocprofv3_kernel_stats_strip1024x6_v1.csv; rm -rf gpurun_out/r03_prof
