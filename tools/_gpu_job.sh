cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 2: MX w4 kernel tests, attention epilogue / workspace changes, attention variants A/B, raw-mesh condition render, bench with roofline_gemm + new CPU baseline
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -s -m gpu > gpurun_out/r03_fp8_tests_b.log 2>&1; echo "fp8 tests rc=$?"
grep -v amdgpu gpurun_out/r03_fp8_tests_b.log | tail -n 12
timeout 900 python -m pytest tests/test_dit_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "attention or pruning or tail_split or sharded or sequence" > gpurun_out/r03_attn_tests_b.log 2>&1; echo "attn tests rc=$?"
tail -n 6 gpurun_out/r03_attn_tests_b.log
timeout 300 python tools/attn_variants_r03.py > gpurun_out/r03_attn_variants_v0.log 2>&1; echo "variants rc=$?"
grep -v amdgpu gpurun_out/r03_attn_variants_v0.log
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r03_pipeline_tests_b.log 2>&1; echo "pipeline rc=$?"
tail -n 6 gpurun_out/r03_pipeline_tests_b.log
timeout 600 python bench.py --steps 4 --warmup 1 > gpurun_out/r03_bench_strip1024x6_v1.json.log 2>&1; echo "bench rc=$?"
tail -n 2 gpurun_out/r03_bench_strip1024x6_v1.json.log | cut -c 1-3000
