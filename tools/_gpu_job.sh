cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 60 python tools/attn_fp8_perf.py > gpurun_out/r04_attn_fp8_perf_v2.log 2>&1; echo "perf rc=$?"; tail -3 gpurun_out/r04_attn_fp8_perf_v2.log
timeout -s KILL 90 python -m pytest -x -q tests/test_attention_fp8_gpu.py > gpurun_out/r04_attn_fp8_tests_v2.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04_attn_fp8_tests_v2.log
