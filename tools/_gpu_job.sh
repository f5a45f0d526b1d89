cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dit_ops_gpu.py tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py tests/test_vae_gpu.py -x -q -m gpu > gpurun_out/r02_gpu_tests_lnmod.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r02_gpu_tests_lnmod.log | tail -5
