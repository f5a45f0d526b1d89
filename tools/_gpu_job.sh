cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dit_ops_gpu.py tests/test_c_host_gpu.py tests/test_abi.py -m gpu -x -q -k "c_built or plain_c or abi" > gpurun_out/r03_c_dit_tests_fp8.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_c_dit_tests_fp8.log
