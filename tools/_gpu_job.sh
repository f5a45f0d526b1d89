cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r02_gpu_tests_gemm_w4.log 2>&1
tail -5 gpurun_out/r02_gpu_tests_gemm_w4.log
