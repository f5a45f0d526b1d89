cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "mx_fp8" > gpurun_out/r03_maxsize_test_fp8.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_maxsize_test_fp8.log | cut -c1-300
