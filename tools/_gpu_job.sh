cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multigpu_gpu.py -x -q -m gpu > gpurun_out/r02_gpu_tests_multigpu.log 2>&1
tail -5 gpurun_out/r02_gpu_tests_multigpu.log
