cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_fullsize_gpu.py -x -q -k "forced_range" > gpurun_out/r02_streamk_forced_tests.log 2>&1
tail -12 gpurun_out/r02_streamk_forced_tests.log
