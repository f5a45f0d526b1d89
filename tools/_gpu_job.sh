cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r02_gpu_tests_attn_full.log 2>&1
tail -5 gpurun_out/r02_gpu_tests_attn_full.log
