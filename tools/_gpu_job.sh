cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -m gpu -x -q -k "sequence_parallel or block_strided" > gpurun_out/r03_zero_copy_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03_zero_copy_tests.log | cut -c1-250
