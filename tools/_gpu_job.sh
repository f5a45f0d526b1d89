cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -m gpu -x -q -k "sequence_parallel_two_ranks" > gpurun_out/r03_sp_tests_d.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_sp_tests_d.log
