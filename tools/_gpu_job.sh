cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 75 python -m pytest tests/test_fullsize_gpu.py tests/test_dit_ops_gpu.py -x -q -k "split_tail or forced_range or edge_shapes or gemm" > gpurun_out/r02_gemm_tests_i.log 2>&1
tail -3 gpurun_out/r02_gemm_tests_i.log
