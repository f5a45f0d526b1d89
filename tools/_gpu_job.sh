cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_dit_ops_gpu.py -x -q -k "gemm or tiny_dit" 2>&1 | tail -3
