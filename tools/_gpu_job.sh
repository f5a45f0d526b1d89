cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "gemm" 2>&1 | tail -3
python tools/gemm_ab.py > gpurun_out/r02_gemm_ab_v3.log 2>&1
cat gpurun_out/r02_gemm_ab_v3.log
