cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/gemm_zero_vs_random.py > gpurun_out/r03_gemm_zero_vs_random.log 2>&1; echo "rc=$?"
cat gpurun_out/r03_gemm_zero_vs_random.log
