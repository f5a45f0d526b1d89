cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/gemm_streamk_check.py 1 > gpurun_out/r02_gemm_streamk_check_v5.log 2>&1; cut -c1-70 gpurun_out/r02_gemm_streamk_check_v5.log > /tmp/a; grep -o "deterministic.*" gpurun_out/r02_gemm_streamk_check_v5.log > /tmp/b; paste /tmp/a /tmp/b
