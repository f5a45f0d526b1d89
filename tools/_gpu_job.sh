cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 16 python tools/gemm_nt_check.py > gpurun_out/r02_gemm_nt_check.log 2>&1
grep -v amdgpu gpurun_out/r02_gemm_nt_check.log
