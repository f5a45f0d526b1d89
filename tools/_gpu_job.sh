cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/small_gemm_census.py > gpurun_out/r02_small_gemm_census.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_small_gemm_census.log
