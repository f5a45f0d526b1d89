# round 6, call 35: census of the step's GEMM launches by shape and kernel, each timed in isolation (where the 128 x 128 kernel's 2 % of the step goes)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/small_gemm_census.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_small_gemm_census.log
