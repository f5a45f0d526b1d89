# round 6, call 17: the GEMM's steady-state K loop as a generated stream (UTX_GEMM_FASTK): bit identity + timing beside hipBLASLt
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/gemm_fastk_check.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_gemm_fastk_check_v0.log
