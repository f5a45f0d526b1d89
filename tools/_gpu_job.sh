# round 6, call 33: the QK^T MFMAs of the 4 x 64 stream as two 16x16x32 each (arm 12, wrong results by design): the bound of what the vendor GEMM's MFMA shape could buy the attention kernel
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
UTX_CHECK_QUICK=1 UTX_CHECK_ARMS=0,12 timeout 900 python tools/attn_q64_check.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_attn_q64_arms_v6.log
UTX_ONE_S=50240 bash tools/attn_pmc_arms.sh gpurun_out/pmc_q64_m16 "q64:UTX_ONE_ABLATE=1,UTX_ATTN_Q64=1,UTX_ATTN_VAR=0" "mfma16qk:UTX_ONE_ABLATE=1,UTX_ATTN_Q64=1,UTX_ATTN_VAR=12" 2>&1 | tee -a gpurun_out/r06_attn_q64_arms_v6.log
rm -rf gpurun_out/pmc_q64_m16/*/
