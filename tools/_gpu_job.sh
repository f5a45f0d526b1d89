# round 6, call 8: upper bound of the 16x16x32 MFMA shape for gemm256_w4_kernel (ablation arm: same operand traffic, wrong results) on the adoption-bar shapes, beside hipBLASLt
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gemm_w4_shape_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm_w4_shape_probe.log
