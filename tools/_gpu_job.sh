cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/gemm_w4_check.py > gpurun_out/r02_gemm_w4_check_v8.log 2>&1
timeout 300 python tools/gemm_w4_trace.py > gpurun_out/r02_gemm_w4_trace_v5.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_gemm_w4_check_v8.log; grep "tile [3-6]:" gpurun_out/r02_gemm_w4_trace_v5.log | head -4
