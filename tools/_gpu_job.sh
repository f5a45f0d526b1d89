cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 7: PMC passes (MFMA busy, clock, HBM bytes) of the MX fp8 and bf16 one-wave-per-SIMD GEMM
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/r03_pmc_gemm_mx8 gemm256_w4 python $GRAFT_REPO_ROOT/tools/gemm_one.py mx8 > gpurun_out/r03_pmc_gemm_mx8.log 2>&1; echo "pmc mx8 rc=$?"
cat gpurun_out/r03_pmc_gemm_mx8.log
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/r03_pmc_gemm_bf16 gemm256_w4 python $GRAFT_REPO_ROOT/tools/gemm_one.py bf16 > gpurun_out/r03_pmc_gemm_bf16.log 2>&1; echo "pmc bf16 rc=$?"
cat gpurun_out/r03_pmc_gemm_bf16.log
tail -n 3 gpurun_out/r03_pmc_gemm_mx8/sq1.log
rm -rf gpurun_out/r03_pmc_gemm_mx8 gpurun_out/r03_pmc_gemm_bf16
