cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 6: the whole GPU suite on the current tree (C-side plan replay is now the default forward path), PMC passes of the GEMMs, reference operating point bench
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03_gpu_tests_e.log 2>&1; echo "gpu suite rc=$?"
tail -n 15 gpurun_out/r03_gpu_tests_e.log
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/r03_pmc_gemm_mx8 gemm256_w4 python tools/gemm_one.py mx8 > gpurun_out/r03_pmc_gemm_mx8.log 2>&1; echo "pmc mx8 rc=$?"
cat gpurun_out/r03_pmc_gemm_mx8.log; tail -n 5 gpurun_out/r03_pmc_gemm_mx8/sq1.log; ls gpurun_out/r03_pmc_gemm_mx8/sq1 | head
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/r03_pmc_gemm_bf16 gemm256_w4 python tools/gemm_one.py bf16 > gpurun_out/r03_pmc_gemm_bf16.log 2>&1; echo "pmc bf16 rc=$?"
cat gpurun_out/r03_pmc_gemm_bf16.log
find gpurun_out/r03_pmc_gemm_mx8 gpurun_out/r03_pmc_gemm_bf16 -name "*.csv" -size +2000k -delete
timeout 300 python bench.py --workload ref512x6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_ref512x6_v0.json.log 2>&1; echo "bench ref512 rc=$?"
tail -n 1 gpurun_out/r03_bench_ref512x6_v0.json.log | cut -c 1-200
timeout 300 python bench.py --workload ref512x6 --steps 10 --warmup 3 --no-cpu-baseline --fp8 > gpurun_out/r03_bench_ref512x6_fp8_v0.json.log 2>&1; echo "bench ref512 fp8 rc=$?"
tail -n 1 gpurun_out/r03_bench_ref512x6_fp8_v0.json.log | cut -c 1-200
