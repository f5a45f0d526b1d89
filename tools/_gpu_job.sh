cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r02_bench_gemm_w4_ab.log
for rep in 1 2; do
  for t in 2560 0; do
    echo "== UTX_GEMM_TILE=$t strip1024x6" >> gpurun_out/r02_bench_gemm_w4_ab.log
    UTX_GEMM_TILE=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['launch_options']['UTX_GEMM_TILE'])" >> gpurun_out/r02_bench_gemm_w4_ab.log
  done
done
for t in 2560 0 2560 0; do
  echo "== UTX_GEMM_TILE=$t ref512x6" >> gpurun_out/r02_bench_gemm_w4_ab.log
  UTX_GEMM_TILE=$t python bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['launch_options']['UTX_GEMM_TILE'])" >> gpurun_out/r02_bench_gemm_w4_ab.log
done
cat gpurun_out/r02_bench_gemm_w4_ab.log
