cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 10: attention A/B (nontemporal output stores), fp8 perf after the scale-load move, bench pair
timeout 300 python tools/attn_variants_r03.py > gpurun_out/r03_attn_variants_v1.log 2>&1; echo "variants rc=$?"
grep -v amdgpu gpurun_out/r03_attn_variants_v1.log
timeout 300 python tools/perf_fp8.py > gpurun_out/r03_perf_fp8_v1.log 2>&1; echo "perf_fp8 rc=$?"
grep -v amdgpu gpurun_out/r03_perf_fp8_v1.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8 > gpurun_out/r03_bench_strip1024x6_fp8_v2.json.log 2>&1; echo "bench fp8 rc=$?"
tail -n 1 gpurun_out/r03_bench_strip1024x6_fp8_v2.json.log | cut -c 1-200
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_strip1024x6_v3.json.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/r03_bench_strip1024x6_v3.json.log | cut -c 1-200
