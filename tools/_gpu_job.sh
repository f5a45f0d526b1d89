cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 400 python -m pytest tests/test_determinism_stress_gpu.py tests/test_dit_ops_gpu.py tests/test_fp8_gpu.py -m gpu -q --durations=5 > gpurun_out/r04_final_subset.log 2>&1; echo "subset rc=$?"; tail -12 gpurun_out/r04_final_subset.log | cut -c1-200
timeout -s KILL 100 python __graft_entry__.py smoke > gpurun_out/r04_smoke_c.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_smoke_c.log
timeout -s KILL 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --workload ref512x6 > gpurun_out/r04_bench_ref512_final.log 2>&1; tail -1 gpurun_out/r04_bench_ref512_final.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref512x6 ms/step', d['ms_per_step'], d['config']['text_half_of_double_blocks'][:40])"
