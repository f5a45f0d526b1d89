cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_attention_fp8_gpu.py tests/test_e2e_tolerance_gpu.py -m gpu -q -s -k "fluxdit_with_fp8 or (full_schedule and fp8-attn)" > gpurun_out/r04_attn_fp8_e2e.log 2>&1; echo "pytest rc=$?"; grep -n "FluxDiT fp8\|full schedule\|passed\|failed\|Error\|assert" gpurun_out/r04_attn_fp8_e2e.log | head -14 | cut -c1-330
timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fp8 --fp8-attn > gpurun_out/r04_bench_fp8_attn.log 2>&1; echo "bench fp8-attn rc=$?"; tail -1 gpurun_out/r04_bench_fp8_attn.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 + fp8-attn ms/step', d['ms_per_step'], d['dtype'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['launch'])"
