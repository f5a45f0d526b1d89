cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout -s KILL 300 python -m pytest tests/test_attention_peel_gpu.py "tests/test_determinism_stress_gpu.py::test_fast_attention_loop_320_launches_against_the_general_loops_bits" "tests/test_dit_ops_gpu.py" -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r05_tests_b.log 2>&1
tail -5 gpurun_out/r05_tests_b.log
timeout -s KILL 120 python tools/attn_peel_ab.py > gpurun_out/r05_attn_fast_loop_ab.log 2>&1; tail -12 gpurun_out/r05_attn_fast_loop_ab.log
timeout -s KILL 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_bench_strip1024x6_v0.json.log 2> gpurun_out/r05_bench_v0.err; echo "bench rc=$?"; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_strip1024x6_v0.json.log') if x.startswith('{')]
if l:
    o=json.loads(l[-1]); c=o['config']
    print(o['ms_per_step'], o['roofline']['achieved'], o['roofline']['frac'], o.get('roofline_gemm',{}).get('frac'))
    for k in ('experiments_summary','ref512x6_ms_per_step','backprojection_total_ms','backprojection_kernel_sum_ms','gemm_frac','ms_per_step_hip_graph_replay','text_half_of_double_blocks'): print(k, c.get(k))
PY
tail -5 gpurun_out/r05_bench_v0.err
