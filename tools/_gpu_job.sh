# round 5: the GPU suite + the default bench line (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout -s KILL 1300 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r05_gpu_suite_a.log 2>&1
echo "suite: $(grep -E 'passed|failed|error' gpurun_out/r05_gpu_suite_a.log | tail -1)"
timeout -s KILL 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_bench_strip1024x6_v1.json.log 2> gpurun_out/r05_bench_v1.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r05_bench_strip1024x6_v1.json.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_strip1024x6_v1.json.log') if x.startswith('{')]
if l:
    o=json.loads(l[-1]); c=o['config']
    print(o['ms_per_step'], o['roofline']['achieved'], o['roofline']['frac'], o.get('roofline_gemm',{}).get('frac'))
    for k in ('experiments_summary','ref512x6_ms_per_step','backprojection_total_ms','backprojection_kernel_sum_ms','gemm_frac','ms_per_step_hip_graph_replay','sec_per_mesh_texture'): print(k, c.get(k))
PY
