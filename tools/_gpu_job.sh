cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_gpu_suite_b.log 2>&1; echo "FULL GPU SUITE rc=$?"
tail -22 gpurun_out/r04_gpu_suite_b.log | cut -c1-220
timeout -s KILL 200 python __graft_entry__.py smoke > gpurun_out/r04_smoke_b.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_smoke_b.log
timeout -s KILL 300 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/r04_bench_d.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r04_bench_d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['launch'], d['config']['text_half_of_double_blocks'][:30], d['config']['ms_per_step_hip_graph_replay'], 'attn frac', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'gemm', d.get('roofline_gemm',{}).get('frac'), 'bp', d['config']['backprojection']['total_ms'], 'sec/mesh', d['config']['sec_per_mesh_texture'])"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04b -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_rocprof_b.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r04b -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; if [ -n "$f" ]; then head -4 "$f" | cut -c1-200; cp "$f" gpurun_out/r04_rocprofv3_kernel_stats_strip1024x6_v2.csv; fi
find gpurun_out/prof_r04b -name "*kernel_trace.csv" -delete 2>/dev/null
