# round 6, call 41: the round's closing evidence on the final tree -- smoke, bench with the driver's flags, bench without flags, rocprofv3 --kernel-trace --stats of the bench command
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r06_smoke_closing.log 2>&1; tail -2 gpurun_out/r06_smoke_closing.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_strip1024x6_closing.json.log 2> gpurun_out/r06_bench_closing.stderr.log; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_bench_strip1024x6_closing.json.log') if x.startswith('{')]
d=json.loads(l[-1]); print(len(l), 'json line(s):', d['value'], d['unit'], d['ms_per_step'], 'ms/step; roofline', d['roofline']['frac'], 'gemm', d.get('gemm_frac'), 'cpu', d['cpu_baseline']['value'])
PY
timeout 600 python bench.py > gpurun_out/r06_bench_default_flags_closing.json.log 2> gpurun_out/r06_bench_default_closing.stderr.log; echo "bench (no flags) rc=$?"; tail -c 300 gpurun_out/r06_bench_default_flags_closing.json.log | head -c 300; echo
cd /tmp
UTX_BENCH_EXPERIMENTS=0 UTX_BENCH_REF_POINT=0 UTX_BENCH_GRAPH_FIGURE=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_closing -o run -- python $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r06_rocprofv3_closing.bench_line.json 2> $R/gpurun_out/r06_rocprofv3_closing.stderr.log; echo "rocprofv3 rc=$?"
cd $R
f=$(find gpurun_out/prof_closing -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp "$f" gpurun_out/r06_rocprofv3_kernel_stats_strip1024x6_closing.csv && head -8 "$f" | cut -c1-160
rm -rf gpurun_out/prof_closing
