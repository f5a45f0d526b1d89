# round 6, call 32: the round's last tree -- full GPU suite, smoke, bench with no flags and with the driver's arguments
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_gpu_suite_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r06_smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06_smoke_final.log
timeout 600 python bench.py > gpurun_out/r06_bench_default_flags.json.log 2> gpurun_out/r06_bench_default.stderr.log; echo "bench (no flags) rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_strip1024x6_final.json.log 2> gpurun_out/r06_bench_final.stderr.log; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_default_flags", "r06_bench_strip1024x6_final"):
    d = json.loads(open("gpurun_out/%s.json.log" % f).read().strip().split("\n")[-1])
    print(f, {k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["roofline"]["achieved"], d["roofline"]["frac"], d.get("roofline_gemm", {}).get("frac"), d["config"].get("experiments_summary"), d["config"].get("ref512x6_ms_per_step"))
PY
