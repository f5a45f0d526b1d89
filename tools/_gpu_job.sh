# round 6, call 45: bench.py's N > 1 control flow after the extras watchdog: 2 ranks on one GPU over gloo -- once normally (full line), once with the watchdog forced (minimal line, rc 0)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
UTX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r06_bench_2ranks_1gpu_v2.json.log 2> gpurun_out/r06_bench_2ranks_v2.stderr.log; echo "2ranks rc=$?"
UTX_BENCH_EXTRAS_TIMEOUT=0.05 UTX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r06_bench_2ranks_1gpu_watchdog.json.log 2> gpurun_out/r06_bench_2ranks_wd.stderr.log; echo "2ranks (watchdog forced) rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_2ranks_1gpu_v2", "r06_bench_2ranks_1gpu_watchdog"):
    try:
        ls = [l for l in open("gpurun_out/%s.json.log" % f).read().strip().split("\n") if l.startswith("{")]
        d = json.loads(ls[-1])
        print(f, len(ls), "line(s)", d["n_gpus"], d["ms_per_step"], d["scaling"], "roofline" in d, d["config"].get("note", "")[:60])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r06_bench_2ranks_wd.stderr.log | cut -c1-200
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('N=1', d['ms_per_step'], d['roofline']['frac'])"
