# round 6, call 42: the sequence-parallel model path after the scratch-sizing fix (S_q > S_kv): SP / S_q > S_kv tests, the 2-rank rehearsal of bench.py on one GPU over gloo
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dit_ops_gpu.py -q -m gpu -k "sequence_parallel or more_queries or sp_ or relayout or ulysses" > gpurun_out/r06_sp_scratch_fix_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r06_sp_scratch_fix_tests.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/r06_sp_scratch_fix_tests.log | head
UTX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r06_bench_strip1024x6_2ranks_1gpu_v1.json.log 2> gpurun_out/r06_bench_2ranks_v1.stderr.log; echo "2ranks rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_strip1024x6_2ranks_1gpu_v1",):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json.log" % f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r06_bench_2ranks_v1.stderr.log | cut -c1-300
