# round 6, call 23: seconds per mesh texture end to end (1024^2 x 6 and the reference's 512^2 x 6), the fp8 bench lines, the 2-rank rehearsal of bench.py on one GPU over gloo (key de-dup path)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/run_full_pipeline.py --view 1024 --reps 1 2>&1 | grep -v amdgpu | tail -25 > gpurun_out/r06_full_pipeline_e2e_1024.log; echo "e2e1024 rc=$?"; tail -6 gpurun_out/r06_full_pipeline_e2e_1024.log
timeout 600 python tools/run_full_pipeline.py --view 512 --reps 2 2>&1 | grep -v amdgpu | tail -25 > gpurun_out/r06_full_pipeline_e2e_512.log; echo "e2e512 rc=$?"; tail -4 gpurun_out/r06_full_pipeline_e2e_512.log
UTX_BENCH_EXPERIMENTS=0 UTX_BENCH_REF_POINT=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --fp8 > gpurun_out/r06_bench_strip1024x6_fp8.json.log 2> gpurun_out/r06_bench_fp8b.stderr.log; echo "fp8 rc=$?"
UTX_BENCH_EXPERIMENTS=0 UTX_BENCH_REF_POINT=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --fp8 --fp8-attn > gpurun_out/r06_bench_strip1024x6_fp8_attn_v1.json.log 2> gpurun_out/r06_bench_fp8c.stderr.log; echo "fp8attn rc=$?"
UTX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r06_bench_strip1024x6_2ranks_1gpu.json.log 2> gpurun_out/r06_bench_2ranks.stderr.log; echo "2ranks rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_strip1024x6_fp8", "r06_bench_strip1024x6_fp8_attn_v1", "r06_bench_strip1024x6_2ranks_1gpu"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json.log" % f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["ms_per_step"], d["roofline"]["achieved"], d["config"].get("sequence_parallel", d["config"].get("parallelism")))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r06_bench_2ranks.stderr.log | cut -c1-300
