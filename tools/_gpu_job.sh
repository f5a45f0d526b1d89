# round 6, call 48: BASELINE configs[4]-sized workloads on the round's kernels: the joint strip at 2048^2 x 8 (S = 263 680 nominal) in bf16, and the per-view shape (S = 34 304) in bf16 / MX fp8 / MX fp8 + fp8 attention
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
export UTX_BENCH_EXPERIMENTS=0 UTX_BENCH_REF_POINT=0 UTX_BENCH_GRAPH_FIGURE=0
timeout 1200 python bench.py --workload strip2048x8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_strip2048x8.json.log 2> gpurun_out/r06_bench_strip2048x8.stderr.log; echo "strip2048x8 rc=$?"
timeout 600 python bench.py --workload view2048 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_view2048_bf16.json.log 2> gpurun_out/r06_bench_view2048.stderr.log; echo "view2048 bf16 rc=$?"
timeout 600 python bench.py --workload view2048 --steps 5 --warmup 2 --no-cpu-baseline --fp8 > gpurun_out/r06_bench_view2048_fp8.json.log 2>> gpurun_out/r06_bench_view2048.stderr.log; echo "view2048 fp8 rc=$?"
timeout 600 python bench.py --workload view2048 --steps 5 --warmup 2 --no-cpu-baseline --fp8 --fp8-attn > gpurun_out/r06_bench_view2048_fp8_attn.json.log 2>> gpurun_out/r06_bench_view2048.stderr.log; echo "view2048 fp8-attn rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_strip2048x8", "r06_bench_view2048_bf16", "r06_bench_view2048_fp8", "r06_bench_view2048_fp8_attn"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json.log" % f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["dtype"][:40], "attn", round(d["roofline"]["achieved"], 1), d["roofline"]["kernel"][:24], round(d["roofline"]["frac"], 4), "gemm", d.get("roofline_gemm", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
