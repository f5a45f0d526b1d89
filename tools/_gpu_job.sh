# round 6, call 18: bench with the generated GEMM K loop + GEMM / fullsize tests
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_determinism_stress_gpu.py "tests/test_dit_ops_gpu.py" -q -m gpu -k "gemm or full_width or stress or plan or lora" > gpurun_out/r06_gemm_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06_gemm_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_strip1024x6_v2.json.log 2> gpurun_out/r06_bench_v2.stderr.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_strip1024x6_v2.json.log").read().strip().split("\n")[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["achieved"], d["roofline"]["frac"], d.get("roofline_gemm", {}).get("frac"), d.get("roofline_gemm", {}).get("sum_ms_per_step"), d["config"].get("experiments_summary"), d["config"].get("ref512x6_ms_per_step"))
PY
