# round 6, call 12: the three-tiles-per-trip stream as the product default: its own test file, the q64 repair test, check tool, bench
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_q64_gpu.py "tests/test_dit_ops_gpu.py::test_attention_q64_kernel_and_repair_pass" tests/test_attention_peel_gpu.py -q -m gpu > gpurun_out/r06_q64_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06_q64_tests.log
timeout 600 python tools/attn_q64_check.py > gpurun_out/r06_attn_q64_check_v1.log 2>&1; echo "check rc=$?"; grep -v amdgpu.ids gpurun_out/r06_attn_q64_check_v1.log | tail -6
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_strip1024x6_v1.json.log 2> gpurun_out/r06_bench_strip1024x6_v1.stderr.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_strip1024x6_v1.json.log").read().strip().split("\n")[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["achieved"], d["roofline"]["frac"], d.get("roofline_gemm", {}).get("frac"), d["config"].get("experiments_summary"), d["config"].get("ref512x6_ms_per_step"), d["config"].get("attn_clock_ghz"))
PY
