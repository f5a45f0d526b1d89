# First GPU call of round 5 (prepared at the end of round 4, which had no GPU minutes left):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/_gpu_job.sh'
# 1. the opt-in peeled attention loop (UTX_ATTN_PEEL, DESIGN 8 "What the default kernel's listing shows"): bit-identity first, then the interleaved A/B
# 2. the GPU suite on the tree as it is (the last complete run of round 4 was one commit series earlier: profiles/r04_gpu_suite_final.log)
# 3. the default bench line + rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
( UTX_RUN_UNVALIDATED=1 timeout -s KILL 300 python -m pytest tests/test_attention_peel_gpu.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r05_attn_peel_tests.log 2>&1
echo "peel tests: $(tail -1 gpurun_out/r05_attn_peel_tests.log)"
if grep -q " passed" gpurun_out/r05_attn_peel_tests.log && ! grep -q "failed" gpurun_out/r05_attn_peel_tests.log; then
  timeout -s KILL 240 python tools/attn_peel_ab.py > gpurun_out/r05_attn_peel_ab.log 2>&1; cat gpurun_out/r05_attn_peel_ab.log | tail -12
fi
( timeout -s KILL 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r05_gpu_suite_a.log 2>&1
echo "suite: $(grep -E 'passed|failed|error' gpurun_out/r05_gpu_suite_a.log | tail -1)"
timeout -s KILL 200 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_bench_strip1024x6_v0.json.log 2> gpurun_out/r05_bench_v0.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r05_bench_strip1024x6_v0.json.log
