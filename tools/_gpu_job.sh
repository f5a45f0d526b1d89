# round 5, call 1: the opt-in peeled attention loops (UTX_ATTN_PEEL 1..6, UTX_ATTN8_PEEL=1) on hardware: bit-identity tests, then the interleaved A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
( UTX_RUN_UNVALIDATED=1 timeout -s KILL 300 python -m pytest tests/test_attention_peel_gpu.py -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r05_attn_peel_tests.log 2>&1
echo "peel tests: $(tail -1 gpurun_out/r05_attn_peel_tests.log)"
timeout -s KILL 300 python tools/attn_peel_ab.py > gpurun_out/r05_attn_peel_ab.log 2>&1; tail -40 gpurun_out/r05_attn_peel_ab.log
