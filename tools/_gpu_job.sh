# round 6, call 13: MX fp8 attention -- row sum over the e4m3 probabilities (UTX_ATTN8_LQ) vs over their fp32 values: the outlier rows and the price
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
{ timeout 600 python -m pytest tests/test_attention_fp8_gpu.py -q -s -m gpu 2>&1 | grep -E "fp8 attention, S|passed|failed"; timeout 600 python tools/attn_fp8_perf.py 2>&1 | grep -v amdgpu; } | tee gpurun_out/r06_fp8_attn_lq.log
