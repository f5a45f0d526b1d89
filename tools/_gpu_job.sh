# round 6, call 47: the maximum-size attention parity case (S = 263 232: the joint strip at configs[4]'s resolution), both kernels
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_q64_gpu.py -q -m gpu --durations=5 > gpurun_out/r06_attn_max_size_test.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu gpurun_out/r06_attn_max_size_test.log | tail -15
