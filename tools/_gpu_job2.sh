cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout -s KILL 300 python -m pytest tests/test_attention_fp8_gpu.py -m gpu -q -s -k "long_diffuse" 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r05_fp8_attn_outlier.log 2>&1
cat gpurun_out/r05_fp8_attn_outlier.log
