cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_attention_fp8_gpu.py -q -s -m gpu 2>&1 | grep -E "key split|passed|failed|Error" | tee gpurun_out/r06_fp8_split_tests.log
