cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/attn_fixed_cost.py > gpurun_out/r03_attn_fixed_cost.log 2>&1; echo "rc=$?"
cat gpurun_out/r03_attn_fixed_cost.log
timeout 600 python -m pytest tests/test_dit_ops_gpu.py -m gpu -x -q -k "attention" > gpurun_out/r03_attn_tests_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_attn_tests_c.log
