cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_fullsize_gpu.py tests/test_dit_ops_gpu.py -m gpu -x -q -k "full_width_dit_blocks or fused" > gpurun_out/r03_fullwidth_fix.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|FAILED\|warn" gpurun_out/r03_fullwidth_fix.log | tail -5 | cut -c1-300
