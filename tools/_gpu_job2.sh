cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout -s KILL 900 python -m pytest tests/test_e2e_tolerance_gpu.py -m gpu -q -x -s -k "full_depth_full_width_fp8" 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r05_full_depth_fp8.log 2>&1
cat gpurun_out/r05_full_depth_fp8.log
