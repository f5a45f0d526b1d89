cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -x -q -m gpu -k "c_built or c_side" > gpurun_out/r03_c_dit_tests_a.log 2>&1; echo "c dit tests rc=$?"
grep -v amdgpu gpurun_out/r03_c_dit_tests_a.log | tail -n 30
