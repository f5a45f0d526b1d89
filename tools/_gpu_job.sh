cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 400 python -m pytest tests/test_geometry_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "backprojection or bvh or pipeline_end_to_end" --durations=5 > gpurun_out/r04_bp_tests_b.log 2>&1; echo "geometry pytest rc=$?"
tail -6 gpurun_out/r04_bp_tests_b.log | cut -c1-300
timeout -s KILL 200 python tools/bp_ab.py > gpurun_out/r04_bp_ab_b.log 2>&1; echo "bp_ab rc=$?"; grep "packet\|packed" gpurun_out/r04_bp_ab_b.log | cut -c1-220
timeout -s KILL 300 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -k "sequence_parallel" > gpurun_out/r04_sp_fp8_test.log 2>&1; echo "sp fp8 pytest rc=$?"; tail -4 gpurun_out/r04_sp_fp8_test.log | cut -c1-300
