cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_pipeline_gpu.py -x -q > gpurun_out/r02_pipeline_tests_h.log 2>&1
tail -3 gpurun_out/r02_pipeline_tests_h.log
timeout 300 python tools/run_full_pipeline.py --reps 2 > gpurun_out/r02_full_pipeline_e2e_v3.log 2>&1
grep -v "^$" gpurun_out/r02_full_pipeline_e2e_v3.log | tail -10
