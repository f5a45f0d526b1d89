cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "checkpoint or callback or add_lora" > gpurun_out/r03_pipeline_tests_g.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_pipeline_tests_g.log
