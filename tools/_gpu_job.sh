cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_gpu_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03_final_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r03_final_bench.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/r03_final_bench.log
