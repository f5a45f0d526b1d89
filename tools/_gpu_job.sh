cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed" gpurun_out/r03_final_gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03_smoke_final.log
timeout 600 python bench.py > gpurun_out/r03_final_bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r03_final_bench.log | cut -c1-200
