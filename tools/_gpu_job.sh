cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 150 python -m pytest -x -q tests/test_attention_fp8_gpu.py tests/test_fp8_gpu.py "tests/test_dit_ops_gpu.py::test_c_built_dit_plan_equals_the_python_built_one" > gpurun_out/r04_fp8_files_final.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r04_fp8_files_final.log
