cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 175 python -m pytest -x -q -s "tests/test_attention_fp8_gpu.py::test_fluxdit_with_fp8_attention_runs_the_fp8_kernel_and_stays_close_to_the_bf16_forward" "tests/test_fp8_gpu.py::test_sequence_parallel_single_blocks_run_their_projection_in_fp8" > gpurun_out/r04_fp8_attn_plan_sp_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r04_fp8_attn_plan_sp_tests.log
