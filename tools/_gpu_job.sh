# round 6, call 46: the WHOLE GPU suite with the conftest guard extended to UTX_* environment variables
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2100 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite_closing_v1.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r06_gpu_suite_closing_v1.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/r06_gpu_suite_closing_v1.log | head -20
