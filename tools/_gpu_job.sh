# round 6, call 44: the WHOLE GPU suite on the tree with the S_q > S_kv scratch fix (library changed: capi.cpp)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2100 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r06_gpu_suite_closing.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r06_gpu_suite_closing.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/r06_gpu_suite_closing.log | head -20
