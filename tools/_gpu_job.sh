# round 6, call 40: the WHOLE GPU suite on the tree with the launch-option leak guard (tests/conftest.py) -- every later test now provably runs on the defaults
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2100 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r06_gpu_suite_guarded.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r06_gpu_suite_guarded.log | tail -1; grep -E "^(FAILED|ERROR)" gpurun_out/r06_gpu_suite_guarded.log | head -20
