# round 6, call 5: the whole GPU suite with the 4 x 64 kernel as the default attention path
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite_a.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r06_gpu_suite_a.log | tail -40
