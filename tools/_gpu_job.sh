# round 5, final tree: the GPU suite (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 1300 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite_full.log 2>&1; echo "rc=$?"
grep -E " passed| failed| error" gpurun_out/r05_gpu_suite_full.log | tail -2 | tee gpurun_out/r05_gpu_suite_final.log
