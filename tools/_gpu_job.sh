# round 6, call 30: tests/test_fullsize_gpu.py with the tile-order model of the re-chosen group_m
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu > gpurun_out/r06_fullsize_tests_gm.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_fullsize_tests_gm.log
