cd $GRAFT_REPO_ROOT
python tools/perf_fp8.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_perf_fp8_v0.log
