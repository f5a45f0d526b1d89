cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python tools/host_profile_pipeline.py > gpurun_out/r03_host_profile_pipeline.log 2>&1; echo "rc=$?"
grep -n ">>>" gpurun_out/r03_host_profile_pipeline.log | tail -8
grep -n "cumulative" -A 45 gpurun_out/r03_host_profile_pipeline.log | head -75 | cut -c1-200
