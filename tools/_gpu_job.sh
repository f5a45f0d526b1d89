cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python tools/mx_race_screen.py 25 > gpurun_out/r03_mx_race_screen.log 2>&1; echo "race screen rc=$?"
grep -v amdgpu gpurun_out/r03_mx_race_screen.log | tail -n 20
