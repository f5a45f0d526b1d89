# round 6, call 34: soak of the whole full-size denoise step on fixed inputs (80 forwards against the first one's bits)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/step_soak.py 80 2>&1 | grep -v amdgpu | tee gpurun_out/r06_step_soak.log
