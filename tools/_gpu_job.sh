# round 6, call 43: epilogue arms of the 4 x 64 attention kernel (ablation library): product stores / no stores (bound) / O through LDS -- bit identity + interleaved timing
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/attn_q64_epi_ab.py > gpurun_out/r06_attn_q64_epi_ab.log 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/r06_attn_q64_epi_ab.log | tail -24
