cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/attn_dma_spread_probe.py > gpurun_out/r02_attn_dma_spread_probe.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_attn_dma_spread_probe.log
