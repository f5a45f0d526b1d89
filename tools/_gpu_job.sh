cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 9: full-depth (19 + 38 blocks) full-width forward vs the oracle; DMA-placement sweep of the MX fp8 GEMM
timeout 900 python -m pytest tests/test_e2e_tolerance_gpu.py -x -q -s -m gpu -k "full_depth" > gpurun_out/r03_full_depth_a.log 2>&1; echo "full depth rc=$?"
grep -v amdgpu gpurun_out/r03_full_depth_a.log | tail -n 12
timeout 400 python tools/mx_dma_sweep.py > gpurun_out/r03_mx_dma_sweep_v0.log 2>&1; echo "sweep rc=$?"
grep -v amdgpu gpurun_out/r03_mx_dma_sweep_v0.log
