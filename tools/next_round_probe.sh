#!/bin/bash
# First GPU job of the next round (DESIGN.md section 9, item 1c'): ~1.5 GPU-minutes.
#   gpurun --timeout 300 -- 'bash tools/next_round_probe.sh'
# 1. Do nontemporal C stores give the operand panels their L2 back?  ABL 0 (full kernel), 16 (no C stores: the 6.4 us per tile to win at most),
#    512 (nontemporal C stores, correct results), 256 (start stagger, for reference) -- slope / intercept per K-tile, M = 50 688, N = 3072.
# 2. The same question at step level is only worth asking if 512 recovers a good part of what 16 does: then make the hint the default of
#    W4_STORE_U (gemm_w4.hip) and A/B bench.py on both workloads.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 200 python tools/gemm_w4_probe.py 0,16,256,512 > gpurun_out/r03_gemm_w4_probe_nt.log 2>&1
grep -v amdgpu gpurun_out/r03_gemm_w4_probe_nt.log
