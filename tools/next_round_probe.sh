#!/bin/bash
# First GPU job of the next round (DESIGN.md section 9, item 1c'): ~1.5 GPU-minutes.
#   gpurun --timeout 300 -- 'bash tools/next_round_probe.sh'
# Nontemporal C stores became the default at the very end of round 2 on the strength of the kernel-level probe (profiles/r02_gemm_w4_probe_nt.log) and a
# bit comparison (tools/gemm_nt_check.py).  Still to do: the step-level A/B -- ABL 512 in the ablation library = plain stores -- and the same hint on the
# other streaming writers (qkv_post, ln_mod, the split tail's partial tiles, the attention output).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 200 python tools/gemm_w4_probe.py 0,16,512 > gpurun_out/r03_gemm_w4_probe_nt.log 2>&1
grep -v amdgpu gpurun_out/r03_gemm_w4_probe_nt.log
