#!/usr/bin/env python
"""One large FLUX linear in isolation for PMC passes (tools/pmc_kernel.sh): M = 50 688, N = 21 504, K = 3072, bias epilogue.
usage: python tools/gemm_one.py [bf16|mx8] [M N K]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops, mx8
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (50688, 21504, 3072)
ctx = ops.get_ctx(0)
A = (torch.randn(M, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
bias = torch.randn(N, device="cuda").to(torch.bfloat16); C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
if mode == "mx8":
    wq, wp = mx8.quantize_weight(B, ctx, packed=True); aq, ap = mx8.quantize_act(A, ctx, packed=True)
    f = lambda: ops.gemm(aq, wq, bias=bias, out=C, a_scale=ap, b_scale=wp)
else:
    f = lambda: ops.gemm(A, B, bias=bias, out=C)
for _ in range(4):
    f()
torch.cuda.synchronize()
