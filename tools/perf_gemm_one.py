"""single-shape GEMM loop for PMC profiling: python tools/perf_gemm_one.py M N K"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unitex_amd.flux import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
A = (torch.randn(M, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
bias = torch.randn(N, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(6):
    ops.gemm(A, B, bias=bias, out=C)
torch.cuda.synchronize()
