#!/usr/bin/env python
"""Effective clock of our GEMM kernels vs the vendor library's on one FLUX shape: run under
   rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE ... (tools/gemm_clock_probe.sh) -- clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.
Decides whether a lead is cycles (schedule) or clock (power).  PROBE_TILES = comma list of UTX_GEMM_TILE[:UTX_GEMM_DEBUG] variants
(ablation build when a debug value is given)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
variants = [v.split(":") for v in os.environ.get("PROBE_TILES", "0").split(",")]
if any(len(v) > 1 for v in variants):
    _lib.use_ablation_library()
from unitex_amd.flux import ops
M, N, K = 50688, 21504, 3072
if len(sys.argv) > 3:
    M, N, K = map(int, sys.argv[1:4])
A = (torch.randn(M, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
bias = torch.randn(N, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    for v in variants:
        _lib.set_option("UTX_GEMM_TILE", int(v[0]))
        if len(v) > 1: _lib.set_option("UTX_GEMM_DEBUG", int(v[1]))
        ops.gemm(A, B, out=C, bias=bias)
    if os.environ.get("PROBE_LIB", "1") == "1":
        torch.nn.functional.linear(A, B, bias)
torch.cuda.synchronize()
