#!/usr/bin/env python
"""Ablation timing of the 8-phase GEMM: UTX_GEMM_DEBUG 0 = real, 1 = no staging (schedule ceiling),
2 = always stage K-tile 0 (fill path with every request an L2 hit); UTX_GEMM_GROUP_M sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda:0"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
from unitex_amd import _lib
_lib.use_ablation_library()   # build with `python unitex_amd/csrc/build.py --ablate`
from unitex_amd.flux import ops
_lib.set_option("UTX_GEMM_TILE", 256)
for (M, N, K) in [(8192, 8192, 8192), (50688, 3072, 3072), (50688, 12288, 3072), (50688, 3072, 12288), (13824, 9216, 3072)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = "gemm8 M=%6d N=%6d K=%6d :" % (M, N, K)
    for dbg in ("0", "1", "2"):
        _lib.set_option("UTX_GEMM_DEBUG", int(dbg))
        ms = timeit(lambda: ops.gemm(A, B, out=C))
        line += "  dbg%s %7.3f ms %7.1f TF/s" % (dbg, ms, 2.0 * M * N * K / ms / 1e9)
    _lib.set_option("UTX_GEMM_DEBUG", 0)
    for gm in ("1", "2", "4", "8", "16"):
        _lib.set_option("UTX_GEMM_GROUP_M", int(gm))
        ms = timeit(lambda: ops.gemm(A, B, out=C))
        line += "  gm%s %6.1f" % (gm, 2.0 * M * N * K / ms / 1e9)
    _lib.set_option("UTX_GEMM_GROUP_M", 0)
    print(line, flush=True)
