#!/usr/bin/env python
"""Where a sub-stage of gemm256_w4_kernel goes: slope (us per 64 k) and intercept (us per tile) of time vs K with parts of the loop
removed (ablation build, wrong results): UTX_GEMM_DEBUG bits 5.. = the kernel's ABL template value (1 no DMA, 2 no fragment reads, 4 no vmcnt wait / barrier, 8 no epilogue,
16 no C stores, 32 DMA cursor parked, 64 two 16x16x32 per MFMA; correct results: 256 start-time stagger of the workgroups, 512 plain instead of nontemporal C stores).  argv[1] = comma list of ABL values to run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
dev = "cuda:0"
def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
_lib.set_option("UTX_GEMM_TILE", 2564)
M, N, rounds = 50688, 3072, 10      # 2376 tiles: with UTX_GEMM_STREAMK on, the 10th round would be split; the probe keeps whole rounds
_lib.set_option("UTX_GEMM_STREAMK", 0)
CASES = (("full", 0), ("no DMA", 1), ("no frag reads", 2), ("no DMA, no reads", 3), ("no wait/barrier", 4), ("MFMA only", 7), ("no epilogue", 8), ("MFMA only, no epilogue", 15), ("no C stores", 16), ("DMA of the same 1 KB", 32), ("two 16x16x32 per MFMA", 64), ("start stagger", 256), ("plain (write-allocate) C stores", 512))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if str(c[1]) in sys.argv[1].split(","))
for name, abl in CASES:
    _lib.set_option("UTX_GEMM_DEBUG", abl << 5)
    pts = []
    for K in (64, 1024, 3072, 6144):
        A = (torch.rand(M, K, device=dev) - 0.5).to(torch.bfloat16); B = (torch.rand(N, K, device=dev) - 0.5).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(A, B, out=C, bias=torch.zeros(N, device=dev, dtype=torch.bfloat16)))
        pts.append((K // 64, ms * 1e3 / rounds))
    (k0, t0), (k1, t1) = pts[-2], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print("%-26s" % name, " ".join("nk=%d:%.1f" % p for p in pts), "| slope %.2f us/K-tile, intercept %.1f us/tile" % (slope, t0 - slope * k0), flush=True)
