#!/usr/bin/env python
"""Per-tile fixed cost of the 8-phase GEMM (prologue + epilogue): time vs K at fixed M, N; intercept of the linear fit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
from unitex_amd import _lib
_lib.set_option("UTX_GEMM_TILE", int(sys.argv[1]) if len(sys.argv) > 1 else 0)   # 0 = persistent kernel (default), 256 = per-tile 8-phase kernel
print("UTX_GEMM_TILE =", _lib.get_options()["UTX_GEMM_TILE"])
M, N = 50688, 3072     # 2376 tiles = 9.28 rounds of 256 CUs -> 10 rounds
rounds = 10
for kind in ("plain", "bias", "gate", "gelu"):
    pts = []
    for K in (64, 256, 1024, 3072, 6144):
        A = (torch.rand(M, K, device=dev) - 0.5).to(torch.bfloat16); B = (torch.rand(N, K, device=dev) - 0.5).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = {}
        if kind != "plain": kw["bias"] = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        if kind == "gate": kw.update(gate=torch.ones(N, device=dev, dtype=torch.bfloat16), res=C)
        if kind == "gelu": kw.update(gelu_from=0)
        ms = timeit(lambda: ops.gemm(A, B, out=C, **kw))
        pts.append((K // 64, ms * 1e3 / rounds))
    (k0, t0), (k1, t1) = pts[-2], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(kind, " ".join("nk=%d:%.1fus" % p for p in pts), "| slope %.2f us/K-tile, intercept %.1f us/tile; nk=1 tile %.1f us" % (slope, t0 - slope * k0, pts[0][1]))
