#!/usr/bin/env python
"""Where does the per-tile intercept of the persistent GEMM come from?  time per tile vs K-tiles (a) by grid size (fewer workgroups
than CUs: less pressure on L2 / fabric / HBM per tile boundary), (b) in the ablation build without C stores / without any epilogue.
   python tools/gemm_pers_probe.py [--ablate]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
ABL = "--ablate" in sys.argv
if ABL:
    _lib.use_ablation_library()
from unitex_amd.flux import ops
dev = "cuda:0"
def timeit(fn, n=12):
    for _ in range(4): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
N = 3072
def sweep(tag, grid, M):
    _lib.set_option("UTX_GEMM_PERS_GRID", grid)
    tiles = (M // 256) * (N // 256)
    g = grid if grid else 256
    per_wg = -(-tiles // g)
    pts = []
    for K in (64, 1024, 3072, 6144):
        A = (torch.rand(M, K, device=dev) - 0.5).to(torch.bfloat16); B = (torch.rand(N, K, device=dev) - 0.5).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(A, B, out=C, bias=torch.zeros(N, device=dev, dtype=torch.bfloat16)))
        pts.append((K // 64, ms * 1e3 / per_wg))
    (k0, t0), (k1, t1) = pts[-2], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print("%-34s grid %3d tiles/WG %3d |" % (tag, g, per_wg), " ".join("nk=%d:%.1f" % p for p in pts), "| slope %.2f us/K-tile, intercept %.1f us/tile" % (slope, t0 - slope * k0), flush=True)
for grid in (0, 128, 64, 32):
    M = 256 * 10 * (grid if grid else 256) // 12     # 10 tiles per workgroup at every grid size
    M = (M // 256) * 256
    sweep("stores on", grid, M)
if ABL:
    for dbg, tag in ((4, "no C stores"), (8, "no epilogue")):
        _lib.set_option("UTX_GEMM_DEBUG", dbg)
        sweep(tag, 0, 256 * 10 * 256 // 12 // 256 * 256)
    _lib.set_option("UTX_GEMM_DEBUG", 0)
