#!/usr/bin/env python
"""Balanced tail round of the one-wave-per-SIMD GEMM (UTX_GEMM_STREAMK): correctness against the unsplit launch and an fp64 reference on
sampled elements, and the same-process interleaved A/B timing on the FLUX shapes whose last round is partly filled.
usage: gemm_streamk_check.py [--cost N ...]   (N > 1 = the margin in K-tiles the launcher's cost model has to clear, 1 = its default; several values are timed side by side)"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
dev = "cuda"
costs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1]
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
# (M, N, K, kind, K2): the image-token linears of the reference strip (S = 13 376 computed rows) and of strip1024x6 (50 240), pruned last block (6144 / 24576)
shapes = [(13376, 3072, 3072, "gate", 64), (13376, 3072, 12288, "gate", 64), (13376, 3072, 15360, "gate", 64), (13376, 9216, 3072, "bias", 64),
          (50240, 3072, 12288, "gate", 64), (50240, 3072, 15360, "gate", 64), (13001, 3072, 12288, "gelu", 64), (13824, 3072, 12288, "bias", 0),
          (13376, 21504, 3072, "bias", 64), (50240, 3072, 3072, "gate", 64), (6144, 3072, 15360, "gate", 64), (24576, 3072, 15360, "gate", 64),
          (6144, 12288, 3072, "gelu", 0), (12352, 3072, 12288, "gate", 64), (49216, 3072, 12288, "gate", 64),
          (13824, 3072, 12288, "gate", 64), (13824, 3072, 15360, "gate", 64), (13824, 3072, 3072, "gate", 64), (50688, 3072, 15360, "gate", 64), (50688, 3072, 3072, "gate", 64)]
import time
T0 = time.time()
for M, N, K, kind, K2 in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    kw = dict(bias=bias)
    if K2:
        kw.update(A2=(torch.randn(M, K2, device=dev, generator=g) * 0.1).to(torch.bfloat16), B2=(torch.randn(N, K2, device=dev, generator=g) * 0.1).to(torch.bfloat16))
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    if kind == "gate":
        kw.update(gate=torch.randn(N, device=dev, generator=g).to(torch.bfloat16), res=res)
    if kind == "gelu":
        kw.update(gelu_from=0)
    def run(mode):
        _lib.set_option("UTX_GEMM_STREAMK", mode)
        return ops.gemm(A, B, **kw)
    C0 = run(0).clone(); C1 = run(costs[0]).clone()
    d = (C0.float() - C1.float()).abs()
    ne = int((C0 != C1).sum())
    # fp64 reference on sampled rows of the tail tiles (the last rows of M) and of the head
    # (no BLAS call: elementwise products + sums on 24 rows x 192 columns)
    rows = torch.cat([torch.arange(0, 8, device=dev), torch.arange(M - 8, M, device=dev), torch.arange(M // 2, M // 2 + 8, device=dev)])
    cols = torch.cat([torch.arange(0, 64, device=dev), torch.arange(N - 64, N, device=dev), torch.arange(N // 2, N // 2 + 64, device=dev)])
    y = (A[rows].double()[:, None, :] * B[cols].double()[None, :, :]).sum(-1)
    if K2: y = y + (kw["A2"][rows].double()[:, None, :] * kw["B2"][cols].double()[None, :, :]).sum(-1)
    y = y + bias[cols].double()
    if kind == "gelu": y = 0.5 * y * (1.0 + torch.tanh(0.7978845608028654 * (y + 0.044715 * y ** 3)))
    if kind == "gate": y = res[rows][:, cols].double() + kw["gate"][cols].double() * y
    e0 = (C0[rows][:, cols].double() - y).abs().max().item(); e1 = (C1[rows][:, cols].double() - y).abs().max().item()
    C1b = run(costs[0])
    det = bool((C1b == C1).all())
    ts = {c: [] for c in [0] + costs}
    for c in ts: run(c); run(c)
    for r in range(7):
        for c in ts: ts[c].append(t1(lambda: run(c)))
    med = {c: sorted(v)[len(v) // 2] for c, v in ts.items()}
    tiles = ((M + 255) // 256) * (N // 256)
    print("M=%6d N=%6d K=%6d %-4s K2=%2d tiles %5d (%.2f rounds) | differing %8d of %d max|d| %.4f | err vs fp64: unsplit %.4f split %.4f | deterministic %s | ms: %s" % (
        M, N, K, kind, K2, tiles, tiles / 256.0, ne, C0.numel(), d.max().item(), e0, e1, det,
        "  ".join("%s %.3f" % ("off" if c == 0 else "on" if c == 1 else "margin%d" % c, med[c]) for c in ts)), "| t=%.0fs" % (time.time() - T0), flush=True)
_lib.set_option("UTX_GEMM_STREAMK", 1)
