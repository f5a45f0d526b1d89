#!/usr/bin/env python
"""Placement of the sixteen LDS-DMA pieces (and the two scale loads) of a K-tile over the two K-steps of the MX fp8 one-wave-per-SIMD GEMM
(gemm_w4.hip, W4_XPIECE_OF): the default (5 + 11, two per three slots) against three other placements, same process, interleaved, every arm correct
and bit-identical.  Ablation library (the variants are template instances); the split tail round is off in every arm."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops, mx8
ctx = ops.get_ctx(0)
_lib.set_option("UTX_GEMM_STREAMK", 0)
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
names = {0: "default 5+11, scales in slots 0,1", 1024: "3+13", 2048: "8+8", 4096: "5+11, scales in slots 8,9"}   # profiles/r03_mx_dma_sweep_v0.log predates the adoption: its "default" had the scales in slots 8, 9
for M, N, K in [(50688, 9216, 3072), (50688, 21504, 3072), (50688, 3072, 12288), (50688, 3072, 15360), (13824, 21504, 3072)]:
    A = (torch.randn(M, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    wq, wp = mx8.quantize_weight(B, ctx, packed=True); aq, ap = mx8.quantize_act(A, ctx, packed=True)
    outs = {v: torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for v in names}
    def run(v):
        _lib.set_option("UTX_GEMM_DEBUG", 32 * v); ops.gemm(aq, wq, bias=bias, out=outs[v], a_scale=ap, b_scale=wp, sk_work=None)
    ts = {v: [] for v in names}
    for v in names: run(v); run(v)
    torch.cuda.synchronize()
    same = {v: bool(torch.equal(outs[v], outs[0])) for v in names}
    for r in range(7):
        for v in names: ts[v].append(t1(lambda: run(v)))
    med = {v: sorted(x)[len(x) // 2] for v, x in ts.items()}
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%6d | " % (M, N, K) + " | ".join("%s %6.3f ms %5.0f TF x%.3f %s" % (names[v], med[v], fl / med[v] / 1e9, med[0] / med[v], "=" if same[v] else "BITS DIFFER") for v in names), flush=True)
_lib.set_option("UTX_GEMM_DEBUG", 0); _lib.set_option("UTX_GEMM_STREAMK", 1)
