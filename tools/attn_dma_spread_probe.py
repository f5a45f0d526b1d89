#!/usr/bin/env python
"""NEGATIVE RESULT, kept for the record (the kernel variant was removed again: profiles/r02_attn_dma_spread_probe.log -- x0.999 at S = 13 376, x0.974 at
S = 50 240, and hipcc moved the builtin DMA across the tile fence: outputs differed).  With two waves per SIMD the partner wave hides the burst.
Attention kernel with the V^T half of the next tile's LDS-DMAs issued in the middle of the tile (VAR 6, correct results, ablation build) instead of
all four at the top: interleaved timing against the default, and agreement of the two outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
BF = torch.bfloat16
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
H = 24
for S in (13376, 50240):
    q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(BF); k = torch.randn(H, S, 128, device="cuda").to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF)
    outs = {v: torch.empty(S, H * 128, dtype=BF, device="cuda") for v in (0, 6)}
    def run(var):
        _lib.set_option("UTX_ATTN_VAR", var); ops.attention(q, k, vt, S=S, out=outs[var], scale=0.0)
    ts = {0: [], 6: []}
    for v in ts: run(v); run(v)
    for r in range(7):
        for v in ts: ts[v].append(t1(lambda: run(v)))
    med = {v: sorted(x)[len(x) // 2] for v, x in ts.items()}
    fl = 4.0 * S * S * 128 * H
    same = torch.equal(outs[0].view(torch.int16), outs[6].view(torch.int16))
    print("attn S=%6d | burst of 4 %7.3f ms %6.0f TF | K at the top, V mid-tile %7.3f ms %6.0f TF | x%.3f | outputs identical %s" % (
        S, med[0], fl / med[0] / 1e9, med[6], fl / med[6] / 1e9, med[0] / med[6], same), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
