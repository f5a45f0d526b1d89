#!/usr/bin/env python
"""Sequence-parallel attention launch (periodic key multiplicity) on one GPU: the KBP instance of the fast loop (UTX_ATTN_PEEL=1) against the general loop (=0): bits + time.
Shape: what ONE of P = 4 / 8 ranks runs: H / P heads x the full sequence of P blocks of (64 text + S_loc image) rows, key weight 8 on every block's first tile."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
BF = torch.bfloat16
for P, S_loc in ((4, 12544), (8, 6272)):
    H = 24 // P
    blk = 64 + S_loc
    S = P * blk
    per = blk // 64
    g = torch.Generator(device="cuda").manual_seed(S)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    fl = 4.0 * S * S * 128 * H
    res = {}
    for peel in (0, 1):
        _lib.set_option("UTX_ATTN_PEEL", peel)
        ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=3.0, key_bias_period=per, out=out); torch.cuda.synchronize()
        res[peel] = out.clone()
    print("P = %d: %d heads x %d tokens, period %d tiles: fast (KBP) loop bit-identical to the general loop: %s" % (P, H, S, per, bool(torch.equal(res[0].view(torch.int16), res[1].view(torch.int16)))), flush=True)
    t = {0: [], 1: []}
    for _ in range(5):
        for peel in (0, 1):
            _lib.set_option("UTX_ATTN_PEEL", peel)
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=3.0, key_bias_period=per, out=out)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _r in range(3): ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=3.0, key_bias_period=per, out=out)
            b.record(); torch.cuda.synchronize()
            t[peel].append(a.elapsed_time(b) / 3)
    for peel in (0, 1):
        m = sorted(t[peel])[2]
        print("   UTX_ATTN_PEEL=%d  med %8.3f ms -> %7.1f TF/s" % (peel, m, fl / m / 1e9), flush=True)
_lib.set_option("UTX_ATTN_PEEL", 1)
