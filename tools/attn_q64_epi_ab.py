#!/usr/bin/env python
"""Epilogue of the 4 x 64 attention kernel (ablation library, UTX_ATTN_DEBUG bits of attention_q64.hip): 0 = the product's stores (a lane writes 8 bytes of ONE row per
instruction: 32 rows x 16 B per store), 32 = no stores at all (the bound), 64 = O through LDS (a store instruction writes 4 rows x 256 contiguous bytes).
Bit identity of arm 64 against arm 0 over shapes incl. S_q < S, S_q > S, ragged query counts and the key-split tail round, then interleaved timing."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
BF = torch.bfloat16


def mk(H, S, Sq, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Sqp = (max(Sq or S, S) + 63) // 64 * 64
    Qh = (torch.randn(H, Sqp, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    return Qh, Kh, Vt


def run(dbg, Qh, Kh, Vt, S, kb=0.0, Sq=None, out=None):
    _lib.set_option("UTX_ATTN_DEBUG", dbg)
    o = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=Sq, out=out)
    torch.cuda.synchronize()
    return o


bad = 0
cases = [(1, 64, None, 0.0), (3, 128, None, 0.0), (2, 512, 300, 3.0), (2, 1024, 1000, 0.0), (3, 2048, 2048 + 192, 2.0), (1, 640, 640 + 448, 0.0), (24, 3328, None, 3.0), (24, 4160, 2816, 0.0),
         (24, 6272, None, 0.0), (24, 13376, None, 3.0), (24, 13376, 2048, 0.0), (24, 50240, None, 3.0), (12, 50240, 50304, 4.0)]
for H, S, Sq, kb in cases:
    Qh, Kh, Vt = mk(H, S, Sq, S + H)
    ref = run(0, Qh, Kh, Vt, S, kb, Sq).clone()
    got = run(64, Qh, Kh, Vt, S, kb, Sq).clone()
    got2 = run(64, Qh, Kh, Vt, S, kb, Sq).clone()
    nd = int((got.view(torch.int16) != ref.view(torch.int16)).sum()); nr = int((got.view(torch.int16) != got2.view(torch.int16)).sum())
    print("H=%2d S=%6d Sq=%s kb=%g: differing %d of %d, rerun differing %d" % (H, S, Sq, kb, nd, got.numel(), nr), flush=True)
    bad += (nd != 0) + (nr != 0)
print("BIT IDENTITY %s (%d bad)" % ("OK" if bad == 0 else "FAILED", bad), flush=True)

for S in (13376, 50240):
    H = 24
    Qh, Kh, Vt = mk(H, S, None, S)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    fl = 4.0 * S * S * 128 * H
    arms = [0, 32, 64]
    times = {a: [] for a in arms}
    for _ in range(7):
        for a in arms:
            run(a, Qh, Kh, Vt, S, 3.0, None, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(3):
                ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out, key_bias_log2=3.0)
            e1.record(); torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / 3)
    base = sorted(times[0])[3]
    for a in arms:
        t = sorted(times[a]); med = t[len(t) // 2]
        print("S=%6d epilogue arm %2d (%s): med %8.3f ms best %8.3f -> %7.1f TF/s  x%.4f" % (S, a, {0: "product stores", 32: "no stores", 64: "through LDS"}[a], med, t[0], fl / med / 1e9, base / med), flush=True)
_lib.set_option("UTX_ATTN_DEBUG", 0)
sys.exit(1 if bad else 0)
