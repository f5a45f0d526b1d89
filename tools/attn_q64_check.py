#!/usr/bin/env python
"""The 4 x 64 attention kernel (UTX_ATTN_Q64=1, attention_q64.hip: generated stream) against the default 8 x 32 fast loop, same process:
bit identity over a sweep of shapes (tile counts 1..5 and beyond, key multiplicity, Sq < S, the key-split tail round), then interleaved timing.
UTX_CHECK_SIZES=13376,50240  UTX_CHECK_QUICK=1 (skip the sweep)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
ARMS = [int(x) for x in os.environ.get("UTX_CHECK_ARMS", "").split(",") if x]      # arms of attention_q64_asm_var.inc (ablation library; UTX_ATTN_VAR), 0 = the product stream
if ARMS:
    _lib.use_ablation_library()
from unitex_amd.flux import ops
BF = torch.bfloat16
NAMES = {0: "q64"}
if ARMS:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gen_attn_q64
    NAMES.update({i + 1: c.name for i, c in enumerate(gen_attn_q64.VARIANTS)})


def mk(H, S, seed, peaked=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0)) * (3.0 if peaked else 1.0)).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    return Qh, Kh, Vt


def run(q64, Qh, Kh, Vt, S, kb=0.0, Sq=None, arm=0):
    _lib.set_option("UTX_ATTN_Q64", 1 if q64 else 0)
    if ARMS:
        _lib.set_option("UTX_ATTN_VAR", arm if q64 else 0)
    out = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=Sq)
    torch.cuda.synchronize()
    return out


bad = 0
if os.environ.get("UTX_CHECK_QUICK", "0") != "1":
    cases = []
    for S in (64, 128, 192, 256, 320, 384, 448, 1024, 1536, 4096):
        for H in (1, 3):
            cases.append((H, S, 0.0, None, False))
    cases += [(2, 512, 3.0, None, False), (2, 1024, 3.0, 300, False), (3, 2048, 0.0, 1000, False), (24, 13376, 3.0, None, False), (24, 13376, 0.0, 2048, False),
              (24, 6272, 0.0, None, False), (24, 3328, 3.0, None, False), (4, 8192, 0.0, None, True), (24, 50240, 3.0, None, False)]
    for arm in ([a for a in ARMS if not NAMES[a].startswith("abl_")] or [0]):
      for H, S, kb, Sq, peaked in cases:
        Qh, Kh, Vt = mk(H, S, S + H, peaked)
        ref = run(False, Qh, Kh, Vt, S, kb, Sq)
        got = run(True, Qh, Kh, Vt, S, kb, Sq, arm)
        got2 = run(True, Qh, Kh, Vt, S, kb, Sq, arm)
        nd = int((got.view(torch.int16) != ref.view(torch.int16)).sum())
        nr = int((got.view(torch.int16) != got2.view(torch.int16)).sum())
        md = (got.float() - ref.float()).abs().max().item()
        fin = bool(torch.isfinite(got.float()).all())
        if nd or nr or not fin or not ARMS:
            print("%-12s H=%2d S=%6d kb=%g Sq=%s peaked=%d: differing %d of %d (max|d| %.3g), rerun differing %d, finite %s" % (NAMES[arm], H, S, kb, Sq, peaked, nd, got.numel(), md, nr, fin), flush=True)
        bad += ((nd != 0) and not peaked) + (nr != 0) + (not fin)
      print("arm %s: sweep done" % NAMES[arm], flush=True)
    print("SWEEP %s (%d bad cases; the peaked case differs by design: the 8 x 32 kernel re-centres there)" % ("OK" if bad == 0 else "FAILED", bad), flush=True)

for S in tuple(int(x) for x in os.environ.get("UTX_CHECK_SIZES", "13376,50240").split(",")):
    H = 24
    Qh, Kh, Vt = mk(H, S, S)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    fl = 4.0 * S * S * 128 * H
    arms = [-1] + (ARMS or [0])
    times = {a: [] for a in arms}
    for _ in range(5):
        for a in arms:
            _lib.set_option("UTX_ATTN_Q64", 0 if a < 0 else 1)
            if ARMS:
                _lib.set_option("UTX_ATTN_VAR", max(a, 0))
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out, key_bias_log2=3.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(3):
                ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out, key_bias_log2=3.0)
            e1.record(); torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / 3)
    for a in arms:
        t = sorted(times[a]); med = t[len(t) // 2]
        print("S=%6d %-14s med %8.3f ms best %8.3f -> %7.1f TF/s" % (S, "fast8x32" if a < 0 else NAMES[a], med, t[0], fl / med / 1e9), flush=True)
_lib.set_option("UTX_ATTN_Q64", 0)
if ARMS:
    _lib.set_option("UTX_ATTN_VAR", 0)
sys.exit(1 if bad else 0)
