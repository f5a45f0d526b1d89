#!/usr/bin/env python
"""A/B: the LDS-DMA attention kernel's general loop ("default" = UTX_ATTN_PEEL=0), its fast loop ("peel1", the product's default since round 5) and the 4 x 64 kernel (+ repair pass, "q64"),
same process, interleaved; no key multiplicity (the 4 x 64 kernel has none).  UTX_AB_ARMS=default,peel1,q64."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
BF, H = torch.bfloat16, 24
ARMS = [a for a in os.environ.get("UTX_AB_ARMS", "default,peel6,q64").split(",")]
def setarm(a):
    _lib.set_option("UTX_ATTN_PEEL", 0); _lib.set_option("UTX_ATTN_Q64", 0)
    if a.startswith("peel"): _lib.set_option("UTX_ATTN_PEEL", int(a[4:]))
    if a.startswith("q64"): _lib.set_option("UTX_ATTN_Q64", 1)
for S in tuple(int(x) for x in os.environ.get("UTX_AB_SIZES", "13376,50240").split(",")):
    g = torch.Generator(device="cuda").manual_seed(S)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    fl = 4.0 * S * S * 128 * H
    ref = None
    for a in ARMS:
        setarm(a); ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else: print("S=%d %s: max|d| vs %s = %g, differing elements %d" % (S, a, ARMS[0], (out.float() - ref.float()).abs().max().item(), int((out.view(torch.int16) != ref.view(torch.int16)).sum())), flush=True)
    times = {a: [] for a in ARMS}
    for _ in range(5):
        for a in ARMS:
            setarm(a); ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(3): ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out)
            e1.record(); torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / 3)
    for a in ARMS:
        t = sorted(times[a]); med = t[len(t) // 2]
        print("S=%6d %-8s med %8.3f ms best %8.3f -> %7.1f TF/s" % (S, a, med, t[0], fl / med / 1e9), flush=True)
setarm("default")
