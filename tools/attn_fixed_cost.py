#!/usr/bin/env python
"""Where the per-workgroup fixed cost of the attention kernel goes (it is 4-5 % of a call at the reference's operating point S = 13 376, 1.4 % at
S = 50 240: T(call) = rounds x (tiles x t_tile + F), F ~ 18 us from the two sequence lengths).  Ablation library, same process, interleaved; arms 9 and 10
give WRONG results on purpose:
  0  = the default kernel
  9  = no output stores                      -> the store tail
  10 = no first-tile max pass (AG_SLOW at t = 0)  -> the prologue's slow path
  7  = 16-byte stores (correct results; for reference)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
BF = torch.bfloat16
H = 24
arms = (0, 9, 10, 7)
for S in (13376, 50240):
    q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(BF); k = torch.randn(H, S, 128, device="cuda").to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF); out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    def run(var):
        _lib.set_option("UTX_ATTN_VAR", var); ops.attention(q, k, vt, S=S, out=out, scale=0.0)
    for v in arms:
        run(v); run(v)
    torch.cuda.synchronize()
    ts = {v: [] for v in arms}
    for r in range(9 if S < 20000 else 5):
        for v in arms:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                run(v)
            b.record(); torch.cuda.synchronize(); ts[v].append(a.elapsed_time(b) / 4)
    med = {v: sorted(x)[len(x) // 2] for v, x in ts.items()}
    nwg = ((S + 255) // 256) * H
    print("S = %6d (%d workgroups, %.2f rounds of 256): " % (S, nwg, nwg / 256.0) + "  ".join("var %d %.4f ms (%+.2f %%)" % (v, med[v], 100.0 * (med[v] - med[0]) / med[0]) for v in arms), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
