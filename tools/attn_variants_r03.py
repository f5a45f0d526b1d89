#!/usr/bin/env python
"""Round 3, VERDICT item 3: the two out-of-mainloop levers the guide names for an 8-wave attention that is already near the best known kernel
(MI355X_MICROARCH.md "Two waves per SIMD" item 4, cdna_hip_programming.md T21), same process, interleaved, correct results in every arm:
  0 = the library default (8-byte epilogue stores, per-tile priority flips)
  6 = STATIC priority: waves 4-7 at s_setprio 1 for the whole kernel, no per-segment flips
  7 = 16-byte epilogue stores through v_permlane32_swap
  8 = nontemporal output stores (second run of the tool: profiles/r03_attn_variants_v1.log; arm 6 dropped)
(the log profiles/r03_attn_variants_v0.log was taken while 16-byte stores were the default and arm 7 the 8-byte form: read its columns that way)
Ablation library only because the variants are template instances selected by UTX_ATTN_VAR."""
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
names = {0: "default (8-B stores)", 8: "nontemporal O stores", 7: "16-B stores (permlane swap)"}
for S in (13376, 50240):
    q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(BF); k = torch.randn(H, S, 128, device="cuda").to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF)
    outs = {}
    def run(var, out):
        _lib.set_option("UTX_ATTN_VAR", var); ops.attention(q, k, vt, S=S, out=out, scale=0.0)
    ts = {v: [] for v in names}
    for v in ts:
        outs[v] = torch.empty(S, H * 128, dtype=BF, device="cuda")
        run(v, outs[v]); run(v, outs[v])
    torch.cuda.synchronize()
    same = {v: bool(torch.equal(outs[v], outs[0])) for v in names}
    for r in range(7):
        for v in ts: ts[v].append(t1(lambda: run(v, outs[v])))
    med = {v: sorted(x)[len(x) // 2] for v, x in ts.items()}
    fl = 4.0 * S * S * 128 * H
    print("attn S=%6d | " % S + " | ".join("%s %7.3f ms %6.0f TF x%.3f bits%s" % (names[v], med[v], fl / med[v] / 1e9, med[0] / med[v], "=" if same[v] else "!=") for v in names), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
