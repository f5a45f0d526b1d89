#!/usr/bin/env python
"""Timing experiment (ablation build, WRONG results): the attention kernel with every 32x32x16 MFMA replaced by two 16x16x32 MFMAs
(same FLOPs / operands), interleaved with the real kernel, at the bench shapes."""
import math, os, sys, torch
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
    vt = torch.randn(H, 128, S, device="cuda").to(BF); out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    def run(var):
        _lib.set_option("UTX_ATTN_VAR", var); ops.attention(q, k, vt, S=S, out=out, scale=0.0)
    ts = {0: [], 5: []}
    for v in ts: run(v); run(v)
    for r in range(5):
        for v in ts: ts[v].append(t1(lambda: run(v)))
    med = {v: sorted(x)[len(x) // 2] for v, x in ts.items()}
    fl = 4.0 * S * S * 128 * H
    print("attn S=%6d | 32x32x16 %7.3f ms %6.0f TF | 2 x 16x16x32 %7.3f ms %6.0f TF | x%.3f" % (S, med[0], fl / med[0] / 1e9, med[5], fl / med[5] / 1e9, med[0] / med[5]), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
