#!/usr/bin/env python
"""Where does a peeled attention variant differ from the default kernel?  (round 5: tests/test_attention_peel_gpu.py case H=24, S=3000, spike)"""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_attention_peel_gpu import _inputs

H, S = 24, 3000
Qh, Kh, Vt = _inputs(H, S, seed=S + 7 * H, spike=True)
def run(peel, tailsplit=None):
    _lib.set_option("UTX_ATTN_PEEL", peel)
    o = ops.attention(Qh, Kh, Vt, S=S, scale=0.0)
    torch.cuda.synchronize()
    _lib.set_option("UTX_ATTN_PEEL", 0)
    return o
ref = run(0)
ref2 = run(0)
print("default vs default equal:", torch.equal(ref.view(torch.int16), ref2.view(torch.int16)))
for peel in (2, 6):
    out = run(peel)
    d = (out.view(torch.int16) != ref.view(torch.int16)).view(S, H, 128)
    idx = d.nonzero()
    print("peel", peel, "differing elements", idx.shape[0])
    rows_heads = sorted(set((int(r), int(h)) for r, h, _ in idx.tolist()))
    print("  (row, head):", rows_heads[:40])
    for r, h in rows_heads[:6]:
        cols = [int(c) for rr, hh, c in idx.tolist() if rr == r and hh == h][:6]
        print("   row", r, "head", h, "cols", cols, "ref", ref.view(S, H, 128)[r, h, cols].float().tolist(), "out", out.view(S, H, 128)[r, h, cols].float().tolist())
for ts in (0,):
    _lib.set_option("UTX_ATTN_TAILSPLIT", ts)
    a = run(0); b = run(6)
    print("tailsplit", ts, "default vs peel6 equal:", torch.equal(a.view(torch.int16), b.view(torch.int16)), " default(tailsplit=0) vs default(tailsplit=1) differing:", int((a.view(torch.int16) != ref.view(torch.int16)).sum()))
    _lib.set_option("UTX_ATTN_TAILSPLIT", 1)
