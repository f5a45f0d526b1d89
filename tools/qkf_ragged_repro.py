#!/usr/bin/env python
"""Reproducer of round 3's one-off: the fused q / k epilogue of the one-wave-per-SIMD GEMM on a RAGGED last row of tiles (M % 256 != 0) with the
cos / sin tables COLD.  In the round-3 build the compiler placed copies of the cos / sin registers in front of the ragged branch's `s_waitcnt vmcnt(0)`
(tools/vmcnt_hazard_check.py finds them in the listing): whenever the tables' loads had not landed by then, rows of the ragged tiles were rotated by
stale angles.  Each repetition evicts the tables (a 1 GB fill between launches) and compares Q / K of the ragged rows with GEMM -> utx_qkv_post.
usage: python tools/qkf_ragged_repro.py [reps] [path/to/libunitex_hip.so]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib  # noqa: E402

if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from oracle import dit_ref  # noqa: E402  (rope tables only; tools are not product)
from unitex_amd.flux import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
BF = torch.bfloat16
H, K = 24, 3072
D = H * 128
M, tok_off = 9280, 0          # the single block's projection in the full-width test: 36.25 row tiles -> 64 valid rows in the ragged tiles
g = torch.Generator(device="cuda").manual_seed(5)
x = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
W = (torch.randn(3 * D, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
bias = torch.randn(3 * D, device="cuda", generator=g).to(BF)
wq = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
wk = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
S = tok_off + M
S_pad = (S + 63) // 64 * 64
ids = torch.stack([torch.zeros(S), torch.arange(S) // 97, torch.arange(S) % 97], 1).float()
cos0, sin0 = [t.cuda().contiguous() for t in dit_ref.rope_tables(ids)]
qs = 0.1275
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")


def run(fused, cos, sin):
    qkv = torch.full((M, 3 * D), 3.0, dtype=BF, device="cuda")
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda")
    Kh = torch.zeros_like(Qh)
    Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
    qk = dict(cols=2 * D, tok_off=tok_off, eps=1e-6, q_scale=qs, wq=wq, wk=wk, cos=cos, sin=sin, Qh=Qh, Kh=Kh) if fused else None
    junk.fill_(1)          # evict the tables (and everything else) from the L2s / MALL
    ops.gemm(x, W, bias=bias, out=qkv, qk_post=qk)
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, M, tok_off, H, q_scale=qs, skip_qk=fused)
    torch.cuda.synchronize()
    return Qh, Kh


_lib.set_option("UTX_GEMM_STREAMK", 0)
ref_q, ref_k = run(False, cos0, sin0)
bad = 0
for r in range(reps):
    cos, sin = cos0.clone(), sin0.clone()       # fresh addresses: cold TLB entries as well
    q, k = run(True, cos, sin)
    dq = (q.view(torch.int16) != ref_q.view(torch.int16))
    dk = (k.view(torch.int16) != ref_k.view(torch.int16))
    n = int(dq.sum()) + int(dk.sum())
    if n:
        bad += 1
        rows = torch.nonzero(dq.any(2).any(0) | dk.any(2).any(0)).flatten()
        print("rep %3d: %6d elements differ, token rows %d..%d (%d rows; ragged rows start at %d)" % (r, n, int(rows.min()), int(rows.max()), rows.numel(), M // 256 * 256), flush=True)
print("library %s: %d of %d cold fused launches differ from GEMM -> qkv_post" % (_lib.LIB_PATH, bad, reps))
