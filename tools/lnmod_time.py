#!/usr/bin/env python
"""ln_mod at the bench shape: time and effective HBM bandwidth (read x + write y)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
S, D = 50240, 3072
x = torch.randn(S, D, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
sh = torch.randn(D, device="cuda").to(torch.bfloat16); sc = torch.randn(D, device="cuda").to(torch.bfloat16)
f = lambda: ops.ln_mod(x, sh, sc, out=y)
for _ in range(5): f()
torch.cuda.synchronize(); ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); ms = ts[len(ts) // 2]
print("ln_mod S=%d D=%d: %.3f ms, %.2f TB/s" % (S, D, ms, 2.0 * S * D * 2 / (ms * 1e-3) / 1e12))
