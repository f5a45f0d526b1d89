#!/usr/bin/env python
"""Is the attention kernel limited per CU or by a chip-wide resource (power / clock, L2, fabric)?
Same per-workgroup work (S = 16384 -> 64 workgroups per head), 1..8 heads -> 64..512 workgroups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
BF = torch.bfloat16
S = 16384
def timeit(fn, n=7):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for H in (1, 2, 3, 4, 8, 16):
    q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(BF)
    k = torch.randn(H, S, 128, device="cuda").to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    ms = timeit(lambda: ops.attention(q, k, vt, S=S, out=out, scale=0.0))
    wg = H * (S // 256)
    fl = 4.0 * S * S * 128 * H
    print("H=%2d  workgroups %4d (%.2f rounds of 256 CUs)  %8.3f ms   %7.1f TF/s   per-CU-round %.3f ms" %
          (H, wg, wg / 256.0, ms, fl / ms / 1e9, ms / max(1.0, -(-wg // 256))), flush=True)
