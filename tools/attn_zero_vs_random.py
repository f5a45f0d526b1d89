#!/usr/bin/env python
"""The attention launch on RANDOM and on ALL-ZERO operands, interleaved in one process (the guide's DVFS give-back check, MI355X_MICROARCH.md "DVFS give-back":
its tuned kernel runs 1 483 TF/s on zeros and 1 247 on random data with the same wave cycles -- the difference is the clock the board sustains).  Same check for
this repo's kernel: if the zero-data figure matches and the random one does not, the gap is power, not issue slots."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops

H = 24
for S in (50240, 13376, 16384):
    dev = "cuda"
    qr = (torch.randn(H, S, 128, device=dev) * 0.1275).to(torch.bfloat16); kr = torch.randn(H, S, 128, device=dev).to(torch.bfloat16)
    vr = torch.randn(H, 128, S, device=dev).to(torch.bfloat16)
    qz, kz, vz = torch.zeros_like(qr), torch.zeros_like(kr), torch.zeros_like(vr)
    out = torch.empty(S, H * 128, dtype=torch.bfloat16, device=dev)
    fl = 4.0 * H * S * S * 128
    for a in ((qr, kr, vr), (qz, kz, vz)):
        for _ in range(2):
            ops.attention(*a, S=S, out=out, scale=0.0)
    torch.cuda.synchronize()
    reps = 6 if S > 20000 else 20
    tot = {"random": 0.0, "zeros": 0.0}
    for r in range(reps):
        for name, a in (("random", (qr, kr, vr)), ("zeros", (qz, kz, vz))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.attention(*a, S=S, out=out, scale=0.0)
            e1.record(); torch.cuda.synchronize()
            tot[name] += e0.elapsed_time(e1) / 3
    ms = {k_: v / reps for k_, v in tot.items()}
    print("S = %6d  H = %d : random %.3f ms = %.0f TF/s   zeros %.3f ms = %.0f TF/s   zeros / random = %.3f   (guide's tuned kernel: 1247 random, 1483 zeros = 1.189)"
          % (S, H, ms["random"], fl / ms["random"] / 1e9, ms["zeros"], fl / ms["zeros"] / 1e9, ms["random"] / ms["zeros"]), flush=True)
