#!/usr/bin/env python
"""The large-M GEMM on RANDOM and on ALL-ZERO operands, interleaved in one process (the DVFS give-back check of tools/attn_zero_vs_random.py for the
one-wave-per-SIMD 256 x 256 kernel, bf16 and MX fp8): the same cycles, the clock the board sustains differs with the operand data."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops, mx8

ctx = ops.get_ctx(0)
for (M, N, K) in ((50176, 9216, 3072), (50240, 21504, 3072), (50240, 3072, 15360), (50176, 3072, 12288)):
    Ar = (torch.randn(M, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16); Br = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Az, Bz = torch.zeros_like(Ar), torch.zeros_like(Br)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16); C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    arms = {"bf16 random": lambda: ops.gemm(Ar, Br, bias=bias, out=C), "bf16 zeros": lambda: ops.gemm(Az, Bz, bias=bias, out=C)}
    if M % 128 == 0 or True:
        wq, wp = mx8.quantize_weight(Br, ctx, packed=True); aq, ap = mx8.quantize_act(Ar, ctx, packed=True)
        wqz, wpz = mx8.quantize_weight(Bz, ctx, packed=True); aqz, apz = mx8.quantize_act(Az, ctx, packed=True)
        arms["mx8 random"] = lambda: ops.gemm(aq, wq, bias=bias, out=C, a_scale=ap, b_scale=wp)
        arms["mx8 zeros"] = lambda: ops.gemm(aqz, wqz, bias=bias, out=C, a_scale=apz, b_scale=wpz)
    for f in arms.values():
        f(); f()
    torch.cuda.synchronize()
    tot = {k: 0.0 for k in arms}
    reps = 5
    for _ in range(reps):
        for k, f in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                f()
            e1.record(); torch.cuda.synchronize()
            tot[k] += e0.elapsed_time(e1) / 4
    fl = 2.0 * M * N * K
    print("M = %d N = %d K = %d : " % (M, N, K) + "   ".join("%s %.3f ms = %.0f TF/s" % (k, v / reps, fl / (v / reps) / 1e9) for k, v in tot.items()), flush=True)
    del Ar, Br, Az, Bz, C
