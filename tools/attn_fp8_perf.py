#!/usr/bin/env python
"""MX fp8 attention (utx_attn_fwd_fp8) vs the bf16 kernel, same process, interleaved, random operands: per-call time and TFLOP/s at the two operating points;
the three quantiser passes (Q, K, V^T) timed beside."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
BF = torch.bfloat16
H = 24
def t_ms(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for S in (13376, 50240):
    q = (torch.randn(H, S, 128, device="cuda") * 0.19).to(BF); k = (torch.randn(H, S, 128, device="cuda") * 1.5).to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    q8, qs = ops.quant_qk_mx8(q); k8, ks = ops.quant_qk_mx8(k); v8, vs = ops.quant_vt_mx8(vt)
    f16 = lambda: ops.attention(q, k, vt, S=S, scale=0.0, out=out)
    f8 = lambda: ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, out=out)
    fq = lambda: (ops.quant_qk_mx8(q, out=(q8, qs)), ops.quant_qk_mx8(k, out=(k8, ks)), ops.quant_vt_mx8(vt, out=(v8, vs)))
    n = 12 if S < 20000 else 4
    for _ in range(2):
        f16(); f8(); fq()
    torch.cuda.synchronize()
    r = {"bf16": [], "fp8": [], "quant": []}
    for _ in range(5):
        r["bf16"].append(t_ms(f16, n)); r["fp8"].append(t_ms(f8, n)); r["quant"].append(t_ms(fq, n))
    med = {k_: sorted(v)[len(v) // 2] for k_, v in r.items()}
    fl = 4.0 * S * S * 128 * H / 1e12
    print("S = %6d: bf16 %.3f ms (%.0f TF/s) | MX fp8 %.3f ms (%.0f TF/s) = %.2fx | Q, K, V^T quantisers %.3f ms -> fp8 incl. quantisers %.2fx" % (
        S, med["bf16"], fl / med["bf16"] * 1e3, med["fp8"], fl / med["fp8"] * 1e3, med["bf16"] / med["fp8"], med["quant"], med["bf16"] / (med["fp8"] + med["quant"])), flush=True)
