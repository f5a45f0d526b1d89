#!/usr/bin/env python
"""MX fp8 GEMMs against the bf16 kernel on the FLUX shapes, same process, interleaved:
  mx8-tiled : 128 x 128 x 128 tiles, row-major scales (round 2)
  mx8-w4    : persistent 256 x 256 tiles, one wave per SIMD, tile-packed scales (gemm_w4.hip, MX; round 3)
the activation quantiser (utx_quant_mx8 / utx_quant_mx8_packed) is timed separately (it runs once per GEMM input in the DiT plan)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops, mx8
dev = "cuda"; ctx = ops.get_ctx(0)
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
shapes = [(13824, 9216, 3072), (13824, 21504, 3072), (13824, 3072, 15360), (50688, 9216, 3072), (50688, 12288, 3072),
          (50688, 3072, 12288), (50688, 21504, 3072), (50688, 3072, 15360), (34304, 21504, 3072), (34304, 3072, 15360)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for M, N, K in shapes:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wq, ws = mx8.quantize_weight(B, ctx); _, wp = mx8.quantize_weight(B, ctx, packed=True)
    aq, as_ = mx8.quantize_act(A, ctx); _, ap = mx8.quantize_act(A, ctx, packed=True)
    fns = {"bf16": lambda: ops.gemm(A, B, bias=bias, out=C),
           "tiled": lambda: ops.gemm(aq, wq, bias=bias, out=C, a_scale=as_, b_scale=ws),
           "w4": lambda: ops.gemm(aq, wq, bias=bias, out=C, a_scale=ap, b_scale=wp),
           "quant": lambda: mx8.quantize_act(A, ctx, out=(aq, as_)),
           "quantp": lambda: mx8.quantize_act(A, ctx, out=(aq, ap))}
    ts = {k: [] for k in fns}
    for f in fns.values():
        f(); f()
    for r in range(5):
        for k, f in fns.items():
            ts[k].append(t1(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%6d | bf16 %7.3f ms %6.0f TF | mx8-tiled %7.3f ms %6.0f TF (x%.2f) | mx8-w4 %7.3f ms %6.0f TF (x%.2f) | quant %6.3f / packed %6.3f ms %5.0f GB/s | "
          "w4+quant vs bf16 x%.2f" % (M, N, K, med["bf16"], fl / med["bf16"] / 1e9, med["tiled"], fl / med["tiled"] / 1e9, med["bf16"] / med["tiled"],
                                      med["w4"], fl / med["w4"] / 1e9, med["bf16"] / med["w4"], med["quant"], med["quantp"],
                                      M * K * 3.03 / med["quantp"] / 1e6, med["bf16"] / (med["w4"] + med["quantp"])), flush=True)
