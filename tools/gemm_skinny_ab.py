#!/usr/bin/env python
"""The LoRA-down products of a denoise step (x . A^T: N = 64 / 192 columns over all rows) on the one-pass streaming kernel (gemm_skinny.hip, UTX_GEMM_SKINNY=1, default)
against the 128 x 128 tile kernel (=0): bits + interleaved timing + the rate of the one pass over x they amount to."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
BF = torch.bfloat16
tot = {0: 0.0, 1: 0.0}
for M, N, K, per_step in ((50240, 192, 3072, 38), (50176, 192, 3072, 19), (50176, 64, 3072, 39), (50176, 64, 12288, 19), (13376, 192, 3072, 57), (13376, 64, 12288, 19)):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn(M, K, generator=g, device="cuda") / math.sqrt(K)).to(BF); B = torch.randn(N, K, generator=g, device="cuda").to(BF)
    T = torch.empty(M, 192, dtype=BF, device="cuda")
    res, t = {}, {0: [], 1: []}
    for sk in (0, 1):
        _lib.set_option("UTX_GEMM_SKINNY", sk)
        ops.gemm(A, B, out=T[:, :N]); torch.cuda.synchronize(); res[sk] = T[:, :N].clone()
    for _ in range(7):
        for sk in (0, 1):
            _lib.set_option("UTX_GEMM_SKINNY", sk)
            ops.gemm(A, B, out=T[:, :N])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _r in range(5):
                ops.gemm(A, B, out=T[:, :N])
            b.record(); torch.cuda.synchronize()
            t[sk].append(a.elapsed_time(b) / 5 * 1e3)
    m = {k: sorted(v)[3] for k, v in t.items()}
    gb = (M * K + N * K + M * N) * 2 / 1e9
    if M > 20000:
        for k in (0, 1):
            tot[k] += m[k] * per_step * 1e-3
    print("M=%6d N=%4d K=%6d: bit-identical %s | tile kernel %7.1f us (%.2f TB/s) | streaming %7.1f us (%.2f TB/s) = x%.2f | %d per step" %
          (M, N, K, bool(torch.equal(res[0].view(torch.int16), res[1].view(torch.int16))), m[0], gb / m[0] * 1e3, m[1], gb / m[1] * 1e3, m[0] / m[1], per_step), flush=True)
print("per step at S = 50 240: tile kernel %.2f ms, streaming kernel %.2f ms" % (tot[0], tot[1]))
_lib.set_option("UTX_GEMM_SKINNY", 1)
