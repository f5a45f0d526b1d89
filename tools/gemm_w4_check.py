#!/usr/bin/env python
"""gemm256_w4_kernel (UTX_GEMM_TILE=2564) against the persistent 8-wave kernel (2560): bit-identity + interleaved timing (+ the vendor
library through torch for scale)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
dev = "cuda"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
shapes = [(512, 256, 64, "bias"), (768, 512, 128, "bias"), (1000, 512, 192, "bias"), (13824, 3072, 3072, "bias"), (13824, 3072, 3072, "gate"),
          (50688, 21504, 3072, "bias"), (50688, 12288, 3072, "gelu"), (50688, 3072, 12288, "gate"), (50688, 9216, 3072, "bias"), (50688, 3072, 3072, "gate")]
if "--small" in sys.argv:
    shapes = shapes[:5]
for M, N, K, kind in shapes:
    torch.manual_seed(M + N + K)
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    kw = dict(bias=bias)
    if kind == "gate":
        kw.update(gate=torch.randn(N, device=dev).to(torch.bfloat16))
    if kind == "gelu":
        kw.update(gelu_from=0)
    outs = {}
    def run(tile, C):
        _lib.set_option("UTX_GEMM_TILE", tile)
        if kind == "gate":
            ops.gemm(A, B, out=C, res=R, **kw)
        else:
            ops.gemm(A, B, out=C, **kw)
    for tile in (2560, 2564):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        run(tile, C); torch.cuda.synchronize()
        outs[tile] = C
    same = torch.equal(outs[2560].view(torch.int16), outs[2564].view(torch.int16))
    nbad = int((outs[2560].view(torch.int16) != outs[2564].view(torch.int16)).sum())
    Ct = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fns = {"pers": lambda: run(2560, Ct), "w4": lambda: run(2564, Ct), "lib": lambda: torch.nn.functional.linear(A, B, bias)}
    ts = {k: [] for k in fns}
    for k, f in fns.items():
        f(); f()
    for r in range(5):
        for k, f in fns.items():
            ts[k].append(t1(f))
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print("M=%6d N=%6d K=%6d %-4s | identical %s (%d differ) | pers %7.3f ms %6.0f TF | w4 %7.3f ms %6.0f TF | lib %7.3f ms %6.0f TF | w4/pers %.3f w4/lib %.3f" % (
        M, N, K, kind, same, nbad, med["pers"], fl / med["pers"] / 1e9, med["w4"], fl / med["w4"] / 1e9, med["lib"], fl / med["lib"] / 1e9,
        med["pers"] / med["w4"], med["lib"] / med["w4"]), flush=True)
_lib.set_option("UTX_GEMM_TILE", 0)
