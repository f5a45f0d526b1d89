#!/usr/bin/env python
"""What the 16x16x32 MFMA shape would buy gemm256_w4_kernel on the FLUX shapes: ablation build, ABL 64 = every 32x32x16 replaced by two
16x16x32 on the same operands (wrong results, same FLOPs / register / LDS / memory traffic), interleaved with the product kernels."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
dev = "cuda"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
shapes = [(50688, 21504, 3072), (50688, 3072, 15360), (50688, 9216, 3072), (50688, 12288, 3072), (50688, 3072, 12288), (50688, 3072, 3072), (13824, 21504, 3072), (13824, 9216, 3072)]      # the first three: VERDICT r5 item 4's adoption bar (>= +2 % on each)
for M, N, K in shapes:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    def run(tile, abl):
        _lib.set_option("UTX_GEMM_TILE", tile); _lib.set_option("UTX_GEMM_DEBUG", abl << 5)
        ops.gemm(A, B, out=C, bias=bias)
    fns = {"pers": lambda: run(2560, 0), "w4": lambda: run(2564, 0), "w4/16x16": lambda: run(2564, 64), "w4/stagger40us": lambda: run(2564, 256), "lib": lambda: torch.nn.functional.linear(A, B, bias)}
    ts = {k: [] for k in fns}
    for k, f in fns.items(): f(); f()
    for r in range(7):
        for k, f in fns.items(): ts[k].append(t1(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print("M=%6d N=%6d K=%6d | " % (M, N, K) + " | ".join("%s %.3f" % (k, med[k]) for k in fns) + " | 16x16/w4 x%.3f  w4/pers %.3f  w4/lib %.3f  stagger/w4 %.3f" % (med["w4"] / med["w4/16x16"], med["pers"] / med["w4"], med["lib"] / med["w4"], med["w4"] / med["w4/stagger40us"]), flush=True)
_lib.set_option("UTX_GEMM_TILE", 0); _lib.set_option("UTX_GEMM_DEBUG", 0)
