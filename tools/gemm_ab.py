#!/usr/bin/env python
"""Same-process, interleaved A/B of the GEMM kernels on the FLUX shapes: persistent (UTX_GEMM_TILE=0), per-tile 8-phase (256),
and the vendor library through torch.  Boxes differ by +-5 % and drift thermally, so only interleaved numbers compare."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
dev = "cuda"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
shapes = [(13824, 9216, 3072, "bias"), (13824, 3072, 3072, "gate"), (13824, 12288, 3072, "gelu"), (13824, 3072, 12288, "gate"),
          (13824, 21504, 3072, "bias"), (13824, 3072, 15360, "gate"), (50688, 9216, 3072, "bias"), (50688, 3072, 3072, "gate"),
          (50688, 12288, 3072, "gelu"), (50688, 3072, 12288, "gate"), (50688, 21504, 3072, "bias"), (50688, 3072, 15360, "gate")]
if "--quick" in sys.argv:
    shapes = shapes[6:]
for M, N, K, kind in shapes:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(bias=bias)
    if kind == "gate":
        kw.update(gate=torch.randn(N, device=dev).to(torch.bfloat16), res=C)
    if kind == "gelu":
        kw.update(gelu_from=0)
    def run(tile, sched=0):
        _lib.set_option("UTX_GEMM_TILE", tile)
        _lib.set_option("UTX_GEMM_PERS_SCHED", sched)
        ops.gemm(A, B, out=C, **kw)
    fns = {"pers": lambda: run(2560, 1), "pers1": lambda: run(2560, 2), "8ph": lambda: run(256), "lib": lambda: torch.nn.functional.linear(A, B, bias)}
    ts = {k: [] for k in fns}
    for k, f in fns.items():
        f(); f()
    for r in range(7):
        for k, f in fns.items():
            ts[k].append(t1(f))
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print("M=%6d N=%6d K=%6d %-4s | pers %7.3f ms %6.0f TF | pers(sched1) %7.3f ms %6.0f TF | 8ph %7.3f ms %6.0f TF | lib %7.3f ms %6.0f TF | pers/8ph %.3f  sched1/8ph %.3f  sched1/lib %.3f" % (
        M, N, K, kind, med["pers"], fl / med["pers"] / 1e9, med["pers1"], fl / med["pers1"] / 1e9, med["8ph"], fl / med["8ph"] / 1e9, med["lib"], fl / med["lib"] / 1e9,
        med["8ph"] / med["pers"], med["8ph"] / med["pers1"], med["lib"] / med["pers1"]), flush=True)
    _lib.set_option("UTX_GEMM_TILE", 0); _lib.set_option("UTX_GEMM_PERS_SCHED", 0)
