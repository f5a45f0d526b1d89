#!/usr/bin/env python
"""UTX_GEMM_GROUP_M sweep of the default large-M GEMM on the FLUX shapes (tile rows per L2 block of the tile order), same process, interleaved."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
dev = "cuda"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
groups = [int(x) for x in os.environ.get("UTX_SWEEP_GROUPS", "2,4,8,16,32").split(",")]
shapes = [(50240, 9216, 3072, "bias"), (50240, 12288, 3072, "gelu"), (50240, 21504, 3072, "bias"), (50240, 3072, 12288, "gate"), (50240, 3072, 15360, "gate"),
          (13376, 9216, 3072, "bias"), (13376, 12288, 3072, "gelu"), (13376, 21504, 3072, "bias"), (13376, 3072, 12288, "gate"), (13376, 3072, 15360, "gate")]
for M, N, K, kind in shapes:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(bias=bias)
    if kind == "gate": kw.update(gate=torch.randn(N, device=dev).to(torch.bfloat16), res=C)
    if kind == "gelu": kw.update(gelu_from=0)
    def run(g):
        _lib.set_option("UTX_GEMM_GROUP_M", g)
        ops.gemm(A, B, out=C, **kw)
    ts = {g: [] for g in groups}
    for g in groups: run(g); run(g)
    for r in range(int(os.environ.get("UTX_SWEEP_REPS", "7"))):
        for g in groups: ts[g].append(t1(lambda: run(g)))
    med = {g: sorted(v)[len(v) // 2] for g, v in ts.items()}
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%6d %-4s | " % (M, N, K, kind) + "  ".join("gm%-2d %.3f ms %5.0f TF" % (g, med[g], fl / med[g] / 1e9) for g in groups), flush=True)
_lib.set_option("UTX_GEMM_GROUP_M", 0)
