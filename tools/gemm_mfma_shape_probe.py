#!/usr/bin/env python
"""Timing experiment (ablation build, WRONG results): the persistent GEMM with every 32x32x16 MFMA replaced by two 16x16x32 MFMAs --
same FLOPs, operand registers, LDS and DMA traffic -- interleaved with the real kernel.  Decides whether converting the kernels to the
16x16x32 shape (more power-efficient on random data: tools/mfma_power_probe.py) is worth the rewrite."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
dev = "cuda"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
_lib.set_option("UTX_GEMM_TILE", 2560)
for sched in (1,):
    _lib.set_option("UTX_GEMM_PERS_SCHED", sched)
    for M, N, K in [(50688, 9216, 3072), (50688, 21504, 3072), (50688, 3072, 15360)]:
        A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev).to(torch.bfloat16); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        def run(dbg):
            _lib.set_option("UTX_GEMM_DEBUG", dbg); ops.gemm(A, B, bias=bias, out=C)
        ts = {0: [], 16: []}
        for d in ts: run(d); run(d)
        for r in range(7):
            for d in ts: ts[d].append(t1(lambda: run(d)))
        med = {d: sorted(v)[len(v) // 2] for d, v in ts.items()}
        fl = 2.0 * M * N * K
        print("sched %d M=%6d N=%6d K=%6d | 32x32x16 %7.3f ms %6.0f TF | 2 x 16x16x32 %7.3f ms %6.0f TF | x%.3f" % (
            sched, M, N, K, med[0], fl / med[0] / 1e9, med[16], fl / med[16] / 1e9, med[0] / med[16]), flush=True)
_lib.set_option("UTX_GEMM_DEBUG", 0)
