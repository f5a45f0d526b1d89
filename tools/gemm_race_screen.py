#!/usr/bin/env python
"""Race screen + A/B timing for the GEMM kernels: the 8-phase 256^2 kernel, the 2-barrier 256^2 kernel and the
128^2 kernel accumulate every output in the same K order with the same MFMA, so their results must be
BIT-IDENTICAL.  Runs each shape several times on fresh random operands and compares exactly."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib  # noqa: E402
from unitex_amd.flux import ops  # noqa: E402

dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 256, 3072), (2048, 3072, 3072), (13824, 3072, 3072),
          (13824, 9216, 3072), (13824, 3072, 15360), (50688, 3072, 3072), (50688, 12288, 3072), (50688, 3072, 12288)]
bad = 0
for (M, N, K) in shapes:
    for r in range(reps):
        g = torch.Generator(device=dev).manual_seed(1000 * r + M % 977 + N + K)
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        bias = (torch.rand(N, device=dev, generator=g) - 0.5).to(torch.bfloat16)
        outs = {}
        for tile in ("128", "2562", "256"):
            _lib.set_option("UTX_GEMM_TILE", int(tile))
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm(A, B, bias=bias, out=C)
            outs[tile] = C
        torch.cuda.synchronize()
        for tile in ("2562", "256"):
            if not torch.equal(outs[tile].view(torch.int16), outs["128"].view(torch.int16)):
                d = (outs[tile].float() - outs["128"].float()).abs()
                nbad = int((d > 0).sum())
                rows = torch.nonzero((d > 0).any(1)).flatten()[:8].tolist()
                print("MISMATCH tile=%s M=%d N=%d K=%d rep=%d: %d elems, max %.4g, rows %s" % (tile, M, N, K, r, nbad, float(d.max()), rows))
                bad += 1
    print("shape M=%6d N=%6d K=%6d : %d reps screened" % (M, N, K, reps), flush=True)
print("RACE SCREEN:", "FAILED (%d)" % bad if bad else "clean")

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

for (M, N, K) in [(13824, 3072, 3072), (13824, 9216, 3072), (13824, 12288, 3072), (13824, 3072, 12288), (13824, 3072, 15360),
                  (50688, 3072, 3072), (50688, 9216, 3072), (50688, 12288, 3072), (50688, 3072, 12288), (50688, 3072, 15360), (8192, 8192, 8192)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = "gemm M=%6d N=%6d K=%6d :" % (M, N, K)
    for tile in ("128", "2562", "256"):
        _lib.set_option("UTX_GEMM_TILE", int(tile))
        ms = timeit(lambda: ops.gemm(A, B, out=C))
        line += "  %s %7.3f ms %7.1f TF/s" % (tile, ms, 2.0 * M * N * K / ms / 1e9)
    print(line, flush=True)
