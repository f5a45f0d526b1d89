#!/usr/bin/env python
"""Race / determinism stress: the pipelined kernels (8-phase GEMM with DMA in flight across barriers, LDS-DMA attention)
must return bit-identical results run after run on the same inputs, also while another stream keeps the chip busy.
usage: python tools/stress_determinism.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
# background load on a second stream (changes timing / memory pressure between repetitions)
side = torch.cuda.Stream()
junk_a = torch.randn(8192, 8192, device=dev, dtype=BF); junk_b = torch.randn(8192, 8192, device=dev, dtype=BF)
def noise(i):
    if i % 3 == 0:
        with torch.cuda.stream(side):
            for _ in range(1 + i % 4):
                torch.mm(junk_a, junk_b)
for (M, N, K, K2) in [(13824, 3072, 3072, 0), (13824, 9216, 3072, 192), (50688, 3072, 15360, 0), (2048, 3072, 12288, 64)]:
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(BF)
    B = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(BF)
    bias = (torch.rand(N, device=dev, generator=g) - 0.5).to(BF)
    kw = {}
    if K2:
        kw = dict(A2=(torch.rand(M, K2, device=dev, generator=g) - 0.5).to(BF), B2=(torch.rand(N, K2, device=dev, generator=g) - 0.5).to(BF))
    ref = ops.gemm(A, B, bias=bias, **kw).clone()
    for i in range(reps):
        noise(i)
        out = ops.gemm(A, B, bias=bias, **kw)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            print("GEMM MISMATCH M=%d N=%d K=%d K2=%d rep %d: %d elems" % (M, N, K, K2, i, int((out != ref).sum())))
    torch.cuda.synchronize()
    print("gemm M=%6d N=%6d K=%6d K2=%3d : %d repetitions bit-identical" % (M, N, K, K2, reps), flush=True)
for (H, S) in [(24, 13824), (8, 50688), (3, 1000)]:
    S_pad = (S + 63) // 64 * 64
    q = torch.zeros(H, S_pad, 128, device=dev, dtype=BF); k = torch.zeros_like(q); vt = torch.zeros(H, 128, S_pad, device=dev, dtype=BF)
    q[:, :S] = (torch.randn(H, S, 128, device=dev, generator=g) * 0.1275).to(BF)
    k[:, :S] = torch.randn(H, S, 128, device=dev, generator=g).to(BF)
    vt[:, :, :S] = torch.randn(H, 128, S, device=dev, generator=g).to(BF)
    k[:, S // 2] = (q[:, 7].float() * 20).to(BF)       # forces the slow (re-centre) path mid-sequence
    ref = ops.attention(q, k, vt, S=S, scale=0.0).clone()
    for i in range(reps):
        noise(i)
        out = ops.attention(q, k, vt, S=S, scale=0.0)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            print("ATTENTION MISMATCH H=%d S=%d rep %d: %d elems" % (H, S, i, int((out != ref).sum())))
    torch.cuda.synchronize()
    print("attention H=%2d S=%6d : %d repetitions bit-identical" % (H, S, reps), flush=True)
print("DETERMINISM STRESS:", "FAILED (%d)" % bad if bad else "clean")
sys.exit(1 if bad else 0)
