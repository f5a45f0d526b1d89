#!/usr/bin/env python
"""gemm256_w4_kernel with its steady-state K loop as the generated stream (UTX_GEMM_FASTK=1, gemm_w4_loop_asm.inc) against hipcc's loop (=0), same process:
bit identity on the epilogue / segment variants (plain + bias, LoRA K-segment + GELU + column split, gated residual, short K, ragged M, split tail round on / off), then
interleaved timing on the FLUX shapes beside hipBLASLt (torch.nn.functional.linear)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
BF = torch.bfloat16
dev = "cuda"


def both(fn):
    outs = {}
    for fk in (0, 1):
        _lib.set_option("UTX_GEMM_FASTK", fk)
        outs[fk] = fn()
        torch.cuda.synchronize()
    return outs


bad = 0
_lib.set_option("UTX_GEMM_TILE", 2564)
for sk in (0, 1):
    _lib.set_option("UTX_GEMM_STREAMK", sk)
    for M in (50688, 13824, 6336, 13001):
        D, R = 3072, 64
        g = torch.Generator(device=dev).manual_seed(M)
        x = (torch.randn(M, D, device=dev, generator=g) / 2).to(BF)
        N = 7 * D
        W = (torch.randn(N, D, device=dev, generator=g) / math.sqrt(D)).to(BF)
        bias = torch.randn(N, device=dev, generator=g).to(BF)
        T = (torch.randn(M, 3 * R, device=dev, generator=g) / 8).to(BF)
        Bl = torch.zeros(N, R, dtype=BF, device=dev)
        Bl[:3 * D] = (torch.randn(3 * D, R, device=dev, generator=g) / 4).to(BF)

        def fused():
            c0 = torch.empty(M, 3 * D, dtype=BF, device=dev); c1 = torch.empty(M, 4 * D, dtype=BF, device=dev)
            ops.gemm(x, W, bias=bias, out=c0, A2=T, B2=Bl, lora_n_limit=3 * D, lora_seg_n=D, gelu_from=3 * D, n_split=3 * D, C1=c1)
            return torch.cat([c0, c1], 1)
        o = both(fused)
        ok1 = torch.equal(o[0].view(torch.int16), o[1].view(torch.int16))
        cat = (torch.randn(M, 5 * D, device=dev, generator=g) / 4).to(BF)
        Wo = (torch.randn(D, 5 * D, device=dev, generator=g) / math.sqrt(5 * D)).to(BF)
        bo = torch.randn(D, device=dev, generator=g).to(BF); gate = torch.randn(D, device=dev, generator=g).to(BF); res = torch.randn(M, D, device=dev, generator=g).to(BF)

        def gated():
            r = res.clone(); ops.gemm(cat, Wo, bias=bo, out=r, gate=gate, res=r); return r
        o = both(gated)
        ok2 = torch.equal(o[0].view(torch.int16), o[1].view(torch.int16))
        oks = []
        for K in (64, 128, 192, 320):      # 1, 2, 3, 5 K-tiles: the stream's entry condition at its edges
            A = (torch.randn(M, K, device=dev, generator=g)).to(BF); Wk = torch.randn(D, K, device=dev, generator=g).to(BF)
            o = both(lambda: ops.gemm(A, Wk, bias=bo))
            oks.append(torch.equal(o[0].view(torch.int16), o[1].view(torch.int16)))
        print("streamk=%d M=%6d: fused qkv|mlp + LoRA %s, gated residual K=15360 %s, K = 64/128/192/320 %s" % (sk, M, ok1, ok2, oks), flush=True)
        bad += (not ok1) + (not ok2) + sum(not v for v in oks)
        del x, W, T, Bl, cat, Wo, res
print("BIT IDENTITY %s (%d mismatches)" % ("OK" if bad == 0 else "FAILED", bad), flush=True)
_lib.set_option("UTX_GEMM_STREAMK", 1)


def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)


for M, N, K in [(50688, 21504, 3072), (50688, 3072, 15360), (50688, 9216, 3072), (50688, 12288, 3072), (50688, 3072, 12288), (50688, 3072, 3072), (13824, 21504, 3072), (13824, 9216, 3072)]:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(BF); B = torch.randn(N, K, device=dev).to(BF)
    bias = torch.randn(N, device=dev).to(BF); C = torch.empty(M, N, dtype=BF, device=dev)

    def run(fk):
        _lib.set_option("UTX_GEMM_FASTK", fk); ops.gemm(A, B, out=C, bias=bias)
    fns = {"w4": lambda: run(0), "w4_fastk": lambda: run(1), "lib": lambda: torch.nn.functional.linear(A, B, bias)}
    ts = {k: [] for k in fns}
    for k, f in fns.items():
        f(); f()
    for r in range(7):
        for k, f in fns.items():
            ts[k].append(t1(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%6d | w4 %.3f ms (%.0f TF/s) | fastk %.3f ms (%.0f TF/s) = x%.3f | hipBLASLt %.3f ms | fastk / vendor %.3f" % (
        M, N, K, med["w4"], fl / med["w4"] / 1e9, med["w4_fastk"], fl / med["w4_fastk"] / 1e9, med["w4"] / med["w4_fastk"], med["lib"], med["lib"] / med["w4_fastk"]), flush=True)
_lib.set_option("UTX_GEMM_TILE", 0); _lib.set_option("UTX_GEMM_FASTK", _lib.get_options().get("UTX_GEMM_FASTK", 0))
sys.exit(1 if bad else 0)
