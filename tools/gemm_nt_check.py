#!/usr/bin/env python
"""Bit comparison of the default large-M GEMM (nontemporal C stores) against the persistent 8-wave kernel (plain stores) on three epilogues: ragged-M bias
+ LoRA, GELU + column split, gated residual in place.  A few seconds."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
dev = "cuda"
_lib.set_option("UTX_GEMM_STREAMK", 0)
g = torch.Generator(device=dev).manual_seed(1)
ok = True
for M, N, K, kind in ((13001, 3072, 1024, "bias"), (12544, 6144, 512, "split"), (12800, 3072, 2048, "gate")):
    A = (torch.randn(M, K, device=dev, generator=g) / 2).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    T = (torch.randn(M, 64, device=dev, generator=g) / 8).to(torch.bfloat16); Bl = (torch.randn(N, 64, device=dev, generator=g) / 4).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16); gate = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    def run(tile):
        _lib.set_option("UTX_GEMM_TILE", tile)
        if kind == "gate":
            r = res.clone(); ops.gemm(A, W, bias=bias, out=r, gate=gate, res=r, A2=T, B2=Bl); return r
        if kind == "split":
            c0 = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev); c1 = torch.empty(M, N - 3072, dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, bias=bias, out=c0, gelu_from=3072, n_split=3072, C1=c1); return torch.cat([c0, c1], 1)
        return ops.gemm(A, W, bias=bias, A2=T, B2=Bl)
    a = run(0); b = run(2560); a2 = run(0)
    torch.cuda.synchronize()
    same = bool(torch.equal(a.view(torch.int16), b.view(torch.int16))) and bool(torch.equal(a.view(torch.int16), a2.view(torch.int16)))
    ok = ok and same
    print(M, N, K, kind, "bit-identical to the 8-wave kernel:", same, flush=True)
_lib.set_option("UTX_GEMM_TILE", 0); _lib.set_option("UTX_GEMM_STREAMK", 1)
print("NT_CHECK", "PASS" if ok else "FAIL")
