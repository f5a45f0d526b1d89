#!/usr/bin/env python
"""Accuracy of the dot2 row-sum arm of the 4 x 64 attention stream (arm 9 of the ablation library) beside the product stream, both against an fp64 attention over the
same bf16 operands: does summing the bf16 probabilities (what PV consumes) instead of the fp32 ones move the distance to the exact result?"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
BF = torch.bfloat16
ARM = int(os.environ.get("UTX_DOT2_ARM", "9"))


def exact(Qh, Kh, Vt, S, kb):
    H = Qh.shape[0]
    out = torch.empty(S, H * 128, dtype=torch.float64, device="cuda")
    for h in range(H):
        s = Qh[h].double() @ Kh[h].double().T          # log2 units (Q pre-scaled)
        if kb:
            s[:, :64] += kb
        p = torch.exp2(s - s.max(dim=1, keepdim=True).values)
        out[:, 128 * h:128 * (h + 1)] = (p @ Vt[h].double().T) / p.sum(dim=1, keepdim=True)
    return out


for H, S, kb, scale in ((2, 2048, 0.0, 1.0), (2, 4096, 3.0, 1.0), (2, 4096, 0.0, 3.0), (1, 8192, 3.0, 0.3)):
    g = torch.Generator(device="cuda").manual_seed(S + H)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0)) * scale).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    ref = exact(Qh, Kh, Vt, S, kb)
    res = {}
    for name, q64, arm in (("fast8x32", 0, 0), ("q64", 1, 0), ("q64_dot2", 1, ARM)):
        _lib.set_option("UTX_ATTN_Q64", q64); _lib.set_option("UTX_ATTN_VAR", arm)
        o = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb).double()
        torch.cuda.synchronize()
        d = (o - ref).abs()
        res[name] = o
        print("H=%d S=%5d kb=%g scale=%g %-9s max|d| %.3e  mean|d| %.3e  rms %.3e  (max|ref| %.3f)  signed mean rel %.3e" % (H, S, kb, scale, name, d.max().item(), d.mean().item(), d.pow(2).mean().sqrt().item(), ref.abs().max().item(),
              ((o - ref) * ref.sign()).mean().item() / ref.abs().mean().item()), flush=True)
    print("   q64_dot2 vs q64: differing bf16 elements %d of %d, max|d| %.3e" % (int((res["q64_dot2"] != res["q64"]).sum()), ref.numel(), (res["q64_dot2"] - res["q64"]).abs().max().item()), flush=True)
_lib.set_option("UTX_ATTN_Q64", 0); _lib.set_option("UTX_ATTN_VAR", 0)
