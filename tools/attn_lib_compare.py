#!/usr/bin/env python
"""fast vs general attention loop of ONE library (UTX_LIB) on seeded operands, against a dense fp64 evaluation: which of the two is off, and by how much?  (round 5: a library built
with -mllvm -amdgpu-sched-strategy=max-memory-clause showed fast != general.)"""
import math, os, sys, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
if os.environ.get("UTX_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UTX_LIB"])
from unitex_amd.flux import ops
BF = torch.bfloat16
for H, S, kb in ((4, 3000, 0.0), (4, 2048, 3.0), (24, 13376, 3.0)):
    g = torch.Generator(device="cuda").manual_seed(S)
    S_pad = (S + 63) // 64 * 64
    Qh = (torch.randn(H, S_pad, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Kh = torch.randn(H, S_pad, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S_pad, generator=g, device="cuda").to(BF)
    outs = {}
    for peel in (0, 1):
        _lib.set_option("UTX_ATTN_PEEL", peel)
        o = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb).clone()
        torch.cuda.synchronize()
        o2 = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb).clone()
        torch.cuda.synchronize()
        outs[peel] = o
        print("H %d S %d kb %.0f peel %d: sha %s  repeat equal %s" % (H, S, kb, peel, hashlib.sha1(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12], bool(torch.equal(o.view(torch.int16), o2.view(torch.int16)))), flush=True)
    _lib.set_option("UTX_ATTN_PEEL", 1)
    d = (outs[0].view(torch.int16) != outs[1].view(torch.int16))
    print("   fast vs general: %d differing elements, max |d| %.3g" % (int(d.sum()), float((outs[0].float() - outs[1].float()).abs().max())), flush=True)
    if S <= 4096:
        s = Qh[:, :S].double() @ Kh[:, :S].double().transpose(1, 2)
        if kb:
            s[:, :, :64] += kb
        p = torch.exp2(s - s.max(-1, keepdim=True).values)
        ref = ((p @ Vt[:, :, :S].double().transpose(1, 2)) / p.sum(-1, keepdim=True)).float()      # [H, S, 128]
        for peel in (0, 1):
            o = outs[peel].view(S, H, 128).permute(1, 0, 2).float()
            print("   peel %d vs fp64: max |d| %.4g mean |d| %.4g" % (peel, float((o - ref).abs().max()), float((o - ref).abs().mean())), flush=True)
        if int(d.sum()):
            idx = d.view(S, H, 128).nonzero()[:8].tolist()
            print("   first differing (row, head, col):", idx, flush=True)
