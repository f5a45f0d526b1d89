#!/usr/bin/env python
"""Is the bf16 attention kernel reproducible on FIXED operands?  Operands = what the fp8 full-width plan feeds its double block's attention (taken from the
workspaces after one forward: Qh / Kh / Vt of the LAST block are there; any fixed operands of that shape do).  Thousands of launches, compared with the first;
arms: plain back to back, with a cache-evicting fill between, with a concurrent GEMM stream beside.  usage: attn_repeat.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
from unitex_amd._lib import ptr
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
H, S = 24, 9280
g = torch.Generator(device="cuda").manual_seed(3)
q = (torch.randn(H, S, 128, device="cuda", generator=g) * 0.2).to(BF)
k = (torch.randn(H, S, 128, device="cuda", generator=g) * 1.5).to(BF)
vt = torch.randn(H, 128, S, device="cuda", generator=g).to(BF)
k[:, :64] = k[:, :1]                      # 64 identical text keys, weighted 8-fold (key_bias_log2 = 3), as the de-duplicated text tokens
vt[:, :, :64] = vt[:, :, :1]
ctx = ops.get_ctx(0)
out = torch.empty(S, H * 128, dtype=BF, device="cuda")
need = int(ctx.lib.utx_attn_workspace_bytes(ctx.handle, H, S, S))
wk = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
junk = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
ga = torch.randn(4096, 4096, device="cuda", dtype=BF); gb = torch.randn(4096, 4096, device="cuda", dtype=BF)
def run():
    rc = ctx.lib.utx_attn_fwd_bf16_ws(ctx.handle, ptr(q), ptr(k), ptr(vt), ptr(out), q.stride(0), q.stride(1), k.stride(0), k.stride(1), vt.stride(0), vt.stride(1),
                                      out.stride(0), H, S, S, 0.0, 3.0, 0, ptr(wk), wk.numel(), ctx.stream())
    ctx.check(rc)
run(); torch.cuda.synchronize()
ref = out.clone()
for arm in ("back to back", "fill between", "gemm stream beside", "unsplit (no scratch)"):
    bad = []
    for i in range(reps):
        if arm == "fill between" and i % 3 == 1:
            junk.fill_(i & 255)
        if arm == "gemm stream beside" and i % 2 == 0:
            with torch.cuda.stream(side):
                torch.mm(ga, gb)
        if arm == "unsplit (no scratch)":
            rc = ctx.lib.utx_attn_fwd_bf16_ws(ctx.handle, ptr(q), ptr(k), ptr(vt), ptr(out), q.stride(0), q.stride(1), k.stride(0), k.stride(1), vt.stride(0), vt.stride(1),
                                              out.stride(0), H, S, S, 0.0, 3.0, 0, None, 0, ctx.stream())
            ctx.check(rc)
            if i == 0:
                torch.cuda.synchronize(); ref = out.clone(); continue
        else:
            run()
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            d = (out != ref)
            rows = torch.nonzero(d.any(1)).flatten(); cols = torch.nonzero(d.any(0)).flatten()
            bad.append((i, int(d.sum()), rows[:4].tolist(), int(cols.min()) // 128, float((out.float() - ref.float()).abs().max())))
    torch.cuda.synchronize()
    print("%-22s: %d of %d launches differ from the first %s" % (arm, len(bad), reps, bad[:6]), flush=True)
