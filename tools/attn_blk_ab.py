#!/usr/bin/env python
"""A/B in isolation: attention on block-strided operands (utx_attn_fwd_bf16_blk: the sequence-parallel receive buffer) against the contiguous call on the same
data, same process, interleaved.  Shapes: the per-group launches of the 2 / 4 / 8-rank jobs (Hg heads x the full sequence) and the 1-rank self-test."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux import ops
BF = torch.bfloat16
ctx = ops.get_ctx(0); lib = ctx.lib
for (P, H, S_loc) in ((1, 6, 50240), (2, 3, 25152), (4, 2, 12608), (8, 3, 6336)):
    S, E = P * S_loc, S_loc * 128
    g = torch.Generator(device="cuda").manual_seed(1)
    q = (torch.randn(H, S, 128, device="cuda", generator=g) * 0.1275).to(BF); k = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    vt = torch.randn(H, 128, S, device="cuda", generator=g).to(BF)
    buf = torch.empty(P, 3, H, E, dtype=BF, device="cuda")
    buf[:, 0] = q.view(H, P, E).transpose(0, 1); buf[:, 1] = k.view(H, P, E).transpose(0, 1)
    buf[:, 2] = vt.view(H, 128, P, S_loc).permute(2, 0, 1, 3).reshape(P, H, E)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda"); out2 = torch.empty_like(out)
    nbytes = int(lib.utx_attn_workspace_bytes(ctx.handle, H, S, S)); work = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    bs = 3 * H * E
    def contiguous():
        rc = lib.utx_attn_fwd_bf16_ws(ctx.handle, C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(vt.data_ptr()), C.c_void_p(out.data_ptr()), S * 128, 128, S * 128, 128,
                                      128 * S, S, out.stride(0), H, S, S, 0.0, 0.0, 0, C.c_void_p(work.data_ptr()), nbytes, ctx.stream()); assert rc == 0
    def blocked():
        rc = lib.utx_attn_fwd_bf16_blk(ctx.handle, C.c_void_p(buf[0, 0].data_ptr()), C.c_void_p(buf[0, 1].data_ptr()), C.c_void_p(buf[0, 2].data_ptr()), C.c_void_p(out2.data_ptr()),
                                       E, 128, E, 128, E, S_loc, out2.stride(0), H, S, S, 0.0, 0.0, 0, C.c_void_p(work.data_ptr()), nbytes, S_loc, bs, bs, bs, ctx.stream()); assert rc == 0
    for f in (contiguous, blocked):
        f(); f()
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    tot = {"contiguous": 0.0, "blocked": 0.0}
    reps = 8
    for _ in range(reps):
        for name, f in (("contiguous", contiguous), ("blocked", blocked)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); f(); f(); b.record(); torch.cuda.synchronize(); tot[name] += a.elapsed_time(b) / 3
    print("P = %d  H = %d  S_loc = %5d : contiguous %.3f ms   block-strided %.3f ms   (%+.2f %%)" % (P, H, S_loc, tot["contiguous"] / reps, tot["blocked"] / reps,
          100.0 * (tot["blocked"] - tot["contiguous"]) / tot["contiguous"]), flush=True)
