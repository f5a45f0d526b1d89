#!/usr/bin/env python
"""In-kernel timeline of the attention kernel's workgroups (ablation library, UTX_ATTN_VAR=11: correct results + 5 wall-clock stamps per workgroup).

Question: at the reference's own shape (S = 13 376: 1272 workgroups of 209 key tiles) the kernel runs 7 % under its rate at S = 50 240.
How much of a workgroup's life is fixed cost a PERSISTENT workgroup (next item's Q prefetched under the last key tiles, output stores drained under the
next item's first tiles) could hide?  Per workgroup: start -> Q fragments loaded -> ring filled (first barrier) -> key loop done -> stores retired, and per
CU slot the gap between a workgroup's end and its successor's start.  Two workgroups share a CU (4 waves per SIMD), so a workgroup's prologue / epilogue
already overlaps its partner's key loop: what a persistent form can win is bounded by the SUM of these phases, and in practice by a fraction of it.
usage: python tools/attn_timeline.py [S ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib  # noqa: E402

_lib.use_ablation_library()
from unitex_amd._lib import ptr  # noqa: E402
from unitex_amd.flux import ops  # noqa: E402

BF = torch.bfloat16
H = 24
ctx = ops.get_ctx(0)
for S in [int(a) for a in sys.argv[1:]] or [13376, 50240]:
    q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(BF)
    k = torch.randn(H, S, 128, device="cuda").to(BF)
    vt = torch.randn(H, 128, S, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    nwg = ((S + 255) // 256) * H
    tr = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")

    def run(var, work=None):
        _lib.set_option("UTX_ATTN_VAR", var)
        rc = ctx.lib.utx_attn_fwd_bf16_ws(ctx.handle, ptr(q), ptr(k), ptr(vt), ptr(out), q.stride(0), q.stride(1), k.stride(0), k.stride(1), vt.stride(0), vt.stride(1),
                                          out.stride(0), H, S, S, 0.0, 0.0, 0, ptr(work), 0 if work is None else work.numel() * 8, ctx.stream())
        ctx.check(rc)
    for _ in range(3):
        run(11, tr)
    torch.cuda.synchronize()
    # timing with and without the stamps (unsplit launches both: the stamps' own cost)
    ts = {}
    for var, w in ((0, None), (11, tr)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run(var, w)
        b.record(); torch.cuda.synchronize()
        ts[var] = a.elapsed_time(b) / 5
    t = tr.cpu().numpy().astype("int64")
    t0 = t[:, 0].min()
    us = lambda x: x / 100.0            # 100 MHz wall clock
    start, qld, ring, loop, end = [us(t[:, i] - t0) for i in range(5)]
    hw, xcc = t[:, 5] & 0xffffffff, t[:, 5] >> 32
    cu = ((xcc & 0xf) << 12) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xf)      # (xcc, se, sh, cu)
    import numpy as np
    pro_q, pro_ring, body, epi = qld - start, ring - qld, loop - ring, end - loop
    total = end - start
    print("S = %d: %d workgroups x %d key tiles; call %.3f ms plain (unsplit) / %.3f ms stamped; distinct CUs seen %d" % (S, nwg, int(t[0, 7]), ts[0], ts[11], len(set(cu.tolist()))))
    for name, x in (("start -> Q loaded", pro_q), ("Q loaded -> ring filled", pro_ring), ("key loop", body), ("loop end -> stores retired", epi), ("whole workgroup", total)):
        print("   %-28s median %8.2f us   p10 %8.2f   p90 %8.2f   mean %8.2f" % (name, np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.mean()))
    print("   per key tile (loop / tiles): median %.3f us" % np.median(body / t[:, 7]))
    # per CU: occupancy over time -- how long a CU ran 2, 1, 0 workgroups between the call's first start and last end
    span = end.max()
    occ2 = occ1 = occ0 = 0.0
    gaps = []
    for c in set(cu.tolist()):
        m = cu == c
        ev = sorted([(s_, 1) for s_ in start[m]] + [(e_, -1) for e_ in end[m]])
        cur, last = 0, 0.0
        for tt, d in ev:
            dur = tt - last
            if cur >= 2: occ2 += dur
            elif cur == 1: occ1 += dur
            else: occ0 += dur
            cur += d; last = tt
        occ0 += span - last
        # successor gaps: for every end, the next start on this CU
        ss = np.sort(start[m])
        for e_ in end[m]:
            j = np.searchsorted(ss, e_ - 1e-9)
            if j < len(ss):
                gaps.append(ss[j] - e_)
    ncu = len(set(cu.tolist()))
    tot = span * ncu
    print("   CU-time with 2 / 1 / 0 resident workgroups: %.1f %% / %.1f %% / %.1f %%  (span %.1f us)" % (100 * occ2 / tot, 100 * occ1 / tot, 100 * occ0 / tot, span))
    print("   end -> next start on the same CU: median %.2f us, p90 %.2f" % (np.median(gaps), np.percentile(gaps, 90)))
    fixed = np.median(pro_q) + np.median(pro_ring) + np.median(epi) + max(0.0, float(np.median(gaps)))
    print("   fixed phases per workgroup (Q + ring fill + store tail + successor gap): %.2f us = %.2f %% of a workgroup's %.1f us" % (fixed, 100 * fixed / np.median(total), np.median(total)), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
