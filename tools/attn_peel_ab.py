#!/usr/bin/env python
"""A/B of the attention kernels' fast loops (UTX_ATTN_PEEL = 1 / UTX_ATTN8_PEEL = 1, the defaults since round 5) against their general loops (= 0, the defaults until round 4):
same process, interleaved launches, at the two operating points (S = 13 376 and 50 240 executed tokens, 24 heads, pre-scaled Q, key multiplicity 8 on tile 0 as in the step).
Prints bit-identity (first launch + repeated launches) and TF/s per arm.  Round 5's first run, with all six candidate loops: profiles/r05_attn_peel_ab.log."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib            # noqa: E402
if os.environ.get("UTX_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UTX_LIB"])      # a differently built library (tools/build_variant.py)
from unitex_amd.flux import ops        # noqa: E402

BF, H = torch.bfloat16, int(os.environ.get("UTX_AB_HEADS", "24"))
SIZES = tuple(int(x) for x in os.environ.get("UTX_AB_SIZES", "13376,50240").split(","))


def ab(name, option, run, fl, rounds, repeats, out):
    _lib.set_option(option, 0)
    run()
    torch.cuda.synchronize()
    ref = out.clone()
    miss = 0
    _lib.set_option(option, 1)
    for _ in range(repeats):
        out.zero_()
        run()
        torch.cuda.synchronize()
        miss += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
    print("%s %s=1: %d of %d launches differ from the general loop" % (name, option, miss, repeats), flush=True)
    times = {0: [], 1: []}
    for _ in range(rounds):
        for arm in (0, 1):
            _lib.set_option(option, arm)
            run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _r in range(3):
                run()
            b.record()
            torch.cuda.synchronize()
            times[arm].append(a.elapsed_time(b) / 3.0)
    _lib.set_option(option, 1)
    for arm in (0, 1):
        t = sorted(times[arm])
        med = t[len(t) // 2]
        print("%s %s=%d  med %8.3f ms  best %8.3f ms  -> %7.1f TF/s" % (name, option, arm, med, t[0], fl / (med * 1e-3) / 1e12), flush=True)


def main():
    rounds = int(os.environ.get("UTX_AB_ROUNDS", "5"))
    repeats = int(os.environ.get("UTX_AB_REPEATS", "10"))
    for S in SIZES:
        g = torch.Generator(device="cuda").manual_seed(S)
        S_pad = (S + 63) // 64 * 64
        Qh = (torch.randn(H, S_pad, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
        Kh = torch.randn(H, S_pad, 128, generator=g, device="cuda").to(BF)
        Vt = torch.randn(H, 128, S_pad, generator=g, device="cuda").to(BF)
        out = torch.empty(S, H * 128, dtype=BF, device="cuda")
        fl = 4.0 * S * S * 128 * H
        ab("bf16 S = %6d" % S, "UTX_ATTN_PEEL", lambda: ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=3.0, out=out), fl, rounds, repeats, out)
        q8, qs = ops.quant_qk_mx8(Qh)
        k8, ks = ops.quant_qk_mx8(Kh)
        v8, vs = ops.quant_vt_mx8(Vt)
        ab("fp8  S = %6d" % S, "UTX_ATTN8_PEEL", lambda: ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, out=out, key_bias_log2=3.0), fl, rounds, repeats, out)


if __name__ == "__main__":
    main()
