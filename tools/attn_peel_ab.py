#!/usr/bin/env python
"""A/B of the opt-in peeled attention loop (UTX_ATTN_PEEL = 1 ... 6: attention_glds.hip VAR 12 ... 17) against the default kernel, same process, interleaved launches,
at the two operating points (S = 13 376 and 50 240 executed tokens, 24 heads, pre-scaled Q, key multiplicity 8 on tile 0 as in the step).  Prints bit-identity
and TF/s per arm.  RUN tests/test_attention_peel_gpu.py FIRST (UTX_RUN_UNVALIDATED=1): these variants had not run on hardware when they were committed.

    UTX_RUN_UNVALIDATED=1 python -m pytest tests/test_attention_peel_gpu.py -m gpu -q && python tools/attn_peel_ab.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib            # noqa: E402
from unitex_amd.flux import ops        # noqa: E402

BF, H = torch.bfloat16, int(os.environ.get("UTX_AB_HEADS", "24"))
SIZES = tuple(int(x) for x in os.environ.get("UTX_AB_SIZES", "13376,50240").split(","))      # executed tokens of the two operating points


def main():
    rounds = int(os.environ.get("UTX_AB_ROUNDS", "5"))
    repeats = int(os.environ.get("UTX_AB_REPEATS", "10"))
    as_json = "--json" in sys.argv      # bench.py: one JSON object on the last line instead of the table
    result = {}
    say = (lambda *a, **k: None) if as_json else print
    for S in SIZES:
        g = torch.Generator(device="cuda").manual_seed(S)
        S_pad = (S + 63) // 64 * 64
        Qh = (torch.randn(H, S_pad, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
        Kh = torch.randn(H, S_pad, 128, generator=g, device="cuda").to(BF)
        Vt = torch.randn(H, 128, S_pad, generator=g, device="cuda").to(BF)
        out = torch.empty(S, H * 128, dtype=BF, device="cuda")
        fl = 4.0 * S * S * 128 * H

        def run(peel):
            _lib.set_option("UTX_ATTN_PEEL", peel)
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=3.0, out=out)

        ref = None
        for peel in (0, 1, 2, 3, 4, 5, 6):
            run(peel)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
                # a first race screen (the variants that move a barrier): the same launch again and again, every result against the default kernel's bits
                miss = 0
                for _rep in range(repeats):
                    out.zero_()
                    run(peel)
                    torch.cuda.synchronize()
                    miss += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
                result.setdefault(str(S), {}).setdefault(str(peel), {}).update(bit_identical_to_default=same, repeats=repeats, mismatches_in_repeats=miss)
                say("S = %6d  UTX_ATTN_PEEL=%d  bit-identical to the default: %s   (%d of %d repeated launches differ)" % (S, peel, same, miss, repeats), flush=True)
        times = {0: [], 1: [], 2: [], 3: [], 4: [], 5: [], 6: []}
        for _ in range(rounds):
            for peel in (0, 1, 2, 3, 4, 5, 6):
                run(peel)      # warm
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _r in range(3):
                    run(peel)
                b.record()
                torch.cuda.synchronize()
                times[peel].append(a.elapsed_time(b) / 3.0)
        for peel in (0, 1, 2, 3, 4, 5, 6):
            t = sorted(times[peel])
            med = t[len(t) // 2]
            result.setdefault(str(S), {}).setdefault(str(peel), {}).update(med_ms=med, best_ms=t[0], tflops=fl / (med * 1e-3) / 1e12)
            say("S = %6d  UTX_ATTN_PEEL=%d  med %8.3f ms  best %8.3f ms  -> %7.1f TF/s" % (S, peel, med, t[0], fl / (med * 1e-3) / 1e12), flush=True)
    _lib.set_option("UTX_ATTN_PEEL", 0)
    if as_json:      # a first complete line: whatever the fp8 arm below does to this process, the bf16 result is out (the reader takes the LAST line that parses)
        import json
        print(json.dumps(result), flush=True)
    # the MX fp8 attention kernel (opt-in path) and its own peeled form, UTX_ATTN8_PEEL = 1 (attention_fp8.hip): same operands, bits against the default fp8 kernel
    try:      # its own try: a failure here must not take the bf16 results above with it
        for S in SIZES:
            g = torch.Generator(device="cuda").manual_seed(S + 1)
            S_pad = (S + 63) // 64 * 64
            Qh = (torch.randn(H, S_pad, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
            Kh = torch.randn(H, S_pad, 128, generator=g, device="cuda").to(BF)
            Vt = torch.randn(H, 128, S_pad, generator=g, device="cuda").to(BF)
            q8, qs = ops.quant_qk_mx8(Qh)
            k8, ks = ops.quant_qk_mx8(Kh)
            v8, vs = ops.quant_vt_mx8(Vt)
            del Qh, Kh, Vt
            out = torch.empty(S, H * 128, dtype=BF, device="cuda")
            fl = 4.0 * S * S * 128 * H

            def run8(peel):
                _lib.set_option("UTX_ATTN8_PEEL", peel)
                ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, out=out, key_bias_log2=3.0)

            run8(0)
            torch.cuda.synchronize()
            ref = out.clone()
            miss = 0
            for _rep in range(repeats + 1):
                out.zero_()
                run8(1)
                torch.cuda.synchronize()
                miss += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            key = "fp8_%d" % S
            result.setdefault(key, {}).setdefault("1", {}).update(bit_identical_to_default=(miss == 0), repeats=repeats + 1, mismatches_in_repeats=miss)
            say("fp8  S = %6d  UTX_ATTN8_PEEL=1  %d of %d launches differ from the default fp8 kernel" % (S, miss, repeats + 1), flush=True)
            times8 = {0: [], 1: []}
            for _ in range(rounds):
                for peel in (0, 1):
                    run8(peel)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _r in range(3):
                        run8(peel)
                    b.record()
                    torch.cuda.synchronize()
                    times8[peel].append(a.elapsed_time(b) / 3.0)
            for peel in (0, 1):
                tt = sorted(times8[peel])
                med = tt[len(tt) // 2]
                result.setdefault(key, {}).setdefault(str(peel), {}).update(med_ms=med, best_ms=tt[0], tflops=fl / (med * 1e-3) / 1e12)
                say("fp8  S = %6d  UTX_ATTN8_PEEL=%d  med %8.3f ms  best %8.3f ms  -> %7.1f TF/s" % (S, peel, med, tt[0], fl / (med * 1e-3) / 1e12), flush=True)
    except Exception as e:  # noqa: BLE001
        result["fp8_error"] = repr(e)[:300]
        say("fp8 arm failed: %r" % (e,), flush=True)
    _lib.set_option("UTX_ATTN8_PEEL", 0)
    if as_json:
        import json
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
