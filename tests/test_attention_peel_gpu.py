"""UTX_ATTN_PEEL = 1 ... 6 (attention_glds.hip, VAR 12 ... 17; opt-in) and UTX_ATTN8_PEEL = 1 (attention_fp8.hip, its last test below): the LDS-DMA attention kernel with the first tile and a ragged last tile run in front of / behind
the loop, so that the loop body carries none of their branches and its QK^T || exp and PV || exp stages are single basic blocks; 5 (VAR 16) also moves the tile's barrier between S2
and S3 and reads the next tile's first K fragments under S3's MFMAs.  Same arithmetic in the same order
per element: every output must equal the default kernel's BIT FOR BIT, on every feature of the launch (ragged S, pruned queries, key multiplicity with and without a
period, the key-split tail round, a spike that forces the exact re-centring path late in the sequence).

These variants were written in a session that had no GPU minutes left: they have been compiled for gfx950 and their listings read (230 / 248 / 234 / 234 / 230 VGPRs, no scratch; the
default instances' listings are byte-identical to what they were before the tile body became a macro), but they have NOT run on hardware yet.  Until they have, this
file only runs on request -- UTX_RUN_UNVALIDATED=1 python -m pytest tests/test_attention_peel_gpu.py -m gpu -- so that code nobody has executed cannot turn the suite
red, and it is the first thing tools/attn_peel_ab.py's user should run.  The default kernel is what every other test and bench.py exercise."""
import math
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("UTX_RUN_UNVALIDATED", "0") != "1",
                                 reason="opt-in attention variants that have not run on hardware yet: set UTX_RUN_UNVALIDATED=1 (see the module docstring)")]
BF = torch.bfloat16


def _inputs(H, S, seed, spike):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(H, S, 128, generator=g).to(BF) for _ in range(3))
    if spike:      # a late key far above the running maximum: the sum check fails and the exact path re-centres inside the loop
        k[:, S - 3] = (q[:, 5].float() * 3.0).to(BF)
        k[:, S // 2] = (q[:, 7].float() * 2.0).to(BF)
    S_pad = (S + 63) // 64 * 64
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda")
    Qh[:, :S] = (q.float() * (1.4426950408889634 / math.sqrt(128.0))).to(BF).cuda()       # pre-scaled, as utx_qkv_post hands it over (scale = 0 selects that form)
    Kh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda")
    Kh[:, :S] = k.cuda()
    Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
    Vt[:, :, :S] = v.cuda().transpose(1, 2)
    return Qh, Kh, Vt


CASES = [
    # H, S, S_q, key_bias_log2, key_bias_period, spike
    (2, 64, None, 0.0, 0, False),            # one tile: first and last at once
    (2, 100, None, 0.0, 0, False),           # two tiles, the second ragged
    (3, 1000, None, 0.0, 0, True),           # ragged last tile + re-centring in the loop
    (2, 2304, None, 0.0, 0, True),           # whole tiles only
    (4, 960, None, 3.0, 0, False),           # key multiplicity on tile 0
    (4, 1024, None, 2.0, 8, False),          # periodic key multiplicity: the variant must fall back to the general loop
    (4, 1500, 700, 3.0, 0, True),            # pruned queries (S_q < S), ragged, key multiplicity
    (24, 3000, None, 0.0, 0, True),          # 288 workgroups > 256 CUs: full round + key-split tail round (both through the same instance)
    (24, 2900, 2816, 3.0, 0, False),         # the same with pruned queries and key multiplicity
]


@pytest.mark.parametrize("peel", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("H,S,S_q,kb,period,spike", CASES)
def test_peeled_attention_loop_equals_the_default_kernel_bit_for_bit(peel, H, S, S_q, kb, period, spike):
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    Qh, Kh, Vt = _inputs(H, S, seed=S + 7 * H, spike=spike)
    assert _lib.get_options()["UTX_ATTN_PEEL"] == 0
    ref = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, key_bias_period=period, S_q=S_q)
    torch.cuda.synchronize()
    try:
        _lib.set_option("UTX_ATTN_PEEL", peel)
        out = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, key_bias_period=period, S_q=S_q)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN_PEEL", 0)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        "UTX_ATTN_PEEL=%d differs from the default kernel: max |d| = %g" % (peel, (out.float() - ref.float()).abs().max().item())


FP8_CASES = [
    # H, S, S_q, key_bias_log2, key_bias_period, spike
    (2, 64, None, 0.0, 0, False),
    (2, 100, None, 0.0, 0, False),
    (3, 1000, None, 0.0, 0, True),
    (2, 2304, None, 0.0, 0, True),
    (4, 960, None, 3.0, 0, False),
    (4, 1024, None, 2.0, 8, False),          # periodic key multiplicity: the variant must fall back to the general loop
    (4, 1500, 700, 3.0, 0, True),
    (24, 3000, None, 0.0, 0, True),
]


@pytest.mark.parametrize("H,S,S_q,kb,period,spike", FP8_CASES)
def test_peeled_fp8_attention_loop_equals_the_default_fp8_kernel_bit_for_bit(H, S, S_q, kb, period, spike):
    """UTX_ATTN8_PEEL=1 (attention_fp8.hip, attn_fwd_fp8_kernel<1>): tile 0 / a ragged last tile outside the loop, the loop's exponentials in quarters under the PV MFMAs
    (running sums carried across the quarters: the same summation order).  Same MX operands in, the default fp8 kernel's bits out."""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    Qh, Kh, Vt = _inputs(H, S, seed=S + 11 * H, spike=spike)
    q8, qs = ops.quant_qk_mx8(Qh)
    k8, ks = ops.quant_qk_mx8(Kh)
    v8, vs = ops.quant_vt_mx8(Vt)
    assert _lib.get_options()["UTX_ATTN8_PEEL"] == 0
    ref = ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, S_q=S_q, key_bias_log2=kb, key_bias_period=period)
    torch.cuda.synchronize()
    try:
        _lib.set_option("UTX_ATTN8_PEEL", 1)
        out = ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, S_q=S_q, key_bias_log2=kb, key_bias_period=period)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN8_PEEL", 0)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        "UTX_ATTN8_PEEL=1 differs from the default fp8 kernel: max |d| = %g" % (out.float() - ref.float()).abs().max().item()
