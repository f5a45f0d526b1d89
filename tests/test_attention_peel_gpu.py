"""The attention kernels' fast loops against their general loops, bit for bit.

bf16 (attention_glds.hip, FAST; UTX_ATTN_PEEL = 1, the default since round 5; 0 = the general loop, the default until round 4): the first tile and a ragged last tile run in
front of / behind the loop, the loop body carries none of their branches, the tile's barrier sits between S2 and S3 and the next tile's first K fragments are read under
S3's MFMAs.  MX fp8 (attention_fp8.hip VAR 1; UTX_ATTN8_PEEL = 1, default; 0 = general loop): tile 0 / ragged tile outside the loop, exponentials in quarters under the PV MFMAs.
Same arithmetic in the same order per element: every output must equal the general loop's BIT FOR BIT, on every feature of the launch (ragged S, pruned queries, key
multiplicity with and without a period, the key-split tail round, a spike that forces the exact re-centring path late in the sequence).

Round 5, first run on hardware (profiles/r05_attn_peel_tests.log): every case equal except (24, 3000, spike) -- 36 of 9.2 M outputs one bf16 ulp apart, in rows that had taken
the re-centring path: hipcc had contracted `l_run *= alpha; ...; l_run += ps` into v_fmac_f32 in the tail-duplicated copies of that path only (profiles/r05_peel_diff_probe.log);
the multiply is now fenced against contraction in every copy (ag_mul_nofuse) and the case is equal."""
import math

import pytest
import torch

pytestmark = [pytest.mark.gpu]
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
    (4, 1024, None, 2.0, 8, False),          # periodic key multiplicity (sequence parallelism): the KBP instance -- key-multiplicity tiles through their own copy of the fast tile
    (24, 3000, None, 3.0, 5, True),          # the same on the full machine: ragged last tile, re-centring, the key-split tail round (a split's first tile index is not 0)
    (4, 1500, 700, 3.0, 3, True),            # periodic + pruned queries + ragged
    (2, 640, None, 1.0, 1, False),           # EVERY tile a key-multiplicity tile (the inner loop of ordinary tiles stays empty)
    (4, 960, None, 3.0, 100, False),         # a period longer than the sequence: tile 0 only
    (24, 6336 * 2, None, 3.0, 99, False),    # two ranks' blocks of 64 text + 6272 image rows (the 8-rank bench shape's per-rank block), period 99
    (4, 1500, 700, 3.0, 0, True),            # pruned queries (S_q < S), ragged, key multiplicity
    (24, 3000, None, 0.0, 0, True),          # 288 workgroups > 256 CUs: full round + key-split tail round (both through the same instance)
    (24, 2900, 2816, 3.0, 0, False),         # the same with pruned queries and key multiplicity
]


@pytest.fixture(autouse=True)
def _the_8x32_kernel():
    """since round 6 the launches the 4 x 64 kernel takes would bypass BOTH loops compared here: this module is about the 8 x 32 kernel (the path of ragged / periodic-key-multiplicity /
    block-strided launches and of the repair pass), so it runs with UTX_ATTN_Q64=0; tests/test_attention_q64_gpu.py holds the 4 x 64 kernel against this one"""
    from unitex_amd import _lib
    prev = _lib.get_options()["UTX_ATTN_Q64"]
    _lib.set_option("UTX_ATTN_Q64", 0)
    yield
    _lib.set_option("UTX_ATTN_Q64", prev)


@pytest.mark.parametrize("H,S,S_q,kb,period,spike", CASES)
def test_fast_attention_loop_equals_the_general_loop_bit_for_bit(H, S, S_q, kb, period, spike):
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    Qh, Kh, Vt = _inputs(H, S, seed=S + 7 * H, spike=spike)
    assert _lib.get_options()["UTX_ATTN_PEEL"] == 1, "the fast loop is the default"
    out = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, key_bias_period=period, S_q=S_q)
    torch.cuda.synchronize()
    try:
        _lib.set_option("UTX_ATTN_PEEL", 0)
        ref = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, key_bias_period=period, S_q=S_q)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN_PEEL", 1)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        "the fast loop differs from the general loop: max |d| = %g" % (out.float() - ref.float()).abs().max().item()


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
def test_fast_fp8_attention_loop_equals_the_general_fp8_loop_bit_for_bit(H, S, S_q, kb, period, spike):
    """UTX_ATTN8_PEEL=1 (default; attention_fp8.hip, attn_fwd_fp8_kernel<1>): tile 0 / a ragged last tile outside the loop, the loop's exponentials in quarters under the PV MFMAs
    (running sums carried across the quarters: the same summation order).  Same MX operands in, the general loop's bits out."""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    Qh, Kh, Vt = _inputs(H, S, seed=S + 11 * H, spike=spike)
    q8, qs = ops.quant_qk_mx8(Qh)
    k8, ks = ops.quant_qk_mx8(Kh)
    v8, vs = ops.quant_vt_mx8(Vt)
    assert _lib.get_options()["UTX_ATTN8_PEEL"] == 1
    out = ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, S_q=S_q, key_bias_log2=kb, key_bias_period=period)
    torch.cuda.synchronize()
    try:
        _lib.set_option("UTX_ATTN8_PEEL", 0)
        ref = ops.attention_fp8(q8, qs, k8, ks, v8, vs, S=S, S_q=S_q, key_bias_log2=kb, key_bias_period=period)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN8_PEEL", 1)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), \
        "the fast fp8 loop differs from the general fp8 loop: max |d| = %g" % (out.float() - ref.float()).abs().max().item()
