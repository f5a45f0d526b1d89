"""The 4 x 64 attention kernel (attention_q64.hip: one wave per SIMD, the generated stream of tools/gen_attn_q64.py; the default of the launches it takes since round 6)
through the C ABI.

Contract under test (the kernel's header):
  * per element the arithmetic and its order are the 8 x 32 kernel's, with m = the exact maximum of the first 32 keys and NO re-centring: wherever the 8 x 32 kernel does not
    re-centre behind block 0 the two are BIT-IDENTICAL -- tile counts 1 .. 5 and beyond (prologue-only launches, the nt mod 3 tails of the three-tiles-per-trip loop), key
    multiplicity on tile 0, S_q < S_kv, the key-split tail round at 24 heads, BASELINE's full size;
  * where the 8 x 32 kernel re-centres (a block's row sum beyond 8192) the results are two valid roundings of the same softmax: both within the same tolerance of the oracle;
  * beyond the 2^96 headroom the repair pass (8 x 32 kernel on the flagged query blocks) rewrites the rows: tests/test_dit_ops_gpu.py::test_attention_q64_kernel_and_repair_pass;
  * launches it does not take (ragged S, natural-exp scale, periodic key multiplicity, block-strided operands, no scratch) run the 8 x 32 kernel -- same entry point."""
import math

import pytest
import torch

from oracle import dit_ref

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
QS = 1.4426950408889634 / math.sqrt(128.0)


def _mk(H, S, seed, gain=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * QS * gain).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    return Qh, Kh, Vt


def _run(q64, Qh, Kh, Vt, S, kb=0.0, Sq=None):
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    prev = _lib.get_options()["UTX_ATTN_Q64"]
    _lib.set_option("UTX_ATTN_Q64", 1 if q64 else 0)
    try:
        out = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=Sq)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN_Q64", prev)
    return out


def test_q64_is_the_default_of_the_launches_it_takes():
    from unitex_amd import _lib
    assert _lib.get_options()["UTX_ATTN_Q64"] == 1


@pytest.mark.parametrize("H,S,kb,Sq", [(1, 64, 0.0, None), (3, 128, 0.0, None), (1, 192, 3.0, None), (3, 256, 0.0, None), (2, 320, 0.0, None), (1, 384, 0.0, 100), (3, 448, 0.0, None),
                                       (2, 512, 3.0, None), (2, 1024, 3.0, 300), (3, 2048, 0.0, 1000), (3, 4096, 0.0, None),
                                       (24, 3328, 3.0, None), (24, 6272, 0.0, None), (24, 13376, 3.0, None), (24, 13376, 0.0, 2048)])
def test_q64_bit_identical_to_the_8x32_kernel(H, S, kb, Sq):
    """tile counts 1, 2, 3, 4, 5, 6, 7 (prologue only; every nt mod 3 tail), then long loops; 24 heads: launches whose last round is cut along the keys (both kernels use the
    same plan, the same partial-row format and the same merge)"""
    Qh, Kh, Vt = _mk(H, S, 1000 * H + S)
    ref = _run(False, Qh, Kh, Vt, S, kb, Sq)
    got = _run(True, Qh, Kh, Vt, S, kb, Sq)
    again = _run(True, Qh, Kh, Vt, S, kb, Sq)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), "%d of %d elements differ, max |d| %g" % (
        int((got.view(torch.int16) != ref.view(torch.int16)).sum()), got.numel(), (got.float() - ref.float()).abs().max().item())
    assert torch.equal(got.view(torch.int16), again.view(torch.int16)), "not reproducible run to run"


def test_q64_bit_identical_at_baseline_full_size():
    """BASELINE configs[1] as executed: 24 heads, 64 de-duplicated text rows with key weight 2^3 + 50 176 image tokens"""
    H, S = 24, 50240
    Qh, Kh, Vt = _mk(H, S, 7)
    ref = _run(False, Qh, Kh, Vt, S, 3.0)
    got = _run(True, Qh, Kh, Vt, S, 3.0)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_q64_where_the_8x32_kernel_recentres_both_hold_the_oracle_tolerance():
    """peaked rows (Q three times the usual scale): later keys exceed the first block's maximum by enough for the 8 x 32 kernel to re-centre; the 4 x 64 kernel keeps m -- the
    two outputs differ in the last bf16 bits of some rows and both sit within the attention tolerance of the fp32 oracle (tests/test_dit_ops_gpu.py: 4e-2 on O(1) outputs)"""
    H, S = 4, 8192
    Qh, Kh, Vt = _mk(H, S, 8192 + H, gain=3.0)
    a = _run(False, Qh, Kh, Vt, S)
    b = _run(True, Qh, Kh, Vt, S)
    ndiff = int((a.view(torch.int16) != b.view(torch.int16)).sum())
    assert ndiff > 0, "the case no longer makes the 8 x 32 kernel re-centre: pick a steeper one"
    rows = torch.arange(0, S, 97)
    q = (Qh[:, rows].float().cpu() / QS)                      # the oracle applies 1 / sqrt(d) itself
    # the oracle's attention (oracle/dit_ref.py::sdpa, fp32: softmax(q k^T / sqrt d) v) on every 97th query row
    s = torch.einsum("hqd,hkd->hqk", q, Kh.float().cpu()) / math.sqrt(128.0)
    ref = torch.einsum("hqk,hkd->hqd", torch.softmax(s, -1), Vt.float().cpu().transpose(1, 2))
    full = dit_ref.sdpa(Qh[:1, :512].float().cpu() / QS, Kh[:1, :512].float().cpu(), Vt[:1, :, :512].float().cpu().transpose(1, 2), False)      # same expression as the oracle's, on a corner
    s0 = torch.einsum("hqd,hkd->hqk", Qh[:1, :512].float().cpu() / QS, Kh[:1, :512].float().cpu()) / math.sqrt(128.0)
    assert (full - torch.einsum("hqk,hkd->hqd", torch.softmax(s0, -1), Vt[:1, :, :512].float().cpu().transpose(1, 2))).abs().max().item() < 1e-5
    for name, out in (("8x32", a), ("4x64", b)):
        o = out.float().cpu().view(S, H, 128)[rows].permute(1, 0, 2)
        err = (o - ref).abs().max().item()
        assert err < 4e-2, "%s kernel vs oracle on peaked rows: %g" % (name, err)
    assert (a.float() - b.float()).abs().max().item() < 4e-2


def test_launches_the_q64_kernel_does_not_take_run_the_8x32_kernel():
    """ragged S and the natural-exp scale go through the same entry point to the 8 x 32 kernel: the option changes nothing for them"""
    from unitex_amd.flux import ops
    H, S = 2, 1000
    g = torch.Generator(device="cuda").manual_seed(3)
    S_pad = 1024
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda"); Kh = torch.zeros_like(Qh); Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
    Qh[:, :S] = (torch.randn(H, S, 128, generator=g, device="cuda") * QS).to(BF); Kh[:, :S] = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt[:, :, :S] = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    assert torch.equal(_run(True, Qh, Kh, Vt, S), _run(False, Qh, Kh, Vt, S))
    from unitex_amd import _lib
    outs = []
    for arm in (0, 1):
        _lib.set_option("UTX_ATTN_Q64", arm)
        outs.append(ops.attention(Qh, Kh, Vt, S=S_pad, scale=None))
        torch.cuda.synchronize()
    _lib.set_option("UTX_ATTN_Q64", 1)
    assert torch.equal(outs[0], outs[1])


def test_both_kernels_at_the_longest_joint_sequence_of_the_bench():
    """maximum size: the joint strip of BASELINE configs[4]'s resolution in the reference's own semantics (bench.py workload strip2048x8: 8 views of 2048^2 = 131 072 noise +
    131 072 control + 1024 dual tokens + 64 de-duplicated text rows = 263 232 executed tokens, 4113 key tiles, 1029 query blocks per head) on two heads: every address product
    beyond the sizes the other tests reach (row offsets past 2^25 elements, V^T rows half a megabyte long), the key-split tail round (2058 work items on 256 CUs), key
    multiplicity on tile 0.  The 4 x 64 stream and the 8 x 32 loop agree bit for bit, and sampled rows -- first, last, around block and tile edges -- match an fp64 softmax
    over ALL keys of the bf16 operands."""
    H, S, kb = 2, 64 + 131072 + 131072 + 1024, 3.0
    Qh, Kh, Vt = _mk(H, S, 77)
    ref = _run(False, Qh, Kh, Vt, S, kb)
    got = _run(True, Qh, Kh, Vt, S, kb)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), "%d of %d elements differ" % (int((got.view(torch.int16) != ref.view(torch.int16)).sum()), got.numel())
    rows = torch.tensor([0, 63, 64, 255, 256, 65535, 65536, 131071, 131136, 200000, S - 257, S - 256, S - 65, S - 1], device="cuda")
    for h in range(H):
        s2 = Qh[h, rows].double() @ Kh[h].double().t()                    # base-2 exponents: Q carries scale * log2(e)
        s2[:, :64] += kb
        p = torch.exp2(s2 - s2.max(dim=-1, keepdim=True).values)
        want = (p @ Vt[h].double().t()) / p.sum(dim=-1, keepdim=True)
        err = (got[rows][:, h * 128:(h + 1) * 128].double() - want).abs().max().item()
        # outputs are means over ~2.6e5 keys of unit-variance values: |o| ~ 1e-2; the bound is the small-S tests' bound scaled by that (P is bf16 in the PV product: 2^-9 relative per term)
        assert err < 4e-4, "head %d: sampled rows differ from the fp64 softmax by %g (max |want| %g)" % (h, err, want.abs().max().item())
