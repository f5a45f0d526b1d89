"""Determinism stress (-m gpu, ~1 minute): every "bit-identical" claim of DESIGN rests on the hand-scheduled kernels returning the same bits run after
run, warm or cold, alone or beside other work.  Round 3 ended with ONE unexplained one-ulp difference in ~15 runs of the full-width test; its cause (a
compiler copy of in-flight load registers in front of a wait, tests/test_asm_hazards_cpu.py) only showed when the loads were slow.  So every loop here
perturbs timing: a second stream keeps the chip busy on some repetitions, a 1 GB fill evicts the L2s / MALL on others, and the tables the fused epilogue
reads are re-allocated (cold TLB)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_ref  # noqa: E402

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class _Perturb:
    """timing noise between repetitions: i % 3 == 1 -> matrix products on a side stream, i % 4 == 2 -> evict the caches"""

    def __init__(self):
        self.side = torch.cuda.Stream()
        self.a = torch.randn(4096, 4096, device="cuda", dtype=BF)
        self.b = torch.randn(4096, 4096, device="cuda", dtype=BF)
        self.junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")

    def __call__(self, i):
        if i % 3 == 1:
            with torch.cuda.stream(self.side):
                for _ in range(1 + i % 4):
                    torch.mm(self.a, self.b)
        if i % 4 == 2:
            self.junk.fill_(i & 255)


def _full_width_model():
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.FluxConfig(num_double=1, num_single=1)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_double=1, num_single=1)
    S_txt = 512
    ids = [dit_ref.latent_image_ids(32, 128), dit_ref.latent_image_ids(32, 128, offset_y=32), dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=32)]
    img_ids = torch.cat(ids, 0)
    g = torch.Generator().manual_seed(63)
    lat = torch.randn(img_ids.shape[0], 64, generator=g).to(BF).cuda()
    enc = torch.zeros(S_txt, cfg.joint_dim).to(BF).cuda()
    pooled = torch.zeros(1, cfg.pooled_dim).to(BF).cuda()
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=2)
    m = FluxDiT(sd, shape, device="cuda:0")
    m.set_lora([(la, 1.0)])
    m.set_positions(torch.zeros(S_txt, 3), img_ids)
    return m, lat, enc, pooled


def test_full_width_plan_fused_and_unfused_100_forwards_bit_identical():
    """the configs[0]-shaped full-width 1 + 1-block plan (9280 executed tokens: the single block's projections have a ragged last row of tiles), default
    plan and fused q / k plan, 50 forwards each under perturbation + the split-tail default: one set of bits"""
    from unitex_amd import _lib
    m, lat, enc, pooled = _full_width_model()
    noise = _Perturb()
    try:
        _lib.set_option("UTX_GEMM_STREAMK", 0)      # the fused projection never splits its tail round: compare like with like
        ref = None
        for fused in (False, True):
            m.fuse_qk = fused
            m._drop_plans()
            m.set_conditioning(enc, pooled, 3.5)
            for i in range(50):
                noise(i)
                o = m.forward(lat, 0.4375).clone()
                if ref is None:
                    ref = o
                assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), "forward %d of the %s plan differs: %d elements, max |d| %g" % (
                    i, "fused" if fused else "default", int((o != ref).sum()), (o.float() - ref.float()).abs().max().item())
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
    m.fuse_qk = False
    m._drop_plans()
    m.set_conditioning(enc, pooled, 3.5)
    ref = m.forward(lat, 0.4375).clone()
    for i in range(30):
        noise(i)
        o = m.forward(lat, 0.4375)
        assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), "forward %d of the default plan (split tail rounds) differs" % i
    torch.cuda.synchronize()


def test_fused_qk_epilogue_on_ragged_tiles_with_cold_tables():
    """the provocation that reproduces round 3's one-off with the round-3 library (tools/qkf_ragged_repro.py): ragged row tiles + cos / sin evicted and
    re-allocated before every launch; Q / K must equal GEMM -> utx_qkv_post on every repetition"""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    H, K = 24, 3072
    D = H * 128
    M = 9280
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
    W = (torch.randn(3 * D, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(3 * D, device="cuda", generator=g).to(BF)
    wq = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
    wk = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
    S_pad = (M + 63) // 64 * 64
    ids = torch.stack([torch.zeros(M), torch.arange(M) // 97, torch.arange(M) % 97], 1).float()
    cos0, sin0 = [t.cuda().contiguous() for t in dit_ref.rope_tables(ids)]
    junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")

    def run(fused, cos, sin):
        qkv = torch.full((M, 3 * D), 3.0, dtype=BF, device="cuda")
        Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda")
        Kh = torch.zeros_like(Qh)
        Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
        qk = dict(cols=2 * D, tok_off=0, eps=1e-6, q_scale=0.1275, wq=wq, wk=wk, cos=cos, sin=sin, Qh=Qh, Kh=Kh) if fused else None
        junk.fill_(1)
        ops.gemm(x, W, bias=bias, out=qkv, qk_post=qk)
        ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, M, 0, H, q_scale=0.1275, skip_qk=fused)
        torch.cuda.synchronize()
        return Qh, Kh
    try:
        _lib.set_option("UTX_GEMM_STREAMK", 0)
        rq, rk = run(False, cos0, sin0)
        for r in range(12):
            q, k = run(True, cos0.clone(), sin0.clone())
            assert torch.equal(q.view(torch.int16), rq.view(torch.int16)) and torch.equal(k.view(torch.int16), rk.view(torch.int16)), \
                "cold fused launch %d differs from GEMM -> qkv_post" % r
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)


def test_gated_residual_gemm_100_launches_against_the_8_wave_kernel():
    """the default out-projection / MLP-down epilogue (gated residual, counted vmcnt waits behind residual loads and C stores) of the one-wave-per-SIMD
    kernel against the 8-wave persistent kernel: same bits on every launch, full and ragged M, in place, under perturbation"""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    noise = _Perturb()
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 0
    for (M, N, K) in [(9216, 3072, 3072), (9280, 3072, 12288), (13824, 3072, 15360)]:
        A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(BF)
        B = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / math.sqrt(K)).to(BF)
        bias = (torch.rand(N, device="cuda", generator=g) - 0.5).to(BF)
        gate = (torch.rand(N, device="cuda", generator=g) * 2 - 1).to(BF)
        res = torch.randn(M, N, device="cuda", generator=g).to(BF)
        try:
            _lib.set_option("UTX_GEMM_TILE", 2560)
            h8 = res.clone()
            ops.gemm(A, B, bias=bias, gate=gate, res=h8, out=h8)
        finally:
            _lib.set_option("UTX_GEMM_TILE", 0)
        assert ops.gemm_plan(M, N, K)["kernel"] == "w4"
        try:
            _lib.set_option("UTX_GEMM_STREAMK", 0)      # whole tiles only: the same K order as the 8-wave kernel
            for i in range(20):
                noise(i)
                h = res.clone()
                ops.gemm(A, B, bias=bias, gate=gate, res=h, out=h)      # in place, as the residual stream is updated
                assert torch.equal(h.view(torch.int16), h8.view(torch.int16)), "M=%d N=%d K=%d launch %d: %d elements differ" % (M, N, K, i, int((h != h8).sum()))
                n += 1
        finally:
            _lib.set_option("UTX_GEMM_STREAMK", 1)
        ref = None                                      # the default (tail round cut along K, fix-up kernel): reproduces itself
        for i in range(14):
            noise(i + 1)
            h = res.clone()
            ops.gemm(A, B, bias=bias, gate=gate, res=h, out=h)
            if ref is None:
                ref = h
                assert (ref.float() - h8.float()).abs().max().item() <= 0.0625 * max(1.0, h8.float().abs().max().item())
            assert torch.equal(h.view(torch.int16), ref.view(torch.int16)), "M=%d N=%d K=%d split-tail launch %d differs" % (M, N, K, i)
            n += 1
    assert n >= 100
    torch.cuda.synchronize()


def test_two_stream_fp8_pruned_plan_600_back_to_back_forwards():
    """THE configuration in which round 4 found the two-stream form of the double blocks not reproducible (1-1.5 % of the forwards: one q / k row of one head wrong
    out of the image half's utx_qkv_post while the text half's MFMA GEMMs ran beside it on the second stream): full width, 1 double + 2 single blocks, MX fp8
    linears, last block pruned, adapters on, forwards back to back with the comparison as the only synchronisation.  The cause was the packed fp32 instructions
    in the elementwise kernels (csrc/build.py builds dit_elementwise.hip without them since); at the old rate 600 forwards would show 6-9 differences."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.FluxConfig(num_double=1, num_single=2)
    shape = FluxShape(num_double=1, num_single=2)
    S_txt = 512
    img_ids = torch.cat([dit_ref.latent_image_ids(32, 128), dit_ref.latent_image_ids(32, 128, offset_y=32), dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=32)], 0)
    enc = torch.zeros(S_txt, cfg.joint_dim).to(BF).cuda(); pooled = torch.zeros(1, cfg.pooled_dim).to(BF).cuda()
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(img_ids.shape[0], 64, generator=g).to(BF).cuda()
    junk = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
    m.set_text_stream(True)
    m.set_lora([(dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=2), 1.0)])
    m.set_positions(torch.zeros(S_txt, 3), img_ids)
    m.set_output_rows(4096)
    m.set_conditioning(enc, pooled, 3.5)
    assert m.overlap_text and any(fn == "par" for fn, _ in next(iter(m._plans.values()))["plan"])
    ref = m.forward(lat, 0.5)[:4096].clone()
    bad = []
    for i in range(600):
        if i % 3 == 1:
            junk.fill_(i & 255)
        o = m.forward(lat, 0.5)[:4096]
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            bad.append(i)
    assert not bad, "%d of 600 two-stream forwards differ from the first (first at %s)" % (len(bad), bad[:5])


def test_fast_attention_loop_320_launches_against_the_general_loops_bits():
    """Round 5: the attention kernel's default loop moved the tile's ONE barrier (between S2 and S3), requests tile t + 2 behind it and reads the next tile's first K
    fragments under S3 -- a change of the class that produced round 3's one-off.  A race there would show as a RARE mismatch, so: 320 launches (two shapes: ragged
    last tile + key-split tail round + key multiplicity; whole tiles, pruned queries) under the perturbing side stream / cache evictions, every one against the
    bits of the general loop (UTX_ATTN_PEEL=0: one barrier at the end of each tile, the default until round 4)."""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    noise = _Perturb()
    q64_prev = _lib.get_options()["UTX_ATTN_Q64"]
    _lib.set_option("UTX_ATTN_Q64", 0)      # this test is about the 8 x 32 kernel's two loops (round 6: the 4 x 64 kernel would take the second shape in both arms)
    for H, S, S_q, kb, reps in ((24, 3000, None, 3.0, 200), (24, 4096, 2816, 0.0, 120)):
        g = torch.Generator(device="cuda").manual_seed(S)
        S_pad = (S + 63) // 64 * 64
        Qh = (torch.randn(H, S_pad, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
        Kh = torch.randn(H, S_pad, 128, generator=g, device="cuda").to(BF)
        Kh[:, S // 2] *= 3.0           # a late outlier key: some waves take the exact re-centring path inside the loop
        Vt = torch.randn(H, 128, S_pad, generator=g, device="cuda").to(BF)
        try:
            _lib.set_option("UTX_ATTN_PEEL", 0)
            ref = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=S_q).clone()
            torch.cuda.synchronize()
        finally:
            _lib.set_option("UTX_ATTN_PEEL", 1)
        out = torch.empty_like(ref)
        bad = []
        for i in range(reps):
            noise(i)
            out.zero_()
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=S_q, out=out)
            if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                bad.append((i, int((out.view(torch.int16) != ref.view(torch.int16)).sum())))
        assert not bad, "S = %d: %d of %d launches of the fast loop differ from the general loop: %s" % (S, len(bad), reps, bad[:5])
    torch.cuda.synchronize()
    _lib.set_option("UTX_ATTN_Q64", q64_prev)


def test_q64_attention_stream_300_launches_against_the_8x32_kernels_bits():
    """Round 6: the default attention kernel is a hand-placed instruction stream whose every wait is counted by its generator (one barrier per tile, K(t + 3) / V(t + 2)
    requested behind it, fragments read three groups ahead) -- a missed wait or a slot reused too early would show as a RARE mismatch.  300 launches on two shapes (key
    multiplicity + key-split tail round; pruned queries, nt mod 3 = 1) under the perturbing side stream / cache evictions, every one against the bits of the 8 x 32 kernel
    (no outlier keys: the shapes on which the two kernels are bit-identical by construction)."""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    assert _lib.get_options()["UTX_ATTN_Q64"] == 1
    noise = _Perturb()
    for H, S, S_q, kb, reps in ((24, 3328, None, 3.0, 180), (24, 4160, 2816, 0.0, 120)):
        g = torch.Generator(device="cuda").manual_seed(S)
        Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
        Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
        Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
        try:
            _lib.set_option("UTX_ATTN_Q64", 0)
            ref = ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=S_q).clone()
            torch.cuda.synchronize()
        finally:
            _lib.set_option("UTX_ATTN_Q64", 1)
        out = torch.empty_like(ref)
        bad = []
        for i in range(reps):
            noise(i)
            out.zero_()
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, key_bias_log2=kb, S_q=S_q, out=out)
            if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                bad.append((i, int((out.view(torch.int16) != ref.view(torch.int16)).sum())))
        assert not bad, "S = %d: %d of %d launches of the 4 x 64 stream differ from the 8 x 32 kernel: %s" % (S, len(bad), reps, bad[:5])
    torch.cuda.synchronize()
