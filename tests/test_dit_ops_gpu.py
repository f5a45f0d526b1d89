"""GPU parity tests (through the C ABI) of the DiT kernels against the CPU oracle (oracle/dit_ref.py).

Tolerances (stated per test): inputs are bf16 on both sides; the oracle accumulates in fp32 and rounds
at the same tensor boundaries as the kernels, so differences come from accumulation order and from
the P-matrix bf16 quantisation inside flash attention.
"""
import math
import os

import pytest
import torch

from oracle import dit_ref

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from unitex_amd.flux import ops
    return ops


def _mk_attn_inputs(H, S, seed, spike=False):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(H, S, 128, generator=g).to(BF)
    k = torch.randn(H, S, 128, generator=g).to(BF)
    v = torch.randn(H, S, 128, generator=g).to(BF)
    if spike:  # force the online-softmax rescale branch late in the sequence (guide rule 26)
        k[:, S - 3] = (q[:, 5].float() * 3.0).to(BF)
        k[:, S // 2] = (q[:, 7].float() * 2.0).to(BF)
    return q, k, v


def _run_attn(q, k, v, prescaled=False):
    ops = _ops()
    H, S, _ = q.shape
    S_pad = (S + 63) // 64 * 64
    qd = q.cuda()
    if prescaled:   # what qkv_post(q_scale) hands to the kernel: Q * scale * log2(e), rounded once to bf16
        qd = (qd.float() * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda"); Qh[:, :S] = qd
    Kh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda"); Kh[:, :S] = k.cuda()
    Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda"); Vt[:, :, :S] = v.cuda().transpose(1, 2)
    out = ops.attention(Qh, Kh, Vt, S=S, scale=0.0 if prescaled else None)
    torch.cuda.synchronize()
    return out.float().cpu().view(S, H, 128).permute(1, 0, 2)


@pytest.mark.parametrize("H,S,spike", [(1, 64, False), (2, 256, False), (3, 200, False), (2, 1000, True),
                                       (2, 2304, True), (24, 512, False)])
def test_attention_matches_oracle(H, S, spike):
    q, k, v = _mk_attn_inputs(H, S, seed=S + H, spike=spike)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    out = _run_attn(q, k, v)
    err = (out - ref).abs().max().item()
    # |v| <= ~4.5; P rounded to bf16 (2^-9 rel) + bf16 output rounding
    assert err < 3e-2, "attention max-abs err %g (H=%d S=%d)" % (err, H, S)
    assert torch.isfinite(out).all()
    # pre-scaled-Q mode (the one FluxDiT uses): one extra bf16 rounding of Q at a different scale
    out2 = _run_attn(q, k, v, prescaled=True)
    err2 = (out2 - ref).abs().max().item()
    assert err2 < 4e-2, "prescaled attention max-abs err %g (H=%d S=%d)" % (err2, H, S)


def test_attention_large_negative_and_positive_scores():
    """first-tile maxima far from 0 (both signs) and a late spike: exercises the re-centring path."""
    H, S = 1, 320
    g = torch.Generator().manual_seed(2)
    q = torch.randn(H, S, 128, generator=g)
    k = torch.randn(H, S, 128, generator=g)
    v = torch.randn(H, S, 128, generator=g)
    q[:, :100] *= 12.0           # |scores| up to ~ +-150 / sqrt(128) * ...
    k[:, 300] = q[:, 3] * 0.5     # late spike for row 3
    k[:, :64] = -q[:, 7:8] * 0.3  # strongly negative first tile for row 7
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    for pre in (False, True):
        out = _run_attn(q, k, v, prescaled=pre)
        assert torch.isfinite(out).all()
        assert (out - ref).abs().max().item() < 6e-2


def test_attention_asymmetric_layout():
    """V = one-hot along d with distinct keys: catches transposed / permuted fragments (guide G9)."""
    H, S = 1, 128
    q = torch.zeros(H, S, 128); k = torch.zeros(H, S, 128); v = torch.zeros(H, S, 128)
    for s in range(S):
        q[0, s, s % 128] = 6.0
        k[0, s, s % 128] = 6.0
        v[0, s, (3 * s + 1) % 128] = float(1 + (s % 7))
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    out = _run_attn(q, k, v)
    assert (out - ref).abs().max().item() < 3e-2


def _gemm_ref(A, B, bias=None, A2=None, B2=None, alpha=1.0, gelu_from=None, gate=None, res=None, lora_seg=None,
              lora_limit=None):
    y = A.float() @ B.float().t()
    N = B.shape[0]
    if A2 is not None:
        K2 = B2.shape[1]
        seg = N if lora_seg is None else lora_seg
        lim = N if lora_limit is None else lora_limit
        for n0 in range(0, lim, seg):
            si = n0 // seg
            y[:, n0:n0 + seg] += A2.float()[:, si * K2:(si + 1) * K2] @ B2.float()[n0:n0 + seg].t()
    y = y * alpha
    if bias is not None:
        y = y + bias.float()
    y = y.to(BF).float()
    if gelu_from is not None:
        y[:, gelu_from:] = dit_ref.gelu_tanh(y[:, gelu_from:])
        y = y.to(BF).float()
    if gate is not None:
        y = (res.float() + (gate.float() * y).to(BF).float()).to(BF).float()
    return y


def _close(out, ref, what):
    out = out.float().cpu()
    denom = ref.abs().clamp_min(1.0)
    rel = ((out - ref).abs() / denom).max().item()
    assert rel < 1.6e-2, "%s: max rel err %g" % (what, rel)  # <= 2 bf16 ulp (2^-7) on O(1) values


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (200, 136, 128), (512, 64, 256), (1000, 3072, 3072)])
def test_gemm_plain_and_bias(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) / math.sqrt(K) * 4).to(BF)
    B = torch.randn(N, K, generator=g).to(BF)
    bias = torch.randn(N, generator=g).to(BF)
    out = ops.gemm(A.cuda(), B.cuda(), bias=bias.cuda())
    torch.cuda.synchronize()
    _close(out, _gemm_ref(A, B, bias), "gemm+bias")


def test_gemm_epilogues_and_lora():
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    M, K, D = 320, 256, 256
    N = 3 * D + 4 * D  # single-block fused [q|k|v|mlp]
    A = (torch.randn(M, K, generator=g) / 4).to(BF)
    B = (torch.randn(N, K, generator=g) / 4).to(BF)
    bias = torch.randn(N, generator=g).to(BF)
    R = 64
    T = (torch.randn(M, 3 * R, generator=g) / 8).to(BF)
    Bl = torch.zeros(N, R).to(BF)
    Bl[:3 * D] = (torch.randn(3 * D, R, generator=g) / 4).to(BF)
    c0 = torch.empty(M, 3 * D, dtype=BF, device="cuda")
    c1 = torch.empty(M, 4 * D, dtype=BF, device="cuda")
    ops.gemm(A.cuda(), B.cuda(), bias=bias.cuda(), out=c0, A2=T.cuda(), B2=Bl.cuda(), lora_n_limit=3 * D,
             lora_seg_n=D, gelu_from=3 * D, n_split=3 * D, C1=c1)
    torch.cuda.synchronize()
    ref = _gemm_ref(A, B, bias, A2=T, B2=Bl, gelu_from=3 * D, lora_seg=D, lora_limit=3 * D)
    _close(c0, ref[:, :3 * D], "fused qkv (+lora)")
    _close(c1, ref[:, 3 * D:], "fused mlp (gelu)")
    # gated residual, in place
    N2 = 256
    B2w = (torch.randn(N2, K, generator=g) / 4).to(BF)
    b2 = torch.randn(N2, generator=g).to(BF)
    gate = torch.randn(N2, generator=g).to(BF)
    res = torch.randn(M, N2, generator=g).to(BF)
    resd = res.cuda().clone()
    ops.gemm(A.cuda(), B2w.cuda(), bias=b2.cuda(), out=resd, gate=gate.cuda(), res=resd)
    torch.cuda.synchronize()
    _close(resd, _gemm_ref(A, B2w, b2, gate=gate, res=res), "gated residual")
    # alpha (LoRA-down scale)
    out = ops.gemm(A.cuda(), B2w.cuda(), alpha=0.37)
    torch.cuda.synchronize()
    _close(out, _gemm_ref(A, B2w, alpha=0.37), "alpha")


def test_gemv():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 256, generator=g).to(BF)
    W = (torch.randn(1000, 256, generator=g) / 16).to(BF)
    b = torch.randn(1000, generator=g).to(BF)
    out = ops.gemv(x.cuda(), W.cuda(), b.cuda(), silu_in=True)
    xin = dit_ref.silu(x.float()).to(BF).float()
    ref = (xin @ W.float().t() + b.float()).to(BF).float()
    _close(out, ref, "gemv silu_in")
    out = ops.gemv(x.cuda(), W.cuda(), b.cuda(), silu_out=True)
    ref = dit_ref.silu((x.float() @ W.float().t() + b.float()).to(BF).float()).to(BF).float()
    _close(out, ref, "gemv silu_out")


def test_ln_mod_and_sched():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(300, 3072, generator=g) * 2 + 0.3).to(BF)
    shift = torch.randn(3072, generator=g).to(BF)
    scale = (torch.randn(3072, generator=g) * 0.5).to(BF)
    out = ops.ln_mod(x.cuda(), shift.cuda(), scale.cuda()).float().cpu()
    ref = dit_ref.layer_norm_mod(x.float(), shift.float(), scale.float(), 1e-6, True)
    bad = ((out - ref).abs() > 2e-2 * ref.abs().clamp_min(1.0)).float().mean().item()
    assert bad == 0.0
    exact = (out == ref).float().mean().item()
    assert exact > 0.99, "ln_mod bit-exact fraction %g" % exact
    # scheduler step + re-pin
    lat = torch.randn(96, 64, generator=g).to(BF)
    v = torch.randn(96, 64, generator=g).to(BF)
    cond = torch.randn(32, 64, generator=g).to(BF)
    xd = lat.cuda().clone()
    ops.sched_step(xd, v.cuda(), -0.0371, n_noise_tokens=64, cond=cond.cuda())
    ref = dit_ref.euler_step(lat[:64], v[:64], 0.5, 0.5 - 0.0371)
    out = xd.float().cpu()
    assert torch.equal(out[64:], cond.float())
    assert (out[:64] - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()


def test_qkv_post():
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    H, S_txt, S_img = 2, 64, 136
    S = S_txt + S_img
    D = H * 128
    qkv = torch.randn(S, 3 * D, generator=g).to(BF)
    wq = (1 + 0.1 * torch.randn(128, generator=g)).to(BF)
    wk = (1 + 0.1 * torch.randn(128, generator=g)).to(BF)
    ids = torch.cat([torch.zeros(S_txt, 3), dit_ref.latent_image_ids(8, 17, offset_y=3)], 0)
    cos, sin = dit_ref.rope_tables(ids)
    S_pad = (S + 63) // 64 * 64
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda")
    Kh = torch.zeros_like(Qh)
    Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
    qd = qkv.cuda()
    ops.qkv_post(qd[S_txt:], 0, D, 2 * D, wq.cuda(), wk.cuda(), cos.cuda(), sin.cuda(), Qh, Kh, Vt, S_img, S_txt, H)
    ops.qkv_post(qd[:S_txt], 0, D, 2 * D, wq.cuda(), wk.cuda(), cos.cuda(), sin.cuda(), Qh, Kh, Vt, S_txt, 0, H)
    torch.cuda.synchronize()
    x = qkv.float()
    q = dit_ref.rms_norm(dit_ref._heads(x[:, :D], H), wq.float(), 1e-6, True)
    k = dit_ref.rms_norm(dit_ref._heads(x[:, D:2 * D], H), wk.float(), 1e-6, True)
    q = dit_ref._rb(dit_ref.apply_rope(q, cos, sin), True)
    k = dit_ref._rb(dit_ref.apply_rope(k, cos, sin), True)
    v = dit_ref._heads(x[:, 2 * D:], H)
    # q_scale: Q is multiplied before its (single) bf16 rounding
    Q2 = torch.zeros_like(Qh); K2 = torch.zeros_like(Kh); V2 = torch.zeros_like(Vt)
    ops.qkv_post(qd[S_txt:], 0, D, 2 * D, wq.cuda(), wk.cuda(), cos.cuda(), sin.cuda(), Q2, K2, V2, S_img, S_txt, H, q_scale=0.1275)
    torch.cuda.synchronize()
    qs = dit_ref._rb(dit_ref.apply_rope(dit_ref.rms_norm(dit_ref._heads(x[:, :D], H), wq.float(), 1e-6, True), cos, sin) * 0.1275, True)
    assert (Q2[:, S_txt:S].float().cpu() - qs[:, S_txt:]).abs().max().item() < 6e-3
    assert torch.equal(K2[:, S_txt:S].cpu(), Kh[:, S_txt:S].cpu())
    for name, got, ref in (("q", Qh[:, :S].float().cpu(), q), ("k", Kh[:, :S].float().cpu(), k)):
        assert (got - ref).abs().max().item() < 4e-2, name
        assert (got == ref).float().mean().item() > 0.98, name
    assert torch.equal(Vt[:, :, :S].float().cpu(), v.transpose(1, 2))
    assert Vt[:, :, S:].abs().max().item() == 0


@pytest.mark.parametrize("use_lora", [False, True])
def test_tiny_dit_forward_matches_oracle(use_lora):
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 64, 8 * 24 + 8 * 24 + 16
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF)
    pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    loras = None
    m = FluxDiT(sd, shape, device="cuda:0")
    if use_lora:
        la = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2)
        lb = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=3)
        loras = [(la, 1.0), (lb, 0.0)]  # reference keeps both injected, inactive one scaled by 0
        m.set_lora(loras)
    ref = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids,
                               loras=loras, emulate_bf16=True)
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
    out = m.forward(lat.cuda(), 0.4375).float().cpu()
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    mx = ref.abs().max().item()
    # bf16 network, 4 blocks: accumulation-order noise flips bf16 roundings; stay within ~3 bf16 ulp of max
    assert err < 0.03 * max(mx, 1.0), "tiny DiT forward err %g (ref max %g)" % (err, mx)
    if use_lora:
        base = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids)
        assert (base - ref).abs().max().item() > 5 * err, "LoRA branch must matter in this test"


@pytest.mark.parametrize("M,tok_off,lora", [(16384, 0, False), (16384 + 100, 64, True)])
def test_gemm_fused_qk_post_is_bit_identical_to_gemm_then_qkv_post(M, tok_off, lora):
    """utx_gemm_desc.qk_cols: the QKV projection's epilogue applies per-head RMSNorm + RoPE (+ the q scale) and writes head-major Q / K itself
    (one-wave-per-SIMD kernel: a wave owns a whole 128-column head); V columns take the plain epilogue and utx_qkv_post(skip_qk) transposes
    them.  Same arithmetic in the same order as GEMM -> utx_qkv_post (attention_processor.py:42-87): Q, K, V^T must be BIT-IDENTICAL,
    with a ragged last row tile, a token offset, and the LoRA K-segment in the accumulators."""
    ops = _ops()
    H, K, R = 2, 256, 64
    D = H * 128
    g = torch.Generator(device="cuda").manual_seed(M)
    x = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
    W = (torch.randn(3 * D, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(3 * D, device="cuda", generator=g).to(BF)
    wq = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
    wk = (1 + 0.1 * torch.randn(128, device="cuda", generator=g)).to(BF)
    S = tok_off + M
    S_pad = (S + 63) // 64 * 64
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 97, torch.arange(S) % 97], 1).float()
    cos, sin = [t.cuda().contiguous() for t in dit_ref.rope_tables(ids)]
    kw = {}
    if lora:
        T = (torch.randn(M, 3 * R, device="cuda", generator=g) / 8).to(BF)
        Bl = (torch.randn(3 * D, R, device="cuda", generator=g) / 4).to(BF)
        kw = dict(A2=T, B2=Bl, lora_n_limit=3 * D, lora_seg_n=D)
    qs = 0.1275
    outs = []
    for fused in (False, True):
        qkv = torch.full((M, 3 * D), 3.0, dtype=BF, device="cuda")
        Qh = torch.zeros(H, S_pad, 128, dtype=BF, device="cuda"); Kh = torch.zeros_like(Qh); Vt = torch.zeros(H, 128, S_pad, dtype=BF, device="cuda")
        qk = dict(cols=2 * D, tok_off=tok_off, eps=1e-6, q_scale=qs, wq=wq, wk=wk, cos=cos, sin=sin, Qh=Qh, Kh=Kh) if fused else None
        ops.gemm(x, W, bias=bias, out=qkv, qk_post=qk, **kw)
        ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, M, tok_off, H, q_scale=qs, skip_qk=fused)
        torch.cuda.synchronize()
        if fused:
            assert bool((qkv[:, : 2 * D] == 3.0).all()), "the fused epilogue must not write the q / k columns of C"
        outs.append((Qh, Kh, Vt))
    for name, a, b in zip("QKV", outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "%s differs between GEMM -> qkv_post and the fused epilogue" % name
    assert outs[0][0].float().abs().max() > 0.05


@pytest.mark.parametrize("use_lora", [False, True])
def test_last_block_pruning_keeps_the_consumed_rows_bit_identical(use_lora):
    """FluxDiT.set_output_rows(n): the texturing pipeline reads only the noise tokens' prediction (the condition tail of the latents is
    re-pinned before every transformer call: flux_piplines/texturing/pipeline.py:645,660,684), so the last block computes query / MLP /
    output projection for those rows only and the final norm + proj_out likewise.  The rows that are read must equal the unpruned
    forward BIT FOR BIT (same weights, same K order, same attention arithmetic per query row), with and without the LoRA K-segment."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd import _lib
    cfg = dit_ref.tiny_config(heads=2, double=1, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=1, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, n_noise, S_img = 64, 8 * 24, 8 * 24 + 8 * 24 + 16
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF)
    pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    m = FluxDiT(sd, shape, device="cuda:0")
    if use_lora:
        m.set_lora([(dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2), 1.0)])
    outs = {}
    try:
        _lib.set_option("UTX_ATTN_TAILSPLIT", 0)      # the key-split tail round picks different query blocks for the two launches
        for rows in (None, n_noise, 100):
            m.set_positions(txt_ids, img_ids)
            m.set_output_rows(rows)
            m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
            outs[rows] = m.forward(lat.cuda(), 0.4375).clone()
            torch.cuda.synchronize()
    finally:
        _lib.set_option("UTX_ATTN_TAILSPLIT", 1)
        m.set_output_rows(None)
    for rows in (n_noise, 100):
        assert torch.equal(outs[rows][:rows].view(torch.int16), outs[None][:rows].view(torch.int16)), "pruned forward differs on rows < %d" % rows
    assert outs[None].float().abs().max() > 0.1


def test_adapter_switching_module_copies_and_all_zero_weights():
    """(i) an adapter that carries a full x_embedder copy (peft modules_to_save, trainer.py:297-304) swaps it in while it is
    switched on, and the other adapter's copy when the weights flip (texture pass / delight pass, pipeline.py:245,263);
    (ii) all-zero weights after an active pass leave no stale adapter behind: the forward equals the base model's;
    (iii) two switched-on adapters that both carry a copy are refused."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=1, single=1, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 64, 192
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF)
    pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF)
    txt_ids, img_ids = torch.zeros(S_txt, 3), dit_ref.latent_image_ids(8, 24)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2)
    lb = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=3)
    for i, l in enumerate((la, lb)):
        l["__full__"] = {"x_embedder.weight": (sd["x_embedder.weight"].float() + 0.05 * torch.randn(sd["x_embedder.weight"].shape, generator=g)).to(BF),
                         "x_embedder.bias": (sd["x_embedder.bias"].float() + 0.1 * (i + 1)).to(BF)}
    m = FluxDiT(sd, shape, device="cuda:0")

    def run(loras):
        m.set_lora(loras)
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        out = m.forward(lat.cuda(), 0.4375).float().cpu()
        torch.cuda.synchronize()
        ref = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids,
                                   loras=loras, emulate_bf16=True)
        return out, ref

    o_t, r_t = run([(la, 1.0), (lb, 0.0)])
    o_d, r_d = run([(la, 0.0), (lb, 1.0)])
    o_0, r_0 = run([(la, 0.0), (lb, 0.0)])
    base = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids, emulate_bf16=True)
    assert torch.equal(r_0, base)
    for o, r in ((o_t, r_t), (o_d, r_d), (o_0, r_0)):
        assert (o - r).abs().max().item() < 0.03 * max(r.abs().max().item(), 1.0)
    assert (r_t - r_d).abs().max().item() > 0.2 and (r_t - base).abs().max().item() > 0.2    # the passes really differ
    o_e, _ = run([])
    assert torch.equal(o_e, o_0)
    with pytest.raises(ValueError):
        m.set_lora([(la, 1.0), (lb, 0.5)])


def test_sequence_parallel_plan_world1_matches_plain_forward():
    """the head-parallel (Ulysses) plan with a 1-rank group must reproduce the plain plan bit for bit (same kernels,
    the exchange degenerates to layout copies); multi-rank exchange logic is covered on CPU (tests/test_multigpu_cpu.py)."""
    import os
    import torch.distributed as dist
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        cfg = dit_ref.tiny_config(heads=2, double=1, single=2, joint_dim=64, pooled_dim=64)
        sd = dit_ref.make_synthetic_state_dict(cfg, seed=3)
        shape = FluxShape(num_heads=2, num_double=1, num_single=2, joint_dim=64, pooled_dim=64)
        S_txt, S_img = 64, 192
        g = torch.Generator().manual_seed(5)
        lat = torch.randn(S_img, 64, generator=g).to(BF).cuda()
        enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF).cuda()
        pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF).cuda()
        txt_ids, img_ids = torch.zeros(S_txt, 3), dit_ref.latent_image_ids(8, 24)
        outs = []
        for sp in (False, True):
            m = FluxDiT(sd, shape, device="cuda:0", sequence_parallel=sp)
            m.set_positions(txt_ids, img_ids)
            m.set_conditioning(enc, pooled, 3.5)
            outs.append(m.forward(lat, 0.5).clone())
            torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("P,H,S_loc,kb", [(2, 2, 192, False), (4, 3, 128, True), (4, 6, 3136, True), (8, 3, 6336, True)])
def test_attention_on_block_strided_operands_equals_the_contiguous_call(P, H, S_loc, kb):
    """utx_attn_fwd_bf16_blk: Q / K / V^T consumed from the sequence-parallel receive buffer [source rank][q | k | v][head][S_loc x 128] (ulysses.py) -- blocks
    of S_loc tokens, 3 H S_loc 128 elements apart -- must give the bits of the contiguous call on the head-major tensors: same tiles, same order.  Cases with
    the key-split tail round (6 heads x 12 544 tokens = 294 workgroups; 3 heads x 50 688 = 594: the 8-rank launch of the real model) and with the key
    multiplicity of the per-rank text tile (period = S_loc / 64 tiles)."""
    import ctypes as C
    from unitex_amd.flux import ops
    ctx = ops.get_ctx(0)
    lib = ctx.lib
    S, E = P * S_loc, S_loc * 128
    g = torch.Generator(device="cuda").manual_seed(P * 1000 + H)
    q = (torch.randn(H, S, 128, device="cuda", generator=g) * 0.1275).to(BF)
    k = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    v = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    vt = v.transpose(1, 2).contiguous()
    kbl, per = (2.0, S_loc // 64) if kb else (0.0, 0)
    ref = ops.attention(q, k, vt, S=S, scale=0.0, key_bias_log2=kbl, key_bias_period=per)
    buf = torch.empty(P, 3, H, E, dtype=BF, device="cuda")
    buf[:, 0] = q.view(H, P, E).transpose(0, 1)
    buf[:, 1] = k.view(H, P, E).transpose(0, 1)
    buf[:, 2] = vt.view(H, 128, P, S_loc).permute(2, 0, 1, 3).reshape(P, H, E)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    nbytes = int(lib.utx_attn_workspace_bytes(ctx.handle, H, S, S))
    work = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    bs = 3 * H * E
    rc = lib.utx_attn_fwd_bf16_blk(ctx.handle, C.c_void_p(buf[0, 0].data_ptr()), C.c_void_p(buf[0, 1].data_ptr()), C.c_void_p(buf[0, 2].data_ptr()),
                                   C.c_void_p(out.data_ptr()), E, 128, E, 128, E, S_loc, out.stride(0), H, S, S, 0.0, kbl, per,
                                   C.c_void_p(work.data_ptr()), nbytes, S_loc, bs, bs, bs, ctx.stream())
    assert rc == 0
    torch.cuda.synchronize()
    plan = (C.c_int * 4)()
    assert lib.utx_attn_plan(H, S, S, torch.cuda.get_device_properties(0).multi_processor_count, plan) == 0
    if H * ((S + 255) // 256) > 256:
        assert plan[2] > 1 and nbytes > 0, "this case is meant to run the key-split tail round"
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), "block-strided attention differs from the contiguous call"
    # refusals: a block that is not whole 64-key tiles, a sequence that is not whole blocks
    for bad_rows in (S_loc + 32, S_loc * P + 64):
        assert lib.utx_attn_fwd_bf16_blk(ctx.handle, C.c_void_p(buf[0, 0].data_ptr()), C.c_void_p(buf[0, 1].data_ptr()), C.c_void_p(buf[0, 2].data_ptr()),
                                         C.c_void_p(out.data_ptr()), E, 128, E, 128, E, S_loc, out.stride(0), H, S, S, 0.0, 0.0, 0,
                                         None, 0, bad_rows, bs, bs, bs, ctx.stream()) != 0


@pytest.mark.parametrize("zero_copy", [False, True])
def test_sequence_parallel_world1_through_rccl_collectives(monkeypatch, zero_copy):
    """RCCL on a one-GPU box: with UTX_SP_FORCE_A2A=1 a 1-rank NCCL group still issues every collective of the sequence-parallel plan -- per layer and
    head group an asynchronous all_to_all_single to itself on ProcessGroupNCCL's stream, work.wait() on the compute stream, then the HIP unpack / attention
    kernels launched through ctypes on that stream, and the same for the return exchange.  What it pins: the call pattern RCCL accepts (views of the
    group-major buffers, async_op), and the ORDERING between RCCL's stream and the kernels the C ABI launches (a missing dependency shows up as stale
    Q / K / V or stale outputs).  One rank sees every key in the plain order, so the result must equal the plain forward bit for bit, every time."""
    import os
    import torch.distributed as dist
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    monkeypatch.setenv("UTX_SP_FORCE_A2A", "1")
    monkeypatch.setenv("UTX_SP_ZERO_COPY", "1" if zero_copy else "0")      # the relayout pass behind the Q / K / V exchange (default), or attention on the receive buffer
    monkeypatch.setenv("UTX_SP_GROUPS", "2")
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        assert dist.get_backend() == "nccl"
        cfg = dit_ref.tiny_config(heads=4, double=2, single=3, joint_dim=64, pooled_dim=64)
        sd = dit_ref.make_synthetic_state_dict(cfg, seed=3)
        shape = FluxShape(num_heads=4, num_double=2, num_single=3, joint_dim=64, pooled_dim=64)
        S_txt, S_img = 64, 64 * 48
        g = torch.Generator().manual_seed(5)
        enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF).cuda()
        pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF).cuda()
        txt_ids, img_ids = torch.zeros(S_txt, 3), dit_ref.latent_image_ids(64, 48)
        plain = FluxDiT(sd, shape, device="cuda:0")
        spm = FluxDiT(sd, shape, device="cuda:0", sequence_parallel=True)
        for m in (plain, spm):
            m.set_positions(txt_ids, img_ids)
            m.set_conditioning(enc, pooled, 3.5)
        ex = spm.ex
        assert ex.force and ex.can_async and ex.G == 2 and ex.recv.data_ptr() != ex.send.data_ptr() and ex.o_recv.data_ptr() != ex.o.data_ptr()
        assert ex.zero_copy == zero_copy
        # round 4: the sequence-parallel step is replayed in C between the host's collectives (utx_plan_run_range over p["segments"]); the Python launch
        # list (run_plan) must give the same bits, and both the plain forward's
        p_sp = next(iter(spm._plans.values()))
        segs = p_sp.get("segments")
        assert p_sp.get("cplan") is not None and segs is not None and len(segs) == 2 * (2 + 3) + 1, "one range in front of every exchange start / attention + the tail"
        assert sum(1 for _, _, op in segs if op is not None and op[0] == "sp_attn") == 5 and segs[-1][2] is None
        assert int(spm.lib.utx_plan_size(p_sp["cplan"])) == segs[-1][1] and all(b0 <= b1 for b0, b1, _ in segs)
        for it in range(6):
            lat = torch.randn(S_img, 64, generator=g).to(BF).cuda()
            a = plain.forward(lat, 0.5 - 0.05 * it).clone()
            b = spm.forward(lat, 0.5 - 0.05 * it).clone()
            spm.run_plan(p_sp)                                  # the same step through the Python launch list (inputs are in the workspaces already)
            c = p_sp["ws"]["out"].clone()
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "iteration %d: the plan through RCCL differs from the plain forward" % it
            assert torch.equal(c.view(torch.int16), b.view(torch.int16)), "iteration %d: C-replayed ranges differ from the Python launch list" % it
    finally:
        if created:
            dist.destroy_process_group()


def _sp_worker(rank, world, port, q, zero_text=False, groups=1):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if groups < 0:      # negative: the zero-copy form of the exchange (UTX_SP_ZERO_COPY=1: attention reads the receive buffer) instead of the default relayout
        os.environ["UTX_SP_ZERO_COPY"] = "1"
        groups = -groups
    os.environ["UTX_SP_GROUPS"] = str(groups)      # head groups per rank whose exchanges are pipelined with attention (ulysses.pick_head_groups)
    os.environ["UTX_TXT_STREAM"] = "1"             # the opt-in two-stream form of the double blocks (off by default since round 5) stays covered here
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import dit_ref as R
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = R.tiny_config(heads=4, double=1, single=2, joint_dim=64, pooled_dim=64)
    sd = R.make_synthetic_state_dict(cfg, seed=3)
    shape = FluxShape(num_heads=4, num_double=1, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = (256 if zero_text else 64) * world, 192 * world
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(S_img, 64, generator=g).to(torch.bfloat16).cuda()
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(torch.bfloat16).cuda()
    pooled = (0.5 * torch.randn(1, 64, generator=g)).to(torch.bfloat16).cuda()
    periodic = (zero_text == "periodic")      # UTX_SP_KV_DEDUP=0: every rank's text copy stays among the keys (round 2's form, what zero copy and fp8 attention run)
    if periodic:
        os.environ["UTX_SP_KV_DEDUP"] = "0"
    if zero_text:      # the reference's conditioning: identical text tokens -> every rank carries 64 of them; round 6 default: the unpack keeps ONE copy among the keys
        enc.zero_()    # (weight 256 world / 64); periodic / zero copy: one text tile at the start of every rank's chunk of the gathered key sequence (key_bias_period)
    txt_ids, img_ids = torch.zeros(S_txt, 3), R.latent_image_ids(8 * world, 24)
    m = FluxDiT(sd, shape, device="cuda:0", sequence_parallel=True)
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning(enc, pooled, 3.5)
    if not zero_text:
        assert m.text_rows is None and not m.sp_kv_dedup and m.ex.S_k == m.ex.S
    elif periodic or os.environ.get("UTX_SP_ZERO_COPY", "0") == "1":
        assert m.text_rows == 64 and m.key_bias_period == (64 + 192) // 64 and abs(m.key_bias_log2 - 2.0) < 1e-6 and not m.sp_kv_dedup and m.ex.S_k == m.ex.S
    else:
        assert m.text_rows == 64 and m.sp_kv_dedup and m.key_bias_period == 0 and abs(m.key_bias_log2 - math.log2(4.0 * world)) < 1e-6
        assert m.ex.kv_text_rows == 64 and m.ex.S == world * 256 and m.ex.S_k == 64 + world * 192 and m.ex.k.shape[1] == m.ex.S_k
    assert m.ex.G == groups and m.ex.Hg * groups * world == 4 and m.overlap_text      # two streams in the double blocks under sequence parallelism as well
    assert m.ex.zero_copy == (os.environ.get("UTX_SP_ZERO_COPY", "0") == "1")
    i0, i1 = m.local_image_range(S_img)
    out_loc = m.forward(lat[i0:i1].contiguous(), 0.5).float().cpu()
    torch.cuda.synchronize()
    if m.sp_kv_dedup:
        # S queries over S_k < S keys: the launch must HAVE scratch (flag bytes of the 4 x 64 kernel at least), or it silently runs the 8 x 32 kernel unsplit --
        # utx_attn_workspace_bytes refused S_q > S_kv until the last session of round 6 while the launcher accepted it
        assert m.ex.S > m.ex.S_k and int(m.lib.utx_attn_workspace_bytes(m.ctx.handle, m.ex.Hg, m.ex.S, m.ex.S_k)) >= m.ex.Hg * (m.ex.S // 64)
        assert all(pl["ws"].get("attn_ws") is not None for pl in m._plans.values())
    parts = [torch.empty_like(out_loc) for _ in range(world)]
    dist.all_gather(parts, out_loc)
    if rank == 0:
        plain = FluxDiT(sd, shape, device="cuda:0")
        plain.text_dedup = False           # the reference semantics: every text token processed
        plain.set_positions(txt_ids, img_ids)
        plain.set_conditioning(enc, pooled, 3.5)
        ref = plain.forward(lat, 0.5).float().cpu()
        got = torch.cat(parts, 0)
        q.put((float((got - ref).abs().max()), float(ref.abs().max()), float((got != ref).float().mean())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,zero_text,groups", [(2, False, 1), (2, True, 1), (2, False, 2), (2, True, 2), (4, True, 1), (4, False, 1), (2, True, -2), (4, True, -1),
                                                    (2, "periodic", 2), (4, "periodic", 1)])
def test_sequence_parallel_two_ranks_match_unsharded_forward(world, zero_text, groups):
    """two processes share cuda:0 and exchange through gloo (host-staged all-to-all): the token-sharded /
    head-sharded FluxDiT plan -- real kernels, real slicing of ids / embeddings / latents -- against the plain forward.
    Differences can only come from the key order inside attention (fp32 summation order): a few bf16 ulps.
    groups = 2: the two heads of a rank as two head groups -- send buffer written through the two-level grouped addressing of
    utx_qkv_post, one all-to-all + unpack + attention launch + return exchange per group (the pipelined form, ulysses.py).
    world = 4: one head per rank (the H / P = 3 regime of 8 ranks on the real model has the same single-group control flow)."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_sp_worker, args=(r, world, port, q, zero_text, groups)) for r in range(world)]
    for p in procs:
        p.start()
    err, mx, frac = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert err <= 0.02 * max(mx, 1.0), "sequence-parallel forward differs: %g (ref max %g, %.3f of elements differ)" % (err, mx, frac)


@pytest.mark.parametrize("H,S_k,extra,kb", [(2, 1024, 64, 2.0), (3, 4096, 192, 3.0), (1, 640, 448, 0.0), (2, 1000, 88, 0.0)])
def test_attention_more_queries_than_keys(H, S_k, extra, kb):
    """S_q > S_kv (round 6: a sequence-parallel rank attends with ALL P x S_loc query rows over keys that carry the ranks' identical text rows once,
    utx_sp_unpack_qkv_dedup): q and k / vt are separate arrays, nothing ties a query row to a key row.  Against the oracle softmax with the key weight on
    tile 0, through both kernels (4 x 64 stream where it takes the shape, 8 x 32 loop), and row for row bit-identical to the S_q = S_kv launch."""
    from unitex_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(S_k + extra)
    S_q = S_k + extra
    Sq_pad, Sk_pad = (S_q + 63) // 64 * 64, (S_k + 63) // 64 * 64
    q = torch.randn(H, S_q, 128, generator=g).to(BF); k = torch.randn(H, S_k, 128, generator=g).to(BF); v = torch.randn(H, S_k, 128, generator=g).to(BF)
    ref = torch.empty(H, S_q, 128)
    for h in range(H):
        sc = (q[h].float() @ k[h].float().t()) / math.sqrt(128.0)
        sc[:, :64] += kb * math.log(2.0)
        ref[h] = torch.softmax(sc, dim=-1) @ v[h].float()
    Qh = torch.zeros(H, Sq_pad, 128, dtype=BF, device="cuda"); Qh[:, :S_q] = (q.float() * (1.4426950408889634 / math.sqrt(128.0))).to(BF).cuda()
    Kh = torch.zeros(H, Sk_pad, 128, dtype=BF, device="cuda"); Kh[:, :S_k] = k.cuda()
    Vt = torch.zeros(H, 128, Sk_pad, dtype=BF, device="cuda"); Vt[:, :, :S_k] = v.cuda().transpose(1, 2)
    prev = _lib.get_options().get("UTX_ATTN_Q64", 1)
    try:
        for q64 in (1, 0):
            _lib.set_option("UTX_ATTN_Q64", q64)
            out = ops.attention(Qh, Kh, Vt, S=S_k, S_q=S_q, scale=0.0, key_bias_log2=kb)
            same = ops.attention(Qh, Kh, Vt, S=S_k, S_q=S_k, scale=0.0, key_bias_log2=kb)
            torch.cuda.synchronize()
            assert out.shape == (S_q, H * 128) and torch.isfinite(out.float()).all()
            err = (out.float().cpu().view(S_q, H, 128).permute(1, 0, 2) - ref).abs().max().item()
            assert err < 3e-2, "S_q %d > S_kv %d: max-abs err %g (q64=%d)" % (S_q, S_k, err, q64)
            assert torch.equal(out[:S_k].view(torch.int16), same.view(torch.int16)), "rows 0 .. S_kv-1 must not depend on the query count (q64=%d)" % q64
    finally:
        _lib.set_option("UTX_ATTN_Q64", prev)
    # block-strided operands share their blocks between queries and keys: S_q > S_kv is refused there
    ctx = ops.get_ctx(0)
    import ctypes as C
    rc = ctx.lib.utx_attn_fwd_bf16_blk(ctx.handle, C.c_void_p(Qh.data_ptr()), C.c_void_p(Kh.data_ptr()), C.c_void_p(Vt.data_ptr()), C.c_void_p(out.data_ptr()),
                                       Qh.stride(0), 128, Kh.stride(0), 128, Vt.stride(0), Sk_pad, out.stride(0), H, Sk_pad + 64, Sk_pad, 0.0, 0.0, 0, None, 0,
                                       64, 64 * 128, 64 * 128, 64, ctx.stream())
    assert rc != 0


def test_attention_running_max_keeps_growing():
    """pathological order for the sum-checked softmax: the scores of every query grow steadily along the key axis, so the
    running max has to be re-centred again and again (slow path on most tiles, in either 32-key block)."""
    H, S = 2, 1536
    g = torch.Generator().manual_seed(11)
    q = torch.randn(H, S, 128, generator=g).to(BF)
    k = (0.05 * torch.randn(H, S, 128, generator=g))
    ramp = torch.linspace(0.0, 6.0, S)[None, :, None]                 # key j is aligned with a common direction, growing with j
    u = torch.nn.functional.normalize(torch.randn(H, 1, 128, generator=g), dim=-1)
    k = (k + ramp * u).to(BF)
    q = (q + 8.0 * u).to(BF)                                           # every query has a large positive component along u
    v = torch.randn(H, S, 128, generator=g).to(BF)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    for prescaled in (False, True):
        out = _run_attn(q, k, v, prescaled=prescaled)
        assert torch.isfinite(out).all()
        err = (out - ref).abs().max().item()
        assert err < 4e-2, "ramp attention err %g (prescaled=%s)" % (err, prescaled)


@pytest.mark.parametrize("S,ramp_max", [(64, 0.0), (512, 0.0), (1536, 6.0), (1536, 45.0), (1536, 110.0), (1024, 150.0)])
def test_attention_q64_kernel_and_repair_pass(S, ramp_max):
    """the 4 x 64 kernel (UTX_ATTN_Q64=1, the default since round 6; whole 64-key tiles, pre-scaled Q) against the oracle: plain inputs, the growing
    running maximum (45 log2-units above the first block: inside its 2^96 headroom, no re-centring at all), and ramps steep enough to leave the headroom
    (110: finite row sums beyond 2^96, 150: fp32 overflow inside the kernel) so that the repair pass with the 8 x 32 kernel has to rewrite the query blocks."""
    H = 2
    g = torch.Generator().manual_seed(S + int(ramp_max))
    q = torch.randn(H, S, 128, generator=g)
    k = torch.randn(H, S, 128, generator=g)
    if ramp_max > 0:
        u = torch.nn.functional.normalize(torch.randn(H, 1, 128, generator=g), dim=-1)
        k = 0.05 * k + torch.linspace(0.0, ramp_max, S)[None, :, None] * u
        q = q + 8.0 * u
    q, k = q.to(BF), k.to(BF)
    v = torch.randn(H, S, 128, generator=g).to(BF)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    from unitex_amd import _lib
    prev = _lib.get_options()["UTX_ATTN_Q64"]
    _lib.set_option("UTX_ATTN_Q64", 1)
    try:
        for prescaled in (False, True):
            out = _run_attn(q, k, v, prescaled=prescaled)
            assert torch.isfinite(out).all()
            err = (out - ref).abs().max().item()
            assert err < 4e-2, "4 x 64 attention err %g (S=%d ramp=%g prescaled=%s)" % (err, S, ramp_max, prescaled)
    finally:
        _lib.set_option("UTX_ATTN_Q64", prev)


def test_hip_graph_replay_is_bit_identical_to_the_eager_plan():
    """FluxDiT.capture_graph(): the whole per-step plan recorded into one HIP graph.  Replays must equal eager launches bit
    for bit, follow new latents / timesteps / conditioning (device-side inputs of the captured kernels), and a changed
    adapter set must drop the graph."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=4)
    shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 64, 8 * 24 + 8 * 24 + 16
    g = torch.Generator().manual_seed(9)
    lats = [torch.randn(S_img, 64, generator=g).to(BF).cuda() for _ in range(3)]
    encs = [(0.5 * torch.randn(S_txt, 64, generator=g)).to(BF).cuda() for _ in range(2)]
    pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF).cuda()
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    m = FluxDiT(sd, shape, device="cuda:0")
    m.set_lora([(dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2), 1.0)])
    m.set_positions(txt_ids, img_ids)
    cases = [(lats[0], 0.4375, encs[0]), (lats[1], 0.875, encs[0]), (lats[2], 0.125, encs[1])]
    eager = []
    for lat, t, enc in cases:
        m.set_conditioning(enc, pooled, 3.5)
        eager.append(m.forward(lat, t).clone())
    m.set_conditioning(encs[0], pooled, 3.5)
    m.capture_graph()
    assert m._graphs
    for (lat, t, enc), ref in zip(cases, eager):
        m.set_conditioning(enc, pooled, 3.5)
        out = m.forward(lat, t).clone()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), "graph replay differs from eager launches"
    m.set_lora([])
    assert not m._graphs, "a new adapter set must drop the captured graph"


@pytest.mark.parametrize("fp8", [False, True])
def test_c_side_plan_replay_is_bit_identical_to_the_python_launch_list(fp8):
    """utx_plan (csrc/plan.cpp; SURVEY 8b `utx_dit_step`): FluxDiT copies its per-step launch list into a utx_plan and forward() replays it with ONE
    C call -- the same launchers, descriptors, order and two streams as the Python loop over ctypes calls, so the outputs must be bit-identical, with
    LoRA, the text half on the second stream, last-block pruning, and (fp8) the quantiser entries; a new timestep / new latents are picked up
    through the device buffers the descriptors point at."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 64, 8 * 24 + 8 * 24 + 16
    g = torch.Generator().manual_seed(12)
    lat = torch.randn(S_img, 64, generator=g).to(BF).cuda(); lat2 = torch.randn(S_img, 64, generator=g).to(BF).cuda()
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF).cuda(); pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF).cuda()
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=fp8)
    m.set_text_stream(True)      # the opt-in two-stream form (off by default since round 5): fork / join entries in the C plan
    m.set_lora([(dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2), 1.0)])
    m.set_positions(txt_ids, img_ids)
    m.set_output_rows(192)
    m.set_conditioning(enc, pooled, 3.5)
    p = next(iter(m._plans.values()))
    assert p.get("cplan") is not None and m.overlap_text, "the plan was not compiled into a utx_plan"
    n_c = m.lib.utx_plan_size(p["cplan"])
    n_py = sum((len(d[0]) + len(d[1]) + 2) if (isinstance(fn, str) and fn == "par") else 1 for fn, d in p["plan"])
    assert n_c == n_py and n_c > 40      # fork + join entries per two-stream section
    a1 = m.forward(lat, 0.5).clone(); a2 = m.forward(lat2, 0.25).clone()
    torch.cuda.synchronize()
    cplan, p["cplan"] = p["cplan"], None          # the Python loop over the same list
    b1 = m.forward(lat, 0.5).clone(); b2 = m.forward(lat2, 0.25).clone()
    torch.cuda.synchronize()
    p["cplan"] = cplan
    assert torch.equal(a1.view(torch.int16), b1.view(torch.int16)) and torch.equal(a2.view(torch.int16), b2.view(torch.int16))
    assert not torch.equal(a1, a2) and a1.float().abs().max() > 0.1
    # and the list can still be captured into a HIP graph
    m.capture_graph()
    c1 = m.forward(lat, 0.5).clone()
    torch.cuda.synchronize()
    m.release_graph()
    assert torch.equal(c1.view(torch.int16), a1.view(torch.int16))
    # a range of the C plan must hold WHOLE two-stream sections: a range that begins behind a fork (on either stream's half) or ends in front of its join is refused
    # before anything of it is launched behind an unrecorded fork (round 5: a begin inside the MAIN half used to pass and its join ran unordered)
    import ctypes as C
    ents = _plan_entries(m.lib, cplan)
    fork = next(i for i, (k, _, _) in enumerate(ents) if k == 7)
    join = next(i for i, (k, _, _) in enumerate(ents) if k == 8)
    main_inside = next(i for i in range(fork + 1, join) if ents[i][1] == 0)
    side_inside = next(i for i in range(fork + 1, join) if ents[i][1] == 1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = C.c_int(-1)
    for b0 in (main_inside, side_inside, join):
        assert m.lib.utx_plan_run_range(cplan, b0, len(ents), st, C.byref(bad)) == -2 and bad.value == b0
    assert m.lib.utx_plan_run_range(cplan, 0, main_inside, st, C.byref(bad)) == -2      # ends inside the section (what was forked is joined before returning)
    assert m.lib.utx_plan_run_range(cplan, fork, join + 1, st, C.byref(bad)) == 0
    torch.cuda.synchronize()


def _plan_entries(lib, h):
    import ctypes as C
    out = []
    buf = C.create_string_buffer(1024)
    for i in range(lib.utx_plan_size(h)):
        kind, side = C.c_int(), C.c_int()
        n = lib.utx_plan_entry(h, i, C.byref(kind), C.byref(side), buf, 1024)
        assert n >= 0
        raw = bytes(buf.raw[:n])
        if kind.value == 4:          # attention: the scratch pointer / size are the caller's (one buffer per plan in either builder; sizes may differ per entry)
            raw = raw[:-16]
        out.append((kind.value, side.value, raw))
    return out


@pytest.mark.parametrize("full_width,lora,rows,fp8", [(False, True, None, False), (False, False, 192, False), (False, True, 192, False), (True, True, 4096, False),
                                                      (False, True, 192, True), (True, True, None, True), (True, True, 4096, True)])
def test_c_built_dit_plan_equals_the_python_built_one(full_width, lora, rows, fp8):
    """utx_dit_load (csrc/dit_plan.cpp): the C-side builder assembles a FLUX step from plain pointer structs.  It must produce the launch list
    FluxDiT builds -- same entries, same order, same streams, every descriptor BYTE-IDENTICAL (all pointers, strides, shapes, LoRA segments, gates,
    split-tail scratch) -- with and without LoRA, with last-block pruning, at a tiny shape and at full width (D = 3072, S = 9728, where the large-M
    kernels and the split tail are in play), on the bf16 and on the MX fp8 path (tiny: row-major scales + quantiser passes; full width: tile-packed scales,
    fp8 emitted by LayerNorm-modulation / GELU epilogues, pruned block on the second scratch); and replaying it gives the same bits."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    if full_width:
        cfg = dit_ref.FluxConfig(num_double=1, num_single=2)
        shape = FluxShape(num_double=1, num_single=2)
        S_txt = 512
        img_ids = torch.cat([dit_ref.latent_image_ids(32, 128), dit_ref.latent_image_ids(32, 128, offset_y=32), dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=32)], 0)
        enc = torch.zeros(S_txt, cfg.joint_dim).to(BF).cuda(); pooled = torch.zeros(1, cfg.pooled_dim).to(BF).cuda()       # zero text: the de-duplicated path (key bias)
        rank = 64
    else:
        cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
        shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
        S_txt = 64
        img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8), dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
        g0 = torch.Generator().manual_seed(4)
        enc = (0.5 * torch.randn(S_txt, 64, generator=g0)).to(BF).cuda(); pooled = (0.5 * torch.randn(1, 64, generator=g0)).to(BF).cuda()
        rank = 16
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(img_ids.shape[0], 64, generator=g).to(BF).cuda()
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=fp8)
    if lora:
        m.set_lora([(dit_ref.make_synthetic_lora(cfg, sd, rank=rank, seed=2), 1.0)])
    m.set_positions(torch.zeros(S_txt, 3), img_ids)
    m.set_output_rows(rows)
    m.set_conditioning(enc, pooled, 3.5)
    p = next(iter(m._plans.values()))
    ref = m.forward(lat, 0.5).clone()           # (first run: lazily sized scratch exists afterwards)
    if fp8:
        kinds = [e[0] for e in _plan_entries(m.lib, p["cplan"])]
        assert 5 in kinds or full_width, "the tiny fp8 plan carries quantiser passes"
    torch.cuda.synchronize()
    py = _plan_entries(m.lib, p["cplan"])
    h = m.build_c_dit_plan(p)
    assert h is not None
    try:
        cc = _plan_entries(m.lib, h)
        assert len(cc) == len(py), "entry count: C %d vs Python %d" % (len(cc), len(py))
        for i, (a, b) in enumerate(zip(cc, py)):
            assert a[0] == b[0] and a[1] == b[1], "entry %d: kind / stream %s vs %s" % (i, a[:2], b[:2])
            if a[2] != b[2]:
                diff = [j for j in range(0, len(a[2]), 8) if a[2][j:j + 8] != b[2][j:j + 8]]
                raise AssertionError("entry %d (kind %d): descriptor differs at byte offsets %s" % (i, a[0], diff[:8]))
        import ctypes as C
        bad = C.c_int(-1)
        assert m.lib.utx_dit_step(h, m.ctx.stream(), C.byref(bad)) == 0, bad.value
        torch.cuda.synchronize()
        got = next(iter(m._plans.values()))["ws"]["out"].clone()
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    finally:
        m.lib.utx_plan_free(h)


@pytest.mark.parametrize("S", [2830, 4096])
def test_attention_tail_split_matches_oracle_and_unsplit_launch(S):
    """more workgroups than CUs: the partly filled last round is cut along the keys (partial outputs + log-sum-exp, merged by
    a second kernel).  Rows of the full rounds must be bit-identical to the unsplit launch, tail rows within the usual
    tolerance of the oracle (one more bf16 rounding of the partial outputs)."""
    H = 24
    q, k, v = _mk_attn_inputs(H, S, seed=S, spike=True)
    from unitex_amd import _lib
    try:
        _lib.set_option("UTX_ATTN_TAILSPLIT", 0)
        plain = _run_attn(q, k, v)
        _lib.set_option("UTX_ATTN_TAILSPLIT", 1)
        split = _run_attn(q, k, v)
    finally:
        _lib.set_option("UTX_ATTN_TAILSPLIT", 1)
    ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
    assert (split - ref).abs().max().item() < 4e-2
    same = (split == plain).all(dim=-1)                       # [H, S] rows identical to the unsplit launch
    frac_changed = 1.0 - same.float().mean().item()
    assert 0.0 < frac_changed < 0.5, "the tail (and only the tail) goes through the split path: %g of the rows changed" % frac_changed
    assert (split - plain).abs().max().item() < 3e-2


@pytest.mark.parametrize("M,N,K,lora", [(5120, 3584, 1024, False), (4900, 3584, 1536, True), (8192, 2304, 3072, False)])
def test_gemm_tail_split_matches_unsplit_launch_and_oracle(M, N, K, lora):
    """opt-in UTX_GEMM_TAILSPLIT=1: with more 256 x 256 tiles than CUs the tiles of the last, partly filled round are cut
    along K (fp32 partials, the last split to arrive sums them in split order and runs the normal epilogue).  Deterministic,
    equal to the unsplit launch up to the fp32 summation order, tiles of the full rounds bit-identical; gated residual in place,
    ragged M and a LoRA K-segment included."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) / 4).to(BF)
    B = (torch.randn(N, K, generator=g) / math.sqrt(K) * 2).to(BF)
    bias = torch.randn(N, generator=g).to(BF)
    gate = (0.5 * torch.randn(N, generator=g)).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    kw, rkw = {}, {}
    if lora:
        R = 64
        T = (torch.randn(M, R, generator=g) / 8).to(BF)
        Bl = (torch.randn(N, R, generator=g) / 4).to(BF)
        kw = dict(A2=T.cuda(), B2=Bl.cuda(), lora_n_limit=N, lora_seg_n=N)
        rkw = dict(A2=T, B2=Bl, lora_seg=N, lora_limit=N)
    Ad, Bd, bd, gd = A.cuda(), B.cuda(), bias.cuda(), gate.cuda()
    outs = []
    from unitex_amd import _lib
    try:
        for mode in (0, 1, 1):
            _lib.set_option("UTX_GEMM_TAILSPLIT", mode)
            rd = res.cuda().clone()
            ops.gemm(Ad, Bd, bias=bd, out=rd, gate=gd, res=rd, **kw)
            torch.cuda.synchronize()
            outs.append(rd.clone())
    finally:
        _lib.set_option("UTX_GEMM_TAILSPLIT", 0)
    plain, s1, s2 = outs
    ref = _gemm_ref(A, B, bias, gate=gate, res=res, **rkw)
    _close(plain, ref, "unsplit gemm")
    _close(s1, ref, "tail-split gemm")
    assert torch.equal(s1.view(torch.int16), s2.view(torch.int16)), "split launch is not deterministic"
    changed = (s1 != plain).float().mean().item()
    assert 0.0 < changed < 0.02, "only roundings inside the tail tiles may move: %g of the elements differ" % changed


@pytest.mark.parametrize("P,G,Hg,S_loc", [(4, 3, 2, 128), (2, 2, 3, 64), (3, 1, 2, 64)])
def test_sequence_parallel_head_group_layouts(P, G, Hg, S_loc):
    """the pipelined exchanges' group-major buffers (ulysses.py, round 3): utx_qkv_post's TWO-LEVEL grouped addressing writes
    send [G][P][3][Hg][E] in place, utx_sp_unpack_o_cols puts one head group's return block at its columns of the consumer's rows;
    both against torch index arithmetic, bit-exact."""
    import ctypes as C
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(P * 100 + G * 10 + Hg)
    Hp, E = G * Hg, S_loc * 128
    H, D = P * Hp, P * Hp * 128
    qkv = torch.randn(S_loc, 3 * D, generator=g).to(BF).cuda()
    wq = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).cuda(); wk = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).cuda()
    ang = torch.rand(S_loc, 64, generator=g) * 6.28
    cos, sin = torch.cos(ang).cuda().contiguous(), torch.sin(ang).cuda().contiguous()
    Qh = torch.zeros(H, S_loc, 128, dtype=BF, device="cuda"); Kh = torch.zeros_like(Qh); Vt = torch.zeros(H, 128, S_loc, dtype=BF, device="cuda")
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, S_loc, 0, H, q_scale=0.1275)
    send = torch.zeros(G, P, 3, Hg, E, dtype=BF, device="cuda")
    flat = send.view(-1)
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, flat, flat[Hg * E:], flat[2 * Hg * E:], S_loc, 0, H, q_scale=0.1275,
                 heads_per_group=Hp, group_stride=3 * Hg * E, head_stride=E, row_stride_v=S_loc, sub_heads=Hg, sub_stride=P * 3 * Hg * E)
    torch.cuda.synchronize()
    assert torch.equal(send[:, :, 0].reshape(G, P, Hg, S_loc, 128), Qh.view(P, G, Hg, S_loc, 128).transpose(0, 1))
    assert torch.equal(send[:, :, 1].reshape(G, P, Hg, S_loc, 128), Kh.view(P, G, Hg, S_loc, 128).transpose(0, 1))
    assert torch.equal(send[:, :, 2].reshape(G, P, Hg, 128, S_loc), Vt.view(P, G, Hg, 128, S_loc).transpose(0, 1))
    # return side: group by group into strided rows
    out = torch.zeros(S_loc, 5 * D, dtype=BF, device="cuda")
    orecv = torch.randn(G, P, S_loc, Hg * 128, generator=g).to(BF).cuda()
    for gi in range(G):
        dst = out[:, gi * Hg * 128:]
        ctx.check(ctx.lib.utx_sp_unpack_o_cols(ctx.handle, C.c_void_p(orecv[gi].data_ptr()), P, Hg, S_loc, C.c_void_p(dst.data_ptr()), out.stride(0),
                                               Hp * 128, ctx.stream()))
    torch.cuda.synchronize()
    exp = orecv.permute(2, 1, 0, 3).reshape(S_loc, P * G * Hg * 128)          # row tok: [src][group][Hg*128] = head (src*Hp + g*Hg + hg)
    assert torch.equal(out[:, :D], exp) and out[:, D:].abs().max().item() == 0


@pytest.mark.parametrize("P,Hp,S_loc", [(4, 3, 128), (8, 3, 64), (2, 1, 192)])
def test_sequence_parallel_relayout_kernels(P, Hp, S_loc):
    """receive-side relayouts of the sequence-parallel exchange (utx_sp_unpack_qkv / utx_sp_unpack_o) and the grouped head
    addressing of utx_qkv_post (send side written in place) against torch index arithmetic: pure data movement, bit-exact."""
    import ctypes as C
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(P * 100 + Hp)
    S, E = P * S_loc, S_loc * 128
    recv = torch.randn(P, 3, Hp, E, generator=g).to(BF).cuda()
    q = torch.empty(Hp, S, 128, dtype=BF, device="cuda"); k = torch.empty_like(q); vt = torch.empty(Hp, 128, S, dtype=BF, device="cuda")
    ctx.check(ctx.lib.utx_sp_unpack_qkv(ctx.handle, C.c_void_p(recv.data_ptr()), P, Hp, S_loc, C.c_void_p(q.data_ptr()),
                                        C.c_void_p(k.data_ptr()), C.c_void_p(vt.data_ptr()), ctx.stream()))
    assert torch.equal(q.view(Hp, P, S_loc, 128), recv[:, 0].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
    assert torch.equal(k.view(Hp, P, S_loc, 128), recv[:, 1].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
    assert torch.equal(vt.view(Hp, 128, P, S_loc), recv[:, 2].view(P, Hp, 128, S_loc).permute(1, 2, 0, 3))
    if S_loc > 64:      # round 6: the ranks' identical 64 leading rows kept ONCE among the keys (utx_sp_unpack_qkv_dedup): [text of src 0 | the other tokens of src 0 .. P-1]
        T, I = 64, S_loc - 64
        S_k = T + P * I
        q2 = torch.full((Hp, S, 128), 7.0, dtype=BF, device="cuda"); k2 = torch.full((Hp, S_k, 128), 7.0, dtype=BF, device="cuda")
        vt2 = torch.full((Hp, 128, S_k), 7.0, dtype=BF, device="cuda")
        ctx.check(ctx.lib.utx_sp_unpack_qkv_dedup(ctx.handle, C.c_void_p(recv.data_ptr()), P, Hp, S_loc, T, C.c_void_p(q2.data_ptr()),
                                                  C.c_void_p(k2.data_ptr()), C.c_void_p(vt2.data_ptr()), ctx.stream()))
        rk, rv = recv[:, 1].view(P, Hp, S_loc, 128), recv[:, 2].view(P, Hp, 128, S_loc)
        assert torch.equal(q2, q)
        assert torch.equal(k2[:, :T], rk[0, :, :T]) and torch.equal(k2[:, T:].view(Hp, P, I, 128), rk[:, :, T:].permute(1, 0, 2, 3))
        assert torch.equal(vt2[:, :, :T], rv[0, :, :, :T]) and torch.equal(vt2[:, :, T:].view(Hp, 128, P, I), rv[:, :, :, T:].permute(1, 2, 0, 3))
        assert ctx.lib.utx_sp_unpack_qkv_dedup(ctx.handle, C.c_void_p(recv.data_ptr()), P, Hp, S_loc, 32, C.c_void_p(q2.data_ptr()),
                                               C.c_void_p(k2.data_ptr()), C.c_void_p(vt2.data_ptr()), ctx.stream()) != 0, "text_rows must be whole 64-key tiles"
    W = Hp * 128
    orecv = torch.randn(P, S_loc, W, generator=g).to(BF).cuda()
    out = torch.zeros(S_loc, 5 * P * W, dtype=BF, device="cuda")                  # strided rows, like the single-block cat buffer
    ctx.check(ctx.lib.utx_sp_unpack_o(ctx.handle, C.c_void_p(orecv.data_ptr()), P, Hp, S_loc, C.c_void_p(out.data_ptr()), out.stride(0), ctx.stream()))
    assert torch.equal(out[:, : P * W].view(S_loc, P, W), orecv.permute(1, 0, 2))
    assert out[:, P * W:].abs().max().item() == 0
    # send side: qkv_post with grouped head addressing == plain qkv_post followed by the pack permutation
    H, D = P * Hp, P * Hp * 128
    qkv = torch.randn(S_loc, 3 * D, generator=g).to(BF).cuda()
    wq = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).cuda(); wk = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).cuda()
    ang = torch.rand(S_loc, 64, generator=g) * 6.28
    cos, sin = torch.cos(ang).cuda().contiguous(), torch.sin(ang).cuda().contiguous()
    Qh = torch.zeros(H, S_loc, 128, dtype=BF, device="cuda"); Kh = torch.zeros_like(Qh); Vt = torch.zeros(H, 128, S_loc, dtype=BF, device="cuda")
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, S_loc, 0, H, q_scale=0.1275)
    send = torch.zeros(P, 3, Hp, E, dtype=BF, device="cuda")
    flat = send.view(-1)
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, flat, flat[Hp * E:], flat[2 * Hp * E:], S_loc, 0, H, q_scale=0.1275,
                 heads_per_group=Hp, group_stride=3 * Hp * E, head_stride=E, row_stride_v=S_loc)
    torch.cuda.synchronize()
    assert torch.equal(send[:, 0].view(P, Hp, S_loc, 128), Qh.view(P, Hp, S_loc, 128))
    assert torch.equal(send[:, 1].view(P, Hp, S_loc, 128), Kh.view(P, Hp, S_loc, 128))
    assert torch.equal(send[:, 2].view(P, Hp, 128, S_loc), Vt.view(P, Hp, 128, S_loc))


@pytest.mark.parametrize("use_lora", [False, True])
def test_text_token_dedup_matches_full_text_and_oracle(use_lora):
    """the reference's text tokens are 512 copies of ONE token (zero embeddings, zero ids: pipeline.py:538-543; SURVEY 7 last
    bullet).  With identical rows FluxDiT carries 64 of them and gives every key of the text tile the weight S_txt / 64 in the
    softmax (utx_attn_fwd_bf16_kb): the image-token output must equal the full-text plan up to fp32 summation order / bf16
    rounding, and both must match the oracle, which always processes all S_txt tokens.  Non-identical text rows must NOT dedup."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 256, 8 * 24 + 8 * 24 + 16
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = torch.zeros(S_txt, 64).to(BF)
    pooled = torch.zeros(1, 64).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    loras = [(dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2), 1.0)] if use_lora else None
    ref, inter = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids, loras=loras,
                                      emulate_bf16=True, return_intermediates=True)
    # the premise, on the oracle: the text rows stay identical through every double block
    for k, v in inter.items():
        if k.endswith("_ctx"):
            assert torch.equal(v, v[:1].expand_as(v)), "text rows diverged in %s" % k
    outs = {}
    for dedup in (True, False):
        m = FluxDiT(sd, shape, device="cuda:0")
        m.text_dedup = dedup
        if loras:
            m.set_lora(loras)
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        assert (m.text_rows == 64) == dedup and (abs(m.key_bias_log2 - 2.0) < 1e-6) == dedup        # 256 / 64 = 4-fold keys
        outs[dedup] = m.forward(lat.cuda(), 0.4375).float().cpu()
        torch.cuda.synchronize()
    mx = max(ref.abs().max().item(), 1.0)
    assert (outs[True] - outs[False]).abs().max().item() < 0.02 * mx
    for d in (True, False):
        assert (outs[d] - ref).abs().max().item() < 0.03 * mx
    # a prompt with real (row-dependent) embeddings is never deduplicated
    m = FluxDiT(sd, shape, device="cuda:0")
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning((0.5 * torch.randn(S_txt, 64, generator=g)).to(BF).cuda(), pooled.cuda(), 3.5)
    assert m.text_rows is None and m.key_bias_log2 == 0.0


def test_attention_key_multiplicity_equals_repeated_keys():
    """utx_attn_fwd_bf16_kb: 64 keys in tile 0 with key_bias_log2 = 3 must give the softmax over those keys repeated 8-fold (512
    text keys) followed by the image keys -- checked against the fp32 oracle on the expanded sequence, including the periodic
    form (a text tile at the start of every rank's chunk of the gathered sequence under sequence parallelism)."""
    ops = _ops()
    H, S_img = 4, 896            # two chunks of 64 + 448 = 512 keys = 8 tiles in the periodic case
    g = torch.Generator().manual_seed(5)
    qt, kt, vt_ = (torch.randn(H, 1, 128, generator=g) for _ in range(3))
    qi, ki, vi = (torch.randn(H, S_img, 128, generator=g) for _ in range(3))
    scale = (1.0 / math.sqrt(128.0))
    for period_chunks in (0, 2):
        if period_chunks == 0:
            q = torch.cat([qt.expand(H, 64, 128), qi], 1); k = torch.cat([kt.expand(H, 64, 128), ki], 1); v = torch.cat([vt_.expand(H, 64, 128), vi], 1)
            kf = torch.cat([kt.expand(H, 512, 128), ki], 1); vf = torch.cat([vt_.expand(H, 512, 128), vi], 1)
            bias, period = 3.0, 0
        else:      # two chunks [64 text | 448 image] each: 128 carried text keys stand for 512 -> 4-fold, bias 2
            half = S_img // 2
            q = torch.cat([qt.expand(H, 64, 128), qi[:, :half], qt.expand(H, 64, 128), qi[:, half:]], 1)
            k = torch.cat([kt.expand(H, 64, 128), ki[:, :half], kt.expand(H, 64, 128), ki[:, half:]], 1)
            v = torch.cat([vt_.expand(H, 64, 128), vi[:, :half], vt_.expand(H, 64, 128), vi[:, half:]], 1)
            kf = torch.cat([kt.expand(H, 512, 128), ki], 1); vf = torch.cat([vt_.expand(H, 512, 128), vi], 1)
            bias, period = 2.0, (64 + half) // 64
        q, k, v, kf, vf = (t.to(BF) for t in (q, k, v, kf, vf))
        S = q.shape[1]
        out = ops.attention((q.float() * scale * 1.4426950408889634).to(BF).cuda().contiguous(), k.cuda().contiguous(),
                            v.transpose(1, 2).contiguous().cuda(), S=S, scale=0.0, key_bias_log2=bias, key_bias_period=period)
        out = out.float().cpu().view(S, H, 128).permute(1, 0, 2)
        ref = dit_ref.sdpa(q.float(), kf.float(), vf.float(), em=False)          # queries of the carried rows, keys expanded
        assert (out - ref).abs().max().item() < 4e-2, "key multiplicity (period %d)" % period
