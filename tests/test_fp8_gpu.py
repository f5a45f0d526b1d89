"""OCP MX fp8 path (BASELINE configs[4] "fp8 MFMA weights"): the quantiser bit for bit against the CPU restatement of the format
(oracle/mx8_ref.py), the mx8 GEMM against the exact MX dot products, and the stated tolerance of the fp8 path against the bf16
path.  The reference has no fp8 mode (it runs FLUX in bf16, /root/reference/pipeline.py:96-103): this is the extension BASELINE.json
asks for, and its tolerance is stated here."""
import math

import pytest
import torch

from oracle import dit_ref, mx8_ref

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from unitex_amd.flux import ops
    return ops


def test_mx8_quantiser_bit_exact():
    from unitex_amd.flux import mx8
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 1024, generator=g) * torch.exp(4 * torch.randn(300, 1, generator=g))).to(BF)
    x[0, :64] = 0                      # all-zero blocks
    x[1, 7] = 3.0e38                   # near the top of the bf16 range
    x[2, :32] = 1.0e-38                # tiny block (scale clamps at 2^-127)
    x[3, 32:64] = 448.0 * 2.0 ** 5     # block maximum exactly on an e4m3 maximum after scaling
    q_ref, s_ref = mx8_ref.quantize(x)
    q, s = mx8.quantize_act(x.cuda(), ctx)
    torch.cuda.synchronize()
    assert torch.equal(s.cpu(), s_ref), "E8M0 scales"
    assert torch.equal(q.cpu(), q_ref), "e4m3 bytes"
    # the load-time weight quantiser is the same kernel
    qw, sw = mx8.quantize_weight(x.cuda(), ctx)
    assert torch.equal(qw.cpu(), q_ref) and torch.equal(sw.cpu(), s_ref)
    # strided rows (the single-block `cat` buffer)
    big = torch.zeros(300, 2048, dtype=BF); big[:, 512:1536] = x
    q2, s2 = mx8.quantize_act(big.cuda()[:, 512:1536], ctx)
    assert torch.equal(q2.cpu(), q_ref) and torch.equal(s2.cpu(), s_ref)


@pytest.mark.parametrize("M,N,K,K2", [(256, 256, 128, 0), (1000, 1152, 512, 64), (640, 768, 3072, 128)])
def test_mx8_gemm_matches_exact_mx_dot_products(M, N, K, K2):
    """utx_gemm_bf16 with mx8 operands: fp8 elements x E8M0 block scales through v_mfma_scale_f32_32x32x64_f8f6f4 -- every product
    is exact, so the only difference to the fp64-accumulated reference is fp32 summation order, then the bf16 rounding points of the epilogue: <= 2 bf16 ulp (1.6e-2 relative, the tolerance of the bf16 GEMM tests).
    Epilogues (bias, GELU from a column, column split, gated residual) and the bf16 LoRA segment ride on the same accumulators."""
    from unitex_amd.flux import mx8
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(M + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(BF)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, generator=g).to(BF)
    aq, a_s = mx8.quantize_act(A.cuda(), ctx)
    wq, w_s = mx8.quantize_weight(W.cuda(), ctx)
    acc = mx8_ref.gemm(aq.cpu(), a_s.cpu(), wq.cpu(), w_s.cpu())
    kw = dict(a_scale=a_s, b_scale=w_s)
    n_lim = (N // 256) * 128
    if K2:
        T = (torch.randn(M, K2, generator=g) / 8).to(BF)
        Bl = torch.zeros(N, K2, dtype=BF); Bl[:n_lim] = (torch.randn(n_lim, K2, generator=g) / 4).to(BF)
        kw.update(A2=T.cuda(), B2=Bl.cuda(), lora_n_limit=n_lim, lora_seg_n=n_lim)
        acc = acc + T.float() @ Bl.float().t()
    split = (N // 256) * 128
    c0 = torch.empty(M, split, dtype=BF, device="cuda"); c1 = torch.empty(M, N - split, dtype=BF, device="cuda")
    ops.gemm(aq, wq, bias=bias.cuda(), out=c0, gelu_from=split, n_split=split, C1=c1, **kw)
    out = torch.cat([c0, c1], 1).float().cpu()
    y = (acc + bias.float()).to(BF).float()
    y[:, split:] = dit_ref.gelu_tanh(y[:, split:]).to(BF).float()
    rel = ((out - y).abs() / y.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "mx8 GEMM (bias / GELU / split) vs exact MX reference: %g" % rel
    gate = torch.randn(N, generator=g).to(BF); res = torch.randn(M, N, generator=g).to(BF)
    r = res.cuda().clone()
    ops.gemm(aq, wq, bias=bias.cuda(), out=r, gate=gate.cuda(), res=r, **kw)
    yy = (acc + bias.float()).to(BF).float()
    yy = (res.float() + (gate.float() * yy).to(BF).float()).to(BF).float()
    rel = ((r.float().cpu() - yy).abs() / yy.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "mx8 GEMM (gated residual) vs exact MX reference: %g" % rel


def test_fp8_path_tolerance_against_bf16_path():
    """STATED TOLERANCE of the fp8 weight path: e4m3 keeps 3 mantissa bits, so one linear with both operands in MX fp8 deviates
    from the bf16 linear by ~ 2-4 % in relative Frobenius norm (measured 0.027-0.035 on N(0,1) data); bound 0.06.  Over a 2 + 2-block
    FLUX-shaped network with the five big linears in fp8 the output deviates by < 0.15 of max|out| from the bf16 oracle."""
    from unitex_amd.flux import mx8
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(3)
    M, N, K = 2048, 1024, 3072
    A = torch.randn(M, K, generator=g).to(BF).cuda(); W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).cuda()
    ref = ops.gemm(A, W).float()
    aq, a_s = mx8.quantize_act(A, ctx); wq, w_s = mx8.quantize_weight(W, ctx)
    out = ops.gemm(aq, wq, a_scale=a_s, b_scale=w_s).float()
    err = ((out - ref).norm() / ref.norm()).item()
    assert 0.005 < err < 0.06, "fp8 linear vs bf16 linear: relative Frobenius error %g" % err
    cfg = dit_ref.tiny_config(heads=2, double=2, single=2, joint_dim=64, pooled_dim=64)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_heads=2, num_double=2, num_single=2, joint_dim=64, pooled_dim=64)
    S_txt, S_img = 64, 8 * 24 + 8 * 24 + 16
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, 64, generator=g)).to(BF); pooled = (0.5 * torch.randn(1, 64, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    img_ids = torch.cat([dit_ref.latent_image_ids(8, 24), dit_ref.latent_image_ids(8, 24, offset_y=8),
                         dit_ref.latent_image_ids(4, 4, offset_x=24, offset_y=8)], 0)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2)
    ref = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids, loras=[(la, 1.0)],
                               emulate_bf16=True)
    outs = {}
    for fp8 in (False, True):
        m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=fp8)
        m.set_lora([(la, 1.0)])
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        outs[fp8] = m.forward(lat.cuda(), 0.4375).float().cpu()
        torch.cuda.synchronize()
    mx = max(ref.abs().max().item(), 1.0)
    assert (outs[False] - ref).abs().max().item() < 0.03 * mx
    d = (outs[True] - ref).abs().max().item()
    assert 1e-4 * mx < d < 0.15 * mx, "fp8-weight DiT vs bf16 oracle: %g of max|out| %g" % (d, mx)


def test_fp8_full_width_blocks_at_config5_per_view_shape():
    """BASELINE configs[4]: 2048^2 per view -> 16 384 noise + 16 384 control + 1024 dual + 512 text = 34 304 joint tokens, real FLUX width
    (D = 3072, 24 heads, rank-64 LoRA), the five big linears of every block on MX fp8 operands; depth cut to 1 + 1 blocks.
    The fp32 oracle needs minutes at this size, so the chain is: bf16 HIP path vs oracle at full width (tests/test_fullsize_gpu.py, S = 9728,
    3e-2 max|out|) + fp8 HIP path vs bf16 HIP path HERE, with the bound derived from the measured error of ONE fp8 linear on this data:
    a block output passes through at most 5 fp8 linears whose relative errors e_lin add in quadrature at unit gain -> relative Frobenius
    error of the forward <= 4 e_lin (sqrt(5) = 2.2 with a factor ~2 for the gains of the gates / the LayerNorm), and max |d| <= 0.15 max|out|
    (the stated fp8 tolerance of test_fp8_path_tolerance_against_bf16_path)."""
    from unitex_amd.flux import mx8
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    ops = _ops()
    ctx = ops.get_ctx(0)
    cfg = dit_ref.FluxConfig(num_double=1, num_single=1)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_double=1, num_single=1)
    S_txt = 512
    img_ids = torch.cat([dit_ref.latent_image_ids(128, 128), dit_ref.latent_image_ids(128, 128, offset_y=128),
                         dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=128)], 0)
    S_img = img_ids.shape[0]
    assert S_txt + S_img == 34304
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = torch.zeros(S_txt, cfg.joint_dim).to(BF); pooled = torch.zeros(1, cfg.pooled_dim).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=2)
    outs, census = {}, {}
    for fp8 in (False, True):
        m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=fp8)
        m.set_lora([(la, 1.0)])
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        outs[fp8] = m.forward(lat.cuda(), 0.4375).float()
        torch.cuda.synchronize()
        census[fp8] = m.gemm_census()
        if fp8:
            # e_lin on the live activation of the first fp8 linear (LayerNorm-modulated hidden states x the QKV weight of the double block)
            ws = next(iter(m._plans.values()))["ws"]
            A = ws["xn"][S_txt:].clone(); W = m.double[0]["qkv_x.w"]
            ref = ops.gemm(A, W).float()
            aq, a_s = mx8.quantize_act(A, ctx); wq, w_s = mx8.quantize_weight(W, ctx)
            e_lin = ((ops.gemm(aq, wq, a_scale=a_s, b_scale=w_s).float() - ref).norm() / ref.norm()).item()
        del m
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[True]).all()
    mxo = max(outs[False].abs().max().item(), 1.0)
    d = outs[True] - outs[False]
    rel_f = (d.norm() / outs[False].norm()).item()
    print("\n[fp8 full width, S = 34304] e_lin %.4f | forward fp8 vs bf16: rel Frobenius %.4f (bound 4 e_lin = %.4f), max|d| %.4g = %.4f of max|out| %.3g, "
          "mean|d| %.4g | GEMM kernels bf16 %s fp8 %s" % (e_lin, rel_f, 4 * e_lin, d.abs().max().item(), d.abs().max().item() / mxo, mxo,
                                                         d.abs().mean().item(), census[False], census[True]))
    assert 0.005 < e_lin < 0.06
    assert rel_f < 4 * e_lin, "fp8 forward vs bf16 forward: relative Frobenius error %g against 4 e_lin = %g" % (rel_f, 4 * e_lin)
    assert 1e-4 * mxo < d.abs().max().item() < 0.15 * mxo
    # fused quantisation (default): LayerNorm-modulation and the GELU epilogues write the next GEMM's activation AS fp8 (utx_ln_mod_desc.q,
    # utx_gemm_desc.q_out) -- the same bytes as bf16 + utx_quant_mx8_packed, so the forward must not change by a single bit
    import os
    os.environ["UTX_FP8_FUSE_QUANT"] = "0"
    try:
        m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
        m.set_lora([(la, 1.0)])
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        unfused = m.forward(lat.cuda(), 0.4375).float()
        torch.cuda.synchronize()
        n_quant = sum(1 for fn, _ in _flat(m) if fn == "quant_mx8")
        del m
    finally:
        del os.environ["UTX_FP8_FUSE_QUANT"]
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
    m.set_lora([(la, 1.0)])
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
    n_quant_fused = sum(1 for fn, _ in _flat(m) if fn == "quant_mx8")
    del m
    torch.cuda.empty_cache()
    assert n_quant == 5 and n_quant_fused == 1, (n_quant, n_quant_fused)      # only the attention output of the single block keeps a quantiser pass
    assert torch.equal(unfused, outs[True]), "fused fp8 quantisation changed the forward: max |d| %g" % (unfused - outs[True]).abs().max().item()
    # last-block pruning on the fp8 path (set_output_rows: queries / MLP / out-projection of the last block for the 16 384 noise tokens only, their
    # activations quantised as a matrix of their own): MX quantisation is row-local and every output element is accumulated over ascending K by the
    # same instruction, so the rows that are read must not change by a single bit (tail splits off: they pick different tiles for the two shapes)
    from unitex_amd import _lib
    n_noise = 16384
    _lib.set_option("UTX_ATTN_TAILSPLIT", 0); _lib.set_option("UTX_GEMM_STREAMK", 0)
    try:
        res = {}
        for rows in (None, n_noise):
            m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
            m.set_lora([(la, 1.0)])
            m.set_positions(txt_ids, img_ids)
            m.set_output_rows(rows)
            m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
            res[rows] = m.forward(lat.cuda(), 0.4375).clone()
            torch.cuda.synchronize()
            if rows:
                ws = next(iter(m._plans.values()))["ws"]
                assert "aq2" in ws, "the pruned last block did not take the fp8 path"
            del m
            torch.cuda.empty_cache()
    finally:
        _lib.set_option("UTX_ATTN_TAILSPLIT", 1); _lib.set_option("UTX_GEMM_STREAMK", 1)
    assert torch.equal(res[n_noise][:n_noise].view(torch.int16), res[None][:n_noise].view(torch.int16)), "fp8 pruned forward differs on the consumed rows"


def test_mx8_packed_scale_quantiser_matches_the_rowmajor_one():
    """utx_quant_mx8_packed: the same q bytes and the same E8M0 values as utx_quant_mx8 (= oracle/mx8_ref.quantize, bit for bit), scale bytes in
    the tile-packed order [K/128][row block][32][4] x 4 bytes; ragged row count (the last row block partly filled), strided rows, a scratch
    buffer larger than the matrix in both directions."""
    from unitex_amd.flux import mx8
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(1)
    for M, K in ((300, 1024), (256, 128), (1000, 3072)):
        x = (torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, 1, generator=g))).to(BF)
        x[0, :64] = 0
        q_ref, s_ref = mx8_ref.quantize(x)
        q, sp = mx8.quantize_act(x.cuda(), ctx, packed=True)
        torch.cuda.synchronize()
        assert torch.equal(q.cpu(), q_ref)
        assert torch.equal(sp.rowmajor().cpu(), s_ref), "packed scales, M=%d K=%d" % (M, K)
        big = torch.zeros(M, K + 512, dtype=BF); big[:, 256:256 + K] = x
        buf = mx8.PackedScales(mx8.packed_scale_buffer(M + 700, K + 256, "cuda"), M, K)
        q2 = torch.empty(M + 5, K + 128, dtype=torch.uint8, device="cuda")
        mx8.quantize_act(big.cuda()[:, 256:256 + K], ctx, out=(q2[:M, :K], buf))
        assert torch.equal(q2[:M, :K].cpu(), q_ref) and torch.equal(buf.rowmajor().cpu(), s_ref)


@pytest.mark.parametrize("M,N,K", [(1100, 512, 512), (4096, 3072, 3072), (13376, 3072, 3072), (6400, 9216, 1024), (2048, 256, 128)])
def test_mx8_one_wave_per_simd_kernel_bit_identical_to_the_tiled_mx_kernel_and_exact_on_sampled_rows(M, N, K):
    """utx_gemm_desc.mx8 = 2 (gemm_w4.hip, MX): 256 x 256 persistent tiles, 32 scaled MFMAs per 128-k K-tile, tile-packed scales.  Both MX kernels
    accumulate every output element over ascending K with the same instruction -> BIT-IDENTICAL results on every epilogue (bias + GELU + column
    split; gated residual), ragged M included; sampled rows against the exact MX dot products (oracle/mx8_ref.py).  With the split tail round
    (13376 x 3072 = 636 tiles = 2.48 rounds) the tail tiles differ by fp32 summation order only."""
    from unitex_amd import _lib
    from unitex_amd.flux import mx8
    ops = _ops()
    ctx = ops.get_ctx(0)
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(BF).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).cuda()
    bias = torch.randn(N, generator=g).to(BF).cuda()
    gate = torch.randn(N, generator=g).to(BF).cuda()
    res = torch.randn(M, N, generator=g).to(BF).cuda()
    aq, a_s = mx8.quantize_act(A, ctx); wq, w_s = mx8.quantize_weight(W, ctx)
    _, a_p = mx8.quantize_act(A, ctx, packed=True); _, w_p = mx8.quantize_weight(W, ctx, packed=True)
    split = N // 2 if (N // 2) % 256 == 0 else N
    outs = {}
    _lib.set_option("UTX_GEMM_STREAMK", 0)
    try:
        for tag, sa, sb in (("tiled", a_s, w_s), ("w4", a_p, w_p)):
            c0 = torch.zeros(M, split, dtype=BF, device="cuda"); c1 = torch.zeros(M, max(N - split, 1), dtype=BF, device="cuda")
            kw = dict(a_scale=sa, b_scale=sb)
            if split < N:
                ops.gemm(aq, wq, bias=bias, out=c0, gelu_from=split, n_split=split, C1=c1, **kw)
            else:
                ops.gemm(aq, wq, bias=bias, out=c0, **kw)
            r = res.clone()
            ops.gemm(aq, wq, bias=bias, out=r, gate=gate, res=r, **kw)
            torch.cuda.synchronize()
            outs[tag] = (c0, c1, r)
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
    for i, nm in enumerate(("bias / GELU / split: C", "C1", "gated residual")):
        a, b = outs["tiled"][i], outs["w4"][i]
        assert torch.equal(a, b), "%s: one-wave-per-SIMD MX kernel differs from the tiled MX kernel: max |d| %g in %d elements" % (
            nm, (a.float() - b.float()).abs().max().item(), (a != b).sum().item())
    # the default launch (split tail round where the model splits): same up to fp32 summation order in the tail tiles
    r2 = res.clone()
    ops.gemm(aq, wq, bias=bias, out=r2, gate=gate, res=r2, a_scale=a_p, b_scale=w_p)
    torch.cuda.synchronize()
    dd = (r2.float() - outs["w4"][2].float()).abs()
    assert (dd / outs["w4"][2].float().abs().clamp_min(1.0)).max().item() <= 3.2e-2 and (dd > 0).float().mean().item() < 0.02
    # sampled rows vs the exact MX dot products
    rows = torch.randperm(M, generator=g)[:96].sort().values
    acc = mx8_ref.gemm(aq.cpu()[rows], a_s.cpu()[rows], wq.cpu(), w_s.cpu())
    y = (acc + bias.float().cpu()).to(BF).float()
    if split < N:
        y[:, split:] = dit_ref.gelu_tanh(y[:, split:]).to(BF).float()
        got = torch.cat([outs["w4"][0], outs["w4"][1]], 1).float().cpu()[rows]
    else:
        got = outs["w4"][0].float().cpu()[rows]
    rel = ((got - y).abs() / y.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "MX one-wave-per-SIMD GEMM vs exact MX reference: %g" % rel


def test_mx8_packed_form_refuses_what_it_cannot_do():
    """mx8 = 2 has no LoRA K-segment and no 128-column boundaries: the library refuses (-2 -> exception) instead of dropping them."""
    from unitex_amd.flux import mx8
    ops = _ops()
    ctx = ops.get_ctx(0)
    A = torch.randn(512, 256).to(BF).cuda(); W = torch.randn(512, 256).to(BF).cuda()
    aq, a_p = mx8.quantize_act(A, ctx, packed=True); wq, w_p = mx8.quantize_weight(W, ctx, packed=True)
    T = torch.randn(512, 64).to(BF).cuda(); Bl = torch.randn(512, 64).to(BF).cuda()
    with pytest.raises(Exception):
        ops.gemm(aq, wq, a_scale=a_p, b_scale=w_p, A2=T, B2=Bl)
    with pytest.raises(Exception):
        c1 = torch.empty(512, 384, dtype=BF, device="cuda")
        ops.gemm(aq, wq, out=torch.empty(512, 128, dtype=BF, device="cuda"), a_scale=a_p, b_scale=w_p, n_split=128, C1=c1)
    torch.cuda.synchronize()


def _flat(m):
    def walk(ops_):
        for fn, d in ops_:
            if isinstance(fn, str) and fn == "par":
                yield from walk(d[0]); yield from walk(d[1])
            else:
                yield fn, d
    return list(walk(next(iter(m._plans.values()))["plan"]))


@pytest.mark.parametrize("fp8_attention", [False, True])
def test_sequence_parallel_single_blocks_run_their_projection_in_fp8(fp8_attention):
    """VERDICT r3 missing #1 / DESIGN 9 (v): under sequence parallelism the single blocks' [q|k|v|mlp] projection is cut at column 3D (q|k|v first, the MLP
    half beside the all-to-all) -- round 3 ran both halves in bf16 even in fp8 mode, i.e. configs[4] ("8 x MI355X, fp8 weights") would have run 38 of 57
    blocks' biggest GEMM in bf16.  Now both halves take the MX kernel on row slices of the same quantised weight.  One rank (a 1-rank RCCL group with every
    collective issued to itself) sees every key in the plain order and every output element keeps its K order, so the sequence-parallel fp8 forward must
    equal the plain fp8 forward BIT FOR BIT (split tail rounds off: the two forms launch different shapes).  fp8_attention: the opt-in MX fp8 attention under
    sequence parallelism -- each head group's Q / K / V^T quantised behind its unpack, utx_attn_fwd_fp8 over the group's heads -- against the plain fp8-attn
    forward (per-head results do not depend on how many heads a launch carries: bit for bit as well)."""
    import os
    import torch.distributed as dist
    from unitex_amd import _lib
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    os.environ["UTX_SP_FORCE_A2A"] = "1"
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29850 + os.getpid() % 100))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        cfg = dit_ref.FluxConfig(num_double=1, num_single=2)
        sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
        shape = FluxShape(num_double=1, num_single=2)
        S_txt, img_ids = 512, dit_ref.latent_image_ids(64, 64)
        S_img = img_ids.shape[0]                      # 4096 image tokens + 64 de-duplicated text rows = 4160 = 65 x 64
        g = torch.Generator().manual_seed(9)
        lat = torch.randn(S_img, 64, generator=g).to(BF).cuda()
        enc = torch.zeros(S_txt, cfg.joint_dim).to(BF).cuda()
        pooled = torch.zeros(1, cfg.pooled_dim).to(BF).cuda()
        _lib.set_option("UTX_GEMM_STREAMK", 0)
        outs = {}
        for name, sp in (("plain", False), ("sp", True)):
            m = FluxDiT(sd, shape, device="cuda:0", sequence_parallel=sp, fp8_weights=True, fp8_attention=fp8_attention)
            m.set_positions(torch.zeros(S_txt, 3), img_ids)
            m.set_conditioning(enc, pooled, 3.5)
            assert m.fp8_attention == fp8_attention
            if fp8_attention and not sp:
                assert sum(1 for fn, _ in _flat(m) if isinstance(fn, str) and fn == "attn8") == 3
            descs = [d for fn, d in _flat(m) if fn is m.lib.utx_gemm_bf16 and d.M >= 4096]
            big = [d for d in descs if d.N in (9216, 12288, 21504)]
            if sp:
                # per single block: the q|k|v half (N = 9216) and the MLP half (N = 12288), both on MX operands with tile-packed scales; the MLP half hands
                # its GELU output over as fp8 (q_out) to the fp8 out-projection
                sgl = [d for d in big if d.N in (9216, 12288) and d.K == 3072]
                assert sum(1 for d in sgl if d.N == 12288 and d.mx8 == 2 and d.q_out) == 2 + 1, [(d.N, d.mx8) for d in sgl]      # + the double block's MLP up-projection
                assert sum(1 for d in sgl if d.N == 9216 and d.mx8 == 2) >= 2 + 1            # 2 single blocks + the double block's image-side q|k|v
                assert m.ex.force and m.ex.can_async
            else:
                assert sum(1 for d in big if d.N == 21504 and d.mx8 == 2) == 2
            outs[name] = [m.forward(lat, 0.5 - 0.1 * i).clone() for i in range(3)]
            torch.cuda.synchronize()
            del m
        for a, b in zip(outs["plain"], outs["sp"]):
            assert torch.isfinite(a.float()).all()
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "sequence-parallel fp8 forward differs from the plain fp8 forward: max |d| %g" % (a.float() - b.float()).abs().max().item()
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
        os.environ.pop("UTX_SP_FORCE_A2A", None)
        if created:
            dist.destroy_process_group()
