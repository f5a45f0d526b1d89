"""GPU parity at BASELINE.json's full sizes (S = 50 688 joint tokens, D = 3072).

The oracle cannot run whole tensors of this size in seconds, so each test uses what the domain offers:
  * attention: sampled query rows against the oracle over ALL keys (exact same arithmetic contract as the small
    tests), plus two size-independent properties -- constant-V rows (softmax weights sum to 1) and invariance of the
    result under a permutation of the key/value order (online softmax must not depend on tile order);
  * GEMM: the three kernels (8-phase 256^2, 2-barrier 256^2, 128^2) accumulate every output in the same K order
    with the same MFMA, so they must agree BIT FOR BIT on every epilogue; sampled rows against the oracle.
"""
import math
import os

import pytest
import torch

from oracle import dit_ref

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
S_FULL, H_FULL = 50688, 24


def _ops():
    from unitex_amd.flux import ops
    return ops


@pytest.mark.parametrize("S", [S_FULL, 49664])     # BASELINE configs[1] texture pass (50 688 joint tokens) and configs[2] delight pass (49 664: no dual image)
def test_attention_full_size_sampled_rows_and_properties(S):
    ops = _ops()
    H = 4                     # 4 of the 24 heads: same per-head work, 1/6 of the memory
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    v = torch.randn(H, S, 128, device="cuda", generator=g).to(BF)
    k[:, 40000] = (q[:, 123].float() * 2.5).to(BF)          # a late spike: forces the re-centre path mid-sequence
    vt = v.transpose(1, 2).contiguous()
    out = ops.attention(q, k, vt, S=S)                      # [S, H*128]
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    rows = torch.tensor([0, 1, 31, 123, 255, 256, 4097, 25000, 40000, S - 1])
    for h in range(H):
        ref = dit_ref.sdpa(q[h:h + 1, rows].float().cpu(), k[h:h + 1].float().cpu(), v[h:h + 1].float().cpu(), em=False)[0]
        got = out[rows][:, h * 128:(h + 1) * 128].float().cpu()
        err = (got - ref).abs().max().item()
        assert err < 3e-2, "head %d: sampled-row max-abs err %g" % (h, err)   # same bound as the small-S tests
    # property 1: V == 1 -> every output is sum(p)/sum(p) = 1 (P is bf16 in the PV product, l is fp32: <= 2^-8 rel)
    ones_t = torch.ones(H, 128, S, dtype=BF, device="cuda")
    o1 = ops.attention(q, k, ones_t, S=S).float()
    assert (o1 - 1.0).abs().max().item() < 8e-3
    # property 2: key/value order does not matter (tile-order independence of the online softmax)
    perm = torch.randperm(S, device="cuda", generator=g)
    o2 = ops.attention(q, k[:, perm].contiguous(), vt[:, :, perm].contiguous(), S=S)
    torch.cuda.synchronize()
    d = (o2.float() - out.float()).abs().max().item()
    assert d < 2e-2, "permutation changed the result by %g" % d


def test_attention_full_size_key_multiplicity_equals_the_expanded_sequence():
    """text-token de-duplication at the bench's size: 64 text rows whose keys carry weight 2^3 + 50 176 image rows (S = 50 240 executed) must give
    what the literal sequence gives (each text key 8 times: S = 50 688) -- for the image queries and for the text queries (softmax over n
    identical keys = one key with weight n; the reference feeds identical prompt embeddings, pipeline.py:556-567)."""
    ops = _ops()
    H, S_d, n_txt, mult = 2, 50240, 64, 8
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(H, S_d, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(H, S_d, 128, device="cuda", generator=g).to(BF)
    v = torch.randn(H, S_d, 128, device="cuda", generator=g).to(BF)
    out_d = ops.attention(q, k, v.transpose(1, 2).contiguous(), S=S_d, key_bias_log2=math.log2(mult))
    exp = lambda t: torch.cat([t[:, :n_txt].repeat(1, mult, 1), t[:, n_txt:]], dim=1).contiguous()
    qe, ke, ve = exp(q), exp(k), exp(v)
    out_e = ops.attention(qe, ke, ve.transpose(1, 2).contiguous(), S=S_d + n_txt * (mult - 1))
    torch.cuda.synchronize()
    d_img = (out_d[n_txt:].float() - out_e[n_txt * mult:].float()).abs().max().item()
    d_txt = (out_d[:n_txt].float() - out_e[:n_txt].float()).abs().max().item()
    assert d_img < 2e-2 and d_txt < 2e-2, "de-duplicated vs expanded sequence: image rows %g, text rows %g" % (d_img, d_txt)


def _set_tile(v):
    from unitex_amd import _lib
    _lib.set_option("UTX_GEMM_TILE", 0 if v is None else int(v))


def _gemm_variants(fn):
    """one result per GEMM kernel.  The kernels promise identical bits for the same summation order, so the split tail round of the
    default kernel (UTX_GEMM_STREAMK: fp32 partial sums over K ranges, its own test below) is off here."""
    from unitex_amd import _lib
    outs = {}
    _lib.set_option("UTX_GEMM_STREAMK", 0)
    try:
        for tile in ("128", "2562", "256", "2560", "2564"):     # 2564 = the persistent one-wave-per-SIMD kernel (gemm_w4.hip), the default for the large shapes; 2560 = the persistent 8-wave kernel (gemm_pers.hip)
            _set_tile(tile)
            outs[tile] = fn()
            torch.cuda.synchronize()
        # round 6: the one-wave-per-SIMD kernel's steady-state K loop is a generated instruction stream (UTX_GEMM_FASTK=1, the default: what "2564" above ran);
        # the same kernel on hipcc's own loop must give the same bits
        assert _lib.get_options()["UTX_GEMM_FASTK"] == 1
        _lib.set_option("UTX_GEMM_FASTK", 0)
        _set_tile("2564")
        slow = fn()
        torch.cuda.synchronize()
        assert _same_bits(slow, outs["2564"]), "gemm256_w4_kernel: generated K loop and hipcc's K loop disagree"
    finally:
        _set_tile(None)
        _lib.set_option("UTX_GEMM_STREAMK", 1)
        _lib.set_option("UTX_GEMM_FASTK", 1)
    return outs


def _same_bits(a, b):
    return torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("M,N,K,K2,kind", [(50240, 3072, 15360, 64, "gate"), (13376, 3072, 12288, 64, "gate"), (13001, 3072, 12288, 64, "gelu"),
                                           (6144, 3072, 15360, 0, "gate"), (13376, 9216, 3072, 64, "split"), (13824, 3072, 12288, 64, "gate")])
def test_gemm_split_tail_round_matches_unsplit_launch_and_oracle_rows(M, N, K, K2, kind):
    """UTX_GEMM_STREAMK (default on; gemm_w4.hip "split tail"): when the 256 x 256 tiles leave the last round of workgroups less than half full,
    the tiles of that round are cut along K, fp32 partial tiles go through utx_gemm_desc.sk_work and gemm_w4_fixup_kernel sums them in K order
    and runs the epilogue.  Shapes: the single-block out-projection of BASELINE's strip (9.23 rounds -> 4 ranges per tail tile), the reference
    strip's MLP down-projection (2.48 rounds -> 2 ranges), a ragged M with GELU, the pruned last block (1.12 rounds -> 8 ranges), a
    column-split QKV projection, and a last round more than half full (2.53 rounds: 136 tiles x 3 ranges = 408 ranges in two passes).  Contract: every tile of the whole rounds has the bits of the unsplit launch; tail tiles differ by fp32
    summation order only (<= 1 bf16 ulp of the pre-residual value); two launches give the same bits; sampled rows match the oracle."""
    from unitex_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda", generator=g).to(BF)
    kw = {}
    if K2:
        kw.update(A2=(torch.randn(M, K2, device="cuda", generator=g) / 8).to(BF), B2=(torch.randn(N, K2, device="cuda", generator=g) / 4).to(BF))
    res = torch.randn(M, N, device="cuda", generator=g).to(BF)
    gate = torch.randn(N, device="cuda", generator=g).to(BF)

    def run():
        if kind == "gate":
            r = res.clone()
            ops.gemm(A, W, bias=bias, out=r, gate=gate, res=r, **kw)        # in place, as the DiT uses it
            return r
        if kind == "gelu":
            return ops.gemm(A, W, bias=bias, gelu_from=0, **kw)
        c0 = torch.empty(M, N - 3072, dtype=BF, device="cuda")
        c1 = torch.empty(M, 3072, dtype=BF, device="cuda")
        ops.gemm(A, W, bias=bias, out=c0, n_split=N - 3072, C1=c1, **kw)
        return torch.cat([c0, c1], 1)
    try:
        _lib.set_option("UTX_GEMM_STREAMK", 0)
        base = run()
        _lib.set_option("UTX_GEMM_STREAMK", 1)
        s1 = run()
        s2 = run()
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
    torch.cuda.synchronize()
    assert _same_bits(s1, s2), "split tail round is not deterministic"
    ntn, tiles = N // 256, ((M + 255) // 256) * (N // 256)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    T = tiles % ncu
    assert tiles > ncu and T > 0, "shape does not exercise the split (tiles %d, CUs %d)" % (tiles, ncu)
    diff = (s1 != base)
    assert bool(diff.any()), "the split launch has the bits of the unsplit one everywhere: the tail round was not split"
    # tiles are numbered in groups of 4 row-tiles x all column tiles, column-major inside a group (gemm_w4.hip W4_TILE_ORIGIN); the tail tiles are the
    # last T of that order.  Everything else must be untouched.
    ntm, gm = (M + 255) // 256, (8 if ntn >= 32 else 2)      # the launcher's default UTX_GEMM_GROUP_M by output width (re-swept in round 6: gemm_w4.hip)
    tail = torch.zeros(ntm, ntn, dtype=torch.bool)
    for w in range(tiles - T, tiles):
        grp, rem = divmod(w, gm * ntn)
        gs = min(gm, ntm - grp * gm)
        tn, tm = divmod(rem, gs)
        tail[grp * gm + tm, tn] = True
    dt = diff.cpu()
    dt = torch.nn.functional.pad(dt, (0, 0, 0, ntm * 256 - M)).view(ntm, 256, ntn, 256).any(3).any(1)
    assert not bool((dt & ~tail).any()), "a tile outside the last round changed"
    # rounding only: the GEMM value moves by at most one bf16 ulp (2^-7 relative; x |gate|, GELU' <= 1.13 on top -> 2^-6 of the pre-residual
    # value), and the gated residual sum is rounded once more (2^-7 of the output)
    y0 = (base.float() - res.float()) if kind == "gate" else base.float()
    y1 = (s1.float() - res.float()) if kind == "gate" else s1.float()
    bound = 2.0 ** -6 * torch.maximum(y0.abs(), y1.abs()).clamp_min(1.0)
    if kind == "gate":
        bound = bound + 2.0 ** -7 * torch.maximum(base.float().abs(), s1.float().abs())
    assert bool(((s1.float() - base.float()).abs() <= bound).all()), "split tail differs from the unsplit launch by more than rounding"
    rows = torch.tensor([0, 255, M // 2 + 17, M - 300, M - 2, M - 1])
    y = A[rows].float().cpu() @ W.float().cpu().t()
    if K2:
        y += kw["A2"][rows].float().cpu() @ kw["B2"].float().cpu().t()
    y = (y + bias.float().cpu()).to(BF).float()
    if kind == "gelu":
        y = dit_ref.gelu_tanh(y).to(BF).float()
    if kind == "gate":
        y = (res[rows].float().cpu() + (gate.float().cpu() * y).to(BF).float()).to(BF).float()
    rel = ((s1[rows].float().cpu() - y).abs() / y.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "split tail vs oracle rows: %g" % rel


@pytest.mark.parametrize("M,N,K,K2,S,kind", [(13376, 3072, 512, 512, 2, "gate"), (13376, 3072, 512, 512, 3, "gate"), (13001, 3072, 192, 64, 4, "plain"),
                                             (13376, 3072, 320, 128, 4, "gate"), (13824, 3072, 576, 0, 3, "plain")])
def test_gemm_split_tail_forced_range_shapes(M, N, K, K2, S, kind):
    """UTX_GEMM_STREAMK = 1000 + S forces S ranges per tail tile on shapes the launcher's cost model never splits -- the range geometries of
    gemm_w4.hip's staging cursor: a range that IS the LoRA segment (K = K2 = 512, S = 2), ranges that straddle the base -> LoRA switch and start
    inside the LoRA segment at an odd K-tile (S = 3), ranges of ONE K-tile (K = 192 + 64, S = 4), four ranges of unequal length over 7 K-tiles in two passes of the grid, and
    136 tail tiles x 3 ranges in two passes of the grid (M = 13 824).  Same contract as the test below."""
    from unitex_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K + S)
    A = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda", generator=g).to(BF)
    kw = {}
    if K2:
        kw.update(A2=(torch.randn(M, K2, device="cuda", generator=g) / 8).to(BF), B2=(torch.randn(N, K2, device="cuda", generator=g) / 4).to(BF))
    res = torch.randn(M, N, device="cuda", generator=g).to(BF)
    gate = torch.randn(N, device="cuda", generator=g).to(BF)

    def run():
        if kind == "gate":
            r = res.clone()
            ops.gemm(A, W, bias=bias, out=r, gate=gate, res=r, **kw)
            return r
        return ops.gemm(A, W, bias=bias, **kw)
    try:
        _lib.set_option("UTX_GEMM_STREAMK", 0)
        base = run()
        _lib.set_option("UTX_GEMM_STREAMK", 1000 + S)
        s1 = run()
        s2 = run()
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
    torch.cuda.synchronize()
    assert _same_bits(s1, s2), "forced split is not deterministic"
    assert bool((s1 != base).any()), "the forced split has the bits of the unsplit launch everywhere: nothing was split"
    y0 = (base.float() - res.float()) if kind == "gate" else base.float()
    y1 = (s1.float() - res.float()) if kind == "gate" else s1.float()
    bound = 2.0 ** -6 * torch.maximum(y0.abs(), y1.abs()).clamp_min(1.0)
    if kind == "gate":
        bound = bound + 2.0 ** -7 * torch.maximum(base.float().abs(), s1.float().abs())
    assert bool(((s1.float() - base.float()).abs() <= bound).all()), "forced split differs from the unsplit launch by more than rounding"
    frac = float((s1 != base).float().mean())
    assert frac < 0.02, "a forced split changes rounding in a few elements of the tail tiles only, not %.3f of the output" % frac


@pytest.mark.parametrize("M", [S_FULL, 13824, 6336])   # 6336 = 50688 / 8: ragged last 256-row tile (sequence-parallel shard)
def test_gemm_full_size_kernels_agree_bitwise_and_match_oracle_rows(M):
    ops = _ops()
    D, R = 3072, 64
    g = torch.Generator(device="cuda").manual_seed(M)
    x = (torch.randn(M, D, device="cuda", generator=g) / 2).to(BF)
    # (1) fused single-stream projection: [q|k|v|mlp] with LoRA K-segment on q,k,v, GELU on mlp, column split
    N = 3 * D + 4 * D
    W = (torch.randn(N, D, device="cuda", generator=g) / math.sqrt(D)).to(BF)
    bias = torch.randn(N, device="cuda", generator=g).to(BF)
    T = (torch.randn(M, 3 * R, device="cuda", generator=g) / 8).to(BF)
    Bl = torch.zeros(N, R, dtype=BF, device="cuda")
    Bl[:3 * D] = (torch.randn(3 * D, R, device="cuda", generator=g) / 4).to(BF)

    def fused():
        c0 = torch.empty(M, 3 * D, dtype=BF, device="cuda")
        c1 = torch.empty(M, 4 * D, dtype=BF, device="cuda")
        ops.gemm(x, W, bias=bias, out=c0, A2=T, B2=Bl, lora_n_limit=3 * D, lora_seg_n=D, gelu_from=3 * D, n_split=3 * D, C1=c1)
        return torch.cat([c0, c1], 1)
    o = _gemm_variants(fused)
    assert all(_same_bits(o[k], o["128"]) for k in ("2562", "256", "2560", "2564")), "fused qkv|mlp: kernels disagree"
    rows = torch.tensor([0, 255, 256, 1000, M // 2 + 17, M - 1])
    # oracle arithmetic (fp32 on the CPU, bf16 rounding at the same tensor boundaries) on the sampled rows
    y = x[rows].float().cpu() @ W.float().cpu().t()
    for si in range(3):
        y[:, si * D:(si + 1) * D] += T[rows].float().cpu()[:, si * R:(si + 1) * R] @ Bl.float().cpu()[si * D:(si + 1) * D].t()
    y = (y + bias.float().cpu()).to(BF).float()
    y[:, 3 * D:] = dit_ref.gelu_tanh(y[:, 3 * D:]).to(BF).float()
    rel = ((o["256"][rows].float().cpu() - y).abs() / y.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "fused qkv|mlp vs oracle rows: %g" % rel
    del o
    # (2) out-projection with the K = 15360 concat input and the gated residual epilogue
    cat = (torch.randn(M, 5 * D, device="cuda", generator=g) / 4).to(BF)
    Wo = (torch.randn(D, 5 * D, device="cuda", generator=g) / math.sqrt(5 * D)).to(BF)
    bo = torch.randn(D, device="cuda", generator=g).to(BF)
    gate = torch.randn(D, device="cuda", generator=g).to(BF)
    res = torch.randn(M, D, device="cuda", generator=g).to(BF)

    def gated():
        r = res.clone()
        ops.gemm(cat, Wo, bias=bo, out=r, gate=gate, res=r)
        return r
    o = _gemm_variants(gated)
    assert all(_same_bits(o[k], o["128"]) for k in ("2562", "256", "2560", "2564")), "gated residual: kernels disagree"
    yy = ((cat[rows].float().cpu() @ Wo.float().cpu().t()) + bo.float().cpu()).to(BF).float()
    yy = (res[rows].float().cpu() + (gate.float().cpu() * yy).to(BF).float()).to(BF).float()
    rel = ((o["256"][rows].float().cpu() - yy).abs() / yy.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "gated residual vs oracle rows: %g" % rel


@pytest.mark.parametrize("M,N,K,K2", [(3500, 3584, 192, 64), (3584, 3584, 64, 0), (7000, 2048, 320, 128), (4096, 12544, 128, 64)])
def test_gemm_persistent_kernel_edge_shapes_bit_identical_to_tiled_kernels(M, N, K, K2):
    """the persistent continuous-stream kernels (gemm_w4.hip: default for >= 192 tiles of 256 x 256; gemm_pers.hip) at the shapes that stress their tile
    boundary logic: ragged M (rows >= M in the last tile row, including waves with no valid row), an ODD number of K-tiles per
    tile (K = 192, and 48 + 1 style LoRA segments: the stream re-enters at odd LDS parity), a single K-tile per tile (K = 64:
    every K-tile is first and last), tiles with and without the LoRA segment in one launch, GELU / column split on a tile
    boundary, and the gated residual updated IN PLACE (res aliases C).  All five kernels (incl. the one-wave-per-SIMD kernel of
    gemm_w4.hip, whose stream unit is a 64-k K-tile: K = 64 is ONE K-tile per tile -- every K-tile first and last --, and its ragged-M gated slow
    path) must agree bit for bit."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) / 2).to(BF)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda", generator=g).to(BF)
    n_lim = (N // 512) * 256           # LoRA on the first half of the column tiles only
    kw = {}
    if K2:
        T = (torch.randn(M, K2, device="cuda", generator=g) / 8).to(BF)
        Bl = torch.zeros(N, K2, dtype=BF, device="cuda")
        Bl[:n_lim] = (torch.randn(n_lim, K2, device="cuda", generator=g) / 4).to(BF)
        kw = dict(A2=T, B2=Bl, lora_n_limit=n_lim, lora_seg_n=n_lim)
    split = (N // 768) * 256

    def plain():
        c0 = torch.full((M, split), 7.0, dtype=BF, device="cuda")
        c1 = torch.full((M, N - split), 7.0, dtype=BF, device="cuda")
        ops.gemm(A, W, bias=bias, out=c0, gelu_from=split, n_split=split, C1=c1, alpha=0.75, **kw)
        return torch.cat([c0, c1], 1)
    o = _gemm_variants(plain)
    assert all(_same_bits(o[k], o["128"]) for k in ("2562", "256", "2560", "2564")), "plain / GELU / split epilogue: kernels disagree"
    ref = (0.75 * (A.float() @ W.float().t() + ((kw["A2"].float() @ kw["B2"].float().t()) if K2 else 0.0)) + bias.float()).to(BF).float()
    ref[:, split:] = dit_ref.gelu_tanh(ref[:, split:].cpu()).to(BF).float().cuda()
    rel = ((o["2560"].float() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel < 1.6e-2, "persistent kernel vs fp32 reference: %g" % rel
    gate = torch.randn(N, device="cuda", generator=g).to(BF)
    res = torch.randn(M, N, device="cuda", generator=g).to(BF)

    def gated():
        r = res.clone()
        ops.gemm(A, W, bias=bias, out=r, gate=gate, res=r, **kw)
        return r
    o = _gemm_variants(gated)
    assert all(_same_bits(o[k], o["128"]) for k in ("2562", "256", "2560", "2564")), "gated residual (in place): kernels disagree"
    # repeated launches: the stream has no state that survives a launch
    again = gated()
    assert _same_bits(again, o["2560"])
    # the alternative DMA placement of the persistent kernel (UTX_GEMM_PERS_SCHED) computes the same bits
    from unitex_amd import _lib
    try:
        _set_tile("2560")
        for forced in (1, 2):       # both DMA placements of the persistent kernel, whatever the shape heuristic picked above
            _lib.set_option("UTX_GEMM_PERS_SCHED", forced)
            assert _same_bits(gated(), o["2560"])
    finally:
        _lib.set_option("UTX_GEMM_PERS_SCHED", 0)
        _set_tile(None)


def test_full_width_dit_blocks_at_config1_shape_match_oracle():
    """BASELINE.json configs[0] shape (512^2 x 4 views: 4096 noise + 4096 control + 1024 dual + 512 text = 9728 tokens) at
    the real FLUX width (D = 3072, 24 heads, joint_dim 4096, rank-64 LoRA), depth cut to 1 double + 1 single block so
    the fp32 CPU oracle finishes in seconds.  This is the shape at which every large-M kernel choice (8-phase GEMM,
    fused LoRA K-segment, split / GELU / gated epilogues, 24-head attention) is the production one."""
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.FluxConfig(num_double=1, num_single=1)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    shape = FluxShape(num_double=1, num_single=1)
    S_txt = 512
    ids = [dit_ref.latent_image_ids(32, 128), dit_ref.latent_image_ids(32, 128, offset_y=32),
           dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=32)]
    img_ids = torch.cat(ids, 0)
    S_img = img_ids.shape[0]
    assert S_txt + S_img == 9728
    g = torch.Generator().manual_seed(63)
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = torch.zeros(S_txt, cfg.joint_dim).to(BF)              # the reference feeds zero prompt embeddings
    pooled = torch.zeros(1, cfg.pooled_dim).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=2)
    lb = dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=3)
    loras = [(la, 1.0), (lb, 0.0)]
    m = FluxDiT(sd, shape, device="cuda:0")
    m.fuse_qk = True            # opt-in (UTX_FUSE_QK=1): q / k post-processing inside the QKV projections' epilogue
    from unitex_amd import _lib
    _lib.set_option("UTX_GEMM_STREAMK", 0)      # the bit-comparison below spans a fused (never split) and an unfused (tail round split) QKV projection
    m.set_lora(loras)
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
    out = m.forward(lat.cuda(), 0.4375).float().cpu()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ref = dit_ref.flux_forward(sd, cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids,
                               loras=loras, emulate_bf16=True)
    err = (out - ref).abs().max().item()
    mx = ref.abs().max().item()
    assert torch.isfinite(out).all()
    assert err < 0.03 * max(mx, 1.0), "full-width DiT blocks: err %g (ref max %g)" % (err, mx)
    assert (out - ref).abs().mean().item() < 0.004 * max(mx, 1.0)
    # the forward above ran with q / k post-processing fused into the QKV projections (these shapes take the one-wave-per-SIMD GEMM);
    # GEMM -> utx_qkv_post must give the same bits
    fused_gemms = sum(1 for fn, d in _flat_plan(m) if fn is m.lib.utx_gemm_bf16 and d.qk_cols > 0)
    assert fused_gemms == 2, "expected the image-stream QKV projection of the double block and the single block's projection to be fused, got %d" % fused_gemms
    def rerun(fused):
        m.fuse_qk = fused
        m._drop_plans()
        m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
        assert (sum(1 for fn, d in _flat_plan(m) if fn is m.lib.utx_gemm_bf16 and d.qk_cols > 0) == 2) == fused
        o_ = m.forward(lat.cuda(), 0.4375).float().cpu()
        torch.cuda.synchronize()
        return o_
    try:
        # STRICT (round 4; restored for real in round 6 -- a botched splice had left the lenient round-3 body as the definition pytest ran): the one-off one-ulp difference
        # of round 3 is explained and fixed -- the compiler had placed copies of the cos / sin registers in front of the ragged-tile branch's `s_waitcnt vmcnt(0)` in the
        # fused epilogue (the single block's M = 9280 has a ragged last row of tiles); the wait macros no longer allow that and tests/test_asm_hazards_cpu.py audits
        # every build's listing.  A first mismatch fails: no re-run, no warning.
        out2 = rerun(False)
        assert torch.equal(out2, out), "fused q / k epilogue changed the forward: max |d| %g in %d elements" % ((out2 - out).abs().max().item(), int((out2 != out).sum()))
        print("STRICT fused-vs-unfused comparison executed: %d elements bit-identical" % out.numel())
    finally:
        _lib.set_option("UTX_GEMM_STREAMK", 1)
    # and with the split tail round of the large GEMMs (the default): same forward up to fp32 summation order in the tail tiles
    m.fuse_qk = False
    m._drop_plans()
    m.set_conditioning(enc.cuda(), pooled.cuda(), 3.5)
    out3 = m.forward(lat.cuda(), 0.4375).float().cpu()
    assert (out3 - ref).abs().max().item() < 0.03 * max(mx, 1.0) and (out3 - out).abs().max().item() < 0.02 * max(mx, 1.0)


def _flat_plan(m):
    def walk(ops_):
        for fn, d in ops_:
            if fn == "par":
                yield from walk(d[0]); yield from walk(d[1])
            else:
                yield fn, d
    return list(walk(next(iter(m._plans.values()))["plan"]))


def test_address_arithmetic_beyond_32_bit_offsets():
    """Maximum sizes: the joint-sequence reading of BASELINE configs[4] (2048^2 x 8 views in ONE strip) is 263 232 executed tokens -- the single-block `cat`
    buffer [S, 15 360] holds 4.04e9 elements (8.1 GB), `qkv` [S, 9216] 2.4e9: element AND byte offsets leave 32 bits.  Every kernel on the step path is run at
    that token count and its rows at the far end are checked (an index that wraps reads / writes the wrong rows): the large-M GEMM (GELU columns into the
    strided cat view), LayerNorm-modulation, q / k / v post-processing (24 heads, the V^T transpose across a 263 232-column row), attention (one head, all keys)."""
    ops = _ops()
    S, D = 263232, 3072
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    far = torch.arange(S - 200, S, device=dev)
    near = torch.arange(0, 200, device=dev)
    rows = torch.cat([near, torch.tensor([70000, 139811, 139812, 209715, 209716], device=dev), far])     # around 2^31 / 15360 and 2^32 / (2 * 15360) rows as well
    # ---- GEMM: cat[:, D:] = GELU(x W^T + b), M = S, N = 12 288, K = 3072, ldc = 15 360
    x = (torch.randn(S, D, device=dev, generator=g) * 0.5).to(BF)
    W = (torch.randn(4 * D, D, device=dev, generator=g) / math.sqrt(D)).to(BF)
    b = torch.randn(4 * D, device=dev, generator=g).to(BF)
    cat = torch.zeros(S, 5 * D, dtype=BF, device=dev)
    ops.gemm(x, W, bias=b, out=cat[:, D:], gelu_from=0)
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(x[rows].float() @ W.float().t() + b.float(), approximate="tanh")
    got = cat[rows][:, D:].float()
    err = (got - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), "GEMM rows at the far end: %g" % err
    assert cat[rows][:, :D].abs().max().item() == 0.0, "the GEMM wrote outside its column window"
    del W, b
    # ---- LayerNorm-modulation over S rows
    shift = torch.randn(D, device=dev, generator=g).to(BF); scale = (0.1 * torch.randn(D, device=dev, generator=g)).to(BF)
    y = ops.ln_mod(x, shift, scale)
    torch.cuda.synchronize()
    y_rows = ops.ln_mod(x[rows].contiguous(), shift, scale)      # a per-row operation: the same rows processed on their own must give the same bits
    torch.cuda.synchronize()
    assert torch.equal(y[rows], y_rows), "ln_mod rows at the far end"
    ref = torch.nn.functional.layer_norm(x[rows].float(), (D,), eps=1e-6) * (1.0 + scale.float()) + shift.float()
    assert (y_rows.float() - ref).abs().max().item() < 0.08      # (the kernel rounds to bf16 where the reference's bf16 modules do: a few ulps of values up to ~6)
    del y, cat
    # ---- q / k / v post-processing: 24 heads, S tokens (rows of the far end must equal the same rows processed on their own)
    H = H_FULL
    qkv = (torch.randn(S, 3 * D, device=dev, generator=g) * 0.5).to(BF)
    wq = (1.0 + 0.1 * torch.randn(128, device=dev, generator=g)).to(BF); wk = (1.0 + 0.1 * torch.randn(128, device=dev, generator=g)).to(BF)
    ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    Qh = torch.zeros(H, S, 128, dtype=BF, device=dev); Kh = torch.zeros(H, S, 128, dtype=BF, device=dev); Vt = torch.zeros(H, 128, S, dtype=BF, device=dev)
    ops.qkv_post(qkv, 0, D, 2 * D, wq, wk, cos, sin, Qh, Kh, Vt, S, 0, H, q_scale=0.1275)
    torch.cuda.synchronize()
    n = 256
    for r0 in (0, S - n):
        q2 = torch.zeros(H, n, 128, dtype=BF, device=dev); k2 = torch.zeros(H, n, 128, dtype=BF, device=dev); v2 = torch.zeros(H, 128, n, dtype=BF, device=dev)
        ops.qkv_post(qkv[r0:r0 + n].contiguous(), 0, D, 2 * D, wq, wk, cos[r0:r0 + n].contiguous(), sin[r0:r0 + n].contiguous(), q2, k2, v2, n, 0, H, q_scale=0.1275)
        torch.cuda.synchronize()
        assert torch.equal(Qh[:, r0:r0 + n], q2) and torch.equal(Kh[:, r0:r0 + n], k2) and torch.equal(Vt[:, :, r0:r0 + n], v2), "qkv_post rows at %d" % r0
    assert torch.equal(Vt[H - 1, 127, S - 8:].float(), qkv[S - 8:, 2 * D + (H - 1) * 128 + 127].float()), "last V^T row, last columns"
    del qkv, cos, sin, ang
    # ---- attention: one head over all 263 232 keys, first / last query rows against the fp32 softmax
    q1, k1, vt1 = Qh[:1].contiguous(), Kh[:1].contiguous(), Vt[:1].contiguous()
    out = ops.attention(q1, k1, vt1, S=S, scale=0.0)       # Q carries scale * log2(e) (q_scale above): scores are base-2 exponents
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    pick = torch.cat([torch.arange(0, 64, device=dev), torch.arange(S - 64, S, device=dev)])
    sc = (q1[0, pick].float() @ k1[0].float().t()) * math.log(2.0)
    p = torch.softmax(sc, dim=-1)
    ref = p @ vt1[0].float().t()
    err = (out[pick].float() - ref).abs().max().item()
    assert err < 3e-2, "attention over 263 232 keys: %g" % err


def test_address_arithmetic_beyond_32_bit_offsets_mx_fp8():
    """The same at 263 232 tokens for the MX fp8 path: the activation scratch [S, 15 360] bytes is 4.04e9 B (beyond 2^31, just below 2^32), its tile-packed
    scale buffer has 2057 row blocks per K-tile slab.  The quantiser (packed) over all rows into the strided scratch, then the one-wave-per-SIMD MX GEMM over
    K = 15 360 (the single blocks' out-projection shape); the last 128-aligned row block must equal, bit for bit, the same rows quantised and multiplied on their own."""
    ops = _ops()
    from unitex_amd.flux import mx8
    ctx = ops.get_ctx(0)
    S, D = 263232, 3072
    K = 5 * D
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.empty(S, K, dtype=BF, device=dev)
    for c0 in range(0, S, 32768):      # filled in slabs: randn of 4e9 elements at once would need 16 GB of fp32
        c1 = min(S, c0 + 32768)
        x[c0:c1] = (torch.randn(c1 - c0, K, device=dev, generator=g) * 0.5).to(BF)
    W = (torch.randn(D, K, device=dev, generator=g) / math.sqrt(K)).to(BF)
    b = torch.randn(D, device=dev, generator=g).to(BF)
    wq, wp = mx8.quantize_weight(W, ctx, packed=True)
    aq = torch.empty(S, K, dtype=torch.uint8, device=dev)
    ap = mx8.PackedScales(mx8.packed_scale_buffer(S, K, dev), S, K)
    mx8.quantize_act(x, ctx, out=(aq, ap), packed=True)
    y = torch.empty(S, D, dtype=BF, device=dev)
    ops.gemm(aq, wq, bias=b, out=y, a_scale=ap, b_scale=wp)
    torch.cuda.synchronize()
    r0 = (S // 128 - 2) * 128          # the last two whole row blocks + the ragged tail
    n = S - r0
    aq2, ap2 = mx8.quantize_act(x[r0:].contiguous(), ctx, packed=True)
    torch.cuda.synchronize()
    assert torch.equal(aq[r0:], aq2), "quantised bytes of the far-end rows"
    assert torch.equal(ap.rowmajor()[r0:], ap2.rowmajor()), "E8M0 scales of the far-end rows"
    y2 = torch.empty(n, D, dtype=BF, device=dev)
    ops.gemm(aq2, wq, bias=b, out=y2, a_scale=ap2, b_scale=wp)
    torch.cuda.synchronize()
    # (these rows lie in the launch's last, partly filled round, which is cut along K and summed by the fix-up kernel: another fp32 summation order than the
    # unsplit launch of the slice -- equal up to one bf16 ulp, not bit for bit)
    dmax = (y[r0:].float() - y2.float()).abs().max().item()
    assert dmax <= 2.0 ** -6 * max(1.0, y2.float().abs().max().item()), "MX GEMM rows at the far end: %g" % dmax
    ref = x[r0:].float() @ W.float().t() + b.float()
    rel = ((y2.float() - ref).norm() / ref.norm()).item()
    assert rel < 0.06, "fp8 product vs the bf16 operands: relative Frobenius %g" % rel
