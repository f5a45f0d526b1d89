"""Functional wrappers over the C ABI for the DiT ops (tensor in / tensor out).

These are what the parity tests call; FluxDiT (transformer.py) uses pre-built descriptors instead.
All tensors are torch CUDA tensors (bf16 unless stated); nothing here computes on the host.
"""
import ctypes as C
import math

import torch

from .. import _lib
from .._lib import GemmDesc, GemvDesc, LnModDesc, QkvPostDesc, SchedDesc, ptr

_CTX = {}


def get_ctx(device=None):
    if device is None:
        device = torch.cuda.current_device()
    if device not in _CTX:
        _CTX[device] = _lib.Context(device)
    return _CTX[device]


def _bf(t):
    assert t.is_cuda and t.dtype == torch.bfloat16, "expected a CUDA bf16 tensor"
    return t


def attention(q, k, vt, S=None, scale=None, out=None, o_ss=None, key_bias_log2=0.0, key_bias_period=0, S_q=None):
    """q,k [H,S_pad,128]; vt [H,128,S_pad] (S_pad multiple of 64, zero padded) -> o [S, H*128].
    key_bias_log2 / key_bias_period: key multiplicity of tile 0 (and every period-th tile), see utx_attn_fwd_bf16_kb.
    S_q != S: rows 0..S_q-1 of q are the queries (o gets S_q rows) over the S keys of k / vt -- fewer (last-block pruning) or, since round 6, more than the keys (a
    sequence-parallel rank's launch over de-duplicated text keys: q [H, >= S_q, 128], k [H, >= S, 128] are separate arrays), see utx_attn_fwd_bf16_kbq."""
    ctx = get_ctx(q.device.index)
    H, S_pad, D = q.shape
    assert D == 128 and vt.shape[1] == 128 and vt.shape[2] % 64 == 0
    S = S_pad if S is None else S
    if scale is None:
        scale = 1.0 / math.sqrt(D)   # pass scale=0.0 when Q was pre-scaled by scale*log2(e) in qkv_post
    S_q = S if S_q is None else S_q
    if out is None:
        out = torch.empty(S_q, H * D, dtype=torch.bfloat16, device=q.device)
    if o_ss is None:
        o_ss = out.stride(0)
    rc = ctx.lib.utx_attn_fwd_bf16_kbq(ctx.handle, ptr(_bf(q)), ptr(_bf(k)), ptr(_bf(vt)), ptr(out),
                                       q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                       vt.stride(0), vt.stride(1), o_ss, H, S_q, S, float(scale), float(key_bias_log2), int(key_bias_period),
                                       ctx.stream())
    ctx.check(rc)
    return out


def quant_qk_mx8(xh, out=None):
    """head-major Q or K [H, S_pad, 128] bf16 -> (e4m3 bytes [H, S_pad, 128], scales [H, S_pad, 4] uint8: the four E8M0 bytes of a row = one dword) for
    utx_attn_fwd_fp8: utx_quant_mx8 over the rows of the [H * S_pad, 128] view (blocks of 32 along d)."""
    ctx = get_ctx(xh.device.index)
    H, S_pad, D = xh.shape
    assert D == 128 and xh.is_contiguous()
    q8, sc = out if out is not None else (torch.empty(H, S_pad, 128, dtype=torch.uint8, device=xh.device), torch.empty(H, S_pad, 4, dtype=torch.uint8, device=xh.device))
    ctx.check(ctx.lib.utx_quant_mx8(ctx.handle, ptr(xh), 128, ptr(q8), 128, ptr(sc), 4, H * S_pad, 128, ctx.stream()))
    return q8, sc


def quant_vt_mx8(vt, out=None):
    """V^T [H, 128, S_pad] bf16 -> (e4m3 bytes [H, 128, S_pad], scales [H, S_pad / 32, 32, 4] uint8: E8M0 per (block of 32 keys, channel d % 32, d / 32))"""
    ctx = get_ctx(vt.device.index)
    H, D, S_pad = vt.shape
    assert D == 128 and S_pad % 64 == 0 and vt.is_contiguous()
    v8, vs = out if out is not None else (torch.empty(H, 128, S_pad, dtype=torch.uint8, device=vt.device), torch.empty(H, S_pad // 32, 32, 4, dtype=torch.uint8, device=vt.device))
    ctx.check(ctx.lib.utx_quant_vt_mx8(ctx.handle, ptr(vt), ptr(v8), ptr(vs), H, S_pad, ctx.stream()))
    return v8, vs


def attention_fp8(q8, qs, k8, ks, v8t, vs, S=None, S_q=None, out=None, o_ss=None, key_bias_log2=0.0, key_bias_period=0):
    """MX fp8 attention (utx_attn_fwd_fp8; opt-in): operands from quant_qk_mx8 / quant_vt_mx8; Q must have been pre-scaled by scale * log2(e).  -> o [S_q, H * 128] bf16"""
    ctx = get_ctx(q8.device.index)
    H, S_pad, D = q8.shape
    assert D == 128 and v8t.shape == (H, 128, S_pad)
    S = S_pad if S is None else S
    S_q = S if S_q is None else S_q
    if out is None:
        out = torch.empty(S_q, H * 128, dtype=torch.bfloat16, device=q8.device)
    if o_ss is None:
        o_ss = out.stride(0)
    ctx.check(ctx.lib.utx_attn_fwd_fp8(ctx.handle, ptr(q8), ptr(qs), ptr(k8), ptr(ks), ptr(v8t), ptr(vs), ptr(out), o_ss, H, S_q, S, S_pad,
                                       float(key_bias_log2), int(key_bias_period), ctx.stream()))
    return out


def make_gemm_desc(A, B, C_out, bias=None, A2=None, B2=None, lora_n_limit=None, lora_seg_n=None, alpha=1.0,
                   gelu_from=None, gate=None, res=None, n_split=None, C1=None, a_scale=None, b_scale=None, qk_post=None, sk_work=None, q_out=None):
    """a_scale / b_scale given: A and B are OCP MX fp8 operands (uint8 e4m3 bytes + E8M0 scales [rows, K/32], flux/mx8.py).
    sk_work: uint8 scratch tensor of streamk_workspace() bytes for the balanced tail of the large-M kernel (utx_gemm_desc.sk_work); one per
    stream that launches GEMMs concurrently.  None = the tail round is never split."""
    M, K = A.shape
    N = B.shape[0]
    d = GemmDesc()
    d.A, d.lda, d.B, d.ldb = ptr(A), A.stride(0), ptr(B), B.stride(0)
    if a_scale is not None:
        assert A.dtype == torch.uint8 and B.dtype == torch.uint8 and b_scale is not None
        if hasattr(a_scale, "row_blocks"):     # mx8.PackedScales on both operands: the one-wave-per-SIMD MX kernel (mx8 = 2)
            assert hasattr(b_scale, "row_blocks"), "packed activation scales need packed weight scales"
            d.a_scale, d.lds_a, d.b_scale, d.lds_b, d.mx8 = ptr(a_scale.data), a_scale.row_blocks, ptr(b_scale.data), b_scale.row_blocks, 2
        else:
            d.a_scale, d.lds_a, d.b_scale, d.lds_b, d.mx8 = ptr(a_scale), a_scale.stride(0), ptr(b_scale), b_scale.stride(0), 1
    if A2 is not None:
        d.A2, d.lda2, d.B2, d.ldb2 = ptr(A2), A2.stride(0), ptr(B2), B2.stride(0)
        d.K2 = B2.shape[1]
        d.lora_n_limit = N if lora_n_limit is None else lora_n_limit
        d.lora_seg_n = N if lora_seg_n is None else lora_seg_n
    else:
        d.K2, d.lora_n_limit, d.lora_seg_n = 0, 0, 128
    d.M, d.N, d.K = M, N, K
    if qk_post is not None:
        # fused q / k post-processing (utx_gemm_desc.qk_cols): dict(cols, tok_off, eps, q_scale, wq, wk, cos, sin, Qh, Kh)
        q = qk_post
        d.qk_cols, d.qk_tok_off, d.qk_eps, d.qk_q_scale = int(q["cols"]), int(q["tok_off"]), float(q["eps"]), float(q["q_scale"])
        d.qk_wq, d.qk_wk, d.qk_cos, d.qk_sin = ptr(q["wq"]), ptr(q["wk"]), ptr(q["cos"]), ptr(q["sin"])
        d.qk_Qh, d.qk_Kh, d.qk_hs = ptr(q["Qh"]), ptr(q["Kh"]), q["Qh"].stride(0)
    d.alpha = alpha
    d.bias = ptr(bias)
    d.gelu_from = N if gelu_from is None else gelu_from
    d.gate = ptr(gate)
    if gate is not None:
        d.res, d.ldres = ptr(res), res.stride(0)
    d.C, d.ldc = ptr(C_out), C_out.stride(0)
    d.n_split = N if n_split is None else n_split
    if C1 is not None:
        d.C1, d.ldc1 = ptr(C1), C1.stride(0)
    if sk_work is not None:
        d.sk_work, d.sk_work_bytes = ptr(sk_work), sk_work.numel() * sk_work.element_size()
    if q_out is not None:
        # (uint8 view [M, N - gelu_from] of the consumer's fp8 activation scratch, its packed scale buffer [K/128, row blocks, 512], first K-tile): the GELU
        # columns leave as MX fp8 (utx_gemm_desc.q_out; mx8 = 2 only)
        qv, qs, kt0 = q_out
        d.q_out, d.ldq_out, d.qs_out, d.qs_out_rb, d.q_out_kt0 = ptr(qv), qv.stride(0), ptr(qs), qs.stride(0) // 512, int(kt0)
    return d


_SK_WS = {}


def streamk_workspace(device, stream=None, shared=True):
    """scratch for utx_gemm_desc.sk_work on `device`.  shared=True: one cached buffer per (device, stream) -- GEMMs launched on the same
    stream are ordered, so they can share it; shared=False: a fresh buffer (a model that owns its plan)."""
    dev = torch.device(device)
    ctx = get_ctx(dev.index)
    nbytes = int(ctx.lib.utx_gemm_streamk_workspace_bytes(ctx.handle))
    if not shared:
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev) if stream is None else stream
    key = (dev.index, st.cuda_stream)
    if key not in _SK_WS:
        _SK_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return _SK_WS[key]


GEMM_KERNELS = ("128x128", "w4", "pers8", "8phase", "2barrier")


def gemm_plan(M, N, K=3072, K2=0, n_split=None, gelu_from=None, lora_seg_n=None, lora_n_limit=None, sk=True, n_cus=256):
    """what utx_gemm_bf16 does with this shape under the current launch options (utx_gemm_plan: the library's own dispatch arithmetic, no device
    needed): dict(kernel=one of GEMM_KERNELS, tiles, tail_tiles, ranges) -- tail_tiles > 0 when the one-wave-per-SIMD kernel cuts its last round
    along K into `ranges` ranges per tile (needs scratch: sk)."""
    lib = _lib.load_library()
    d = GemmDesc()
    d.A = d.B = d.C = 1                       # never dereferenced by the plan
    d.M, d.N, d.K, d.K2 = int(M), int(N), int(K), int(K2)
    d.lora_n_limit = (N if lora_n_limit is None else lora_n_limit) if K2 else 0
    d.lora_seg_n = (N if lora_seg_n is None else lora_seg_n) if K2 else 128
    d.n_split = N if n_split is None else n_split
    d.gelu_from = N if gelu_from is None else gelu_from
    d.sk_work = 1 if sk else None
    out = (C.c_int * 4)()
    rc = lib.utx_gemm_plan(C.byref(d), int(n_cus), C.byref(out))
    if rc != 0:
        raise ValueError("utx_gemm_plan -> %d" % rc)
    return {"kernel": GEMM_KERNELS[out[0]], "tiles": out[1], "tail_tiles": out[2], "ranges": out[3]}


def gemm_takes_w4(M, N, n_split=None, gelu_from=None, K2=0, lora_seg_n=None, lora_n_limit=None):
    """True when utx_gemm_bf16 dispatches this shape to the one-wave-per-SIMD 256 x 256 kernel (gemm_w4.hip) -- the kernel that carries the
    fused q / k epilogue.  Asked of the library (utx_gemm_plan), not restated here."""
    return gemm_plan(M, N, K2=K2, n_split=n_split, gelu_from=gelu_from, lora_seg_n=lora_seg_n, lora_n_limit=lora_n_limit)["kernel"] == "w4"


def mx8_uses_packed(M, N, n_split=None, gelu_from=None):
    """MX fp8 GEMM of this shape on the one-wave-per-SIMD kernel (tile-packed scales, utx_gemm_desc.mx8 = 2)?  The kernel's own constraints (N and
    every column boundary on 256) and the bf16 dispatch's fill rule (>= 192 tiles of 256 x 256); smaller shapes keep the 128 x 128-tile MX kernel."""
    if N % 256 or (n_split is not None and n_split < N and n_split % 256) or (gelu_from is not None and gelu_from < N and gelu_from % 256):
        return False
    return ((M + 255) // 256) * (N // 256) >= 192


def gemm(A, B, bias=None, out=None, **kw):
    """out = epi(alpha*(A B^T + A2 B2^T) + bias); see make_gemm_desc / gemm.hip for the epilogues."""
    ctx = get_ctx(A.device.index)
    M, N = A.shape[0], B.shape[0]
    n_split = kw.get("n_split")
    if out is None:
        out = torch.empty(M, N if n_split is None else n_split, dtype=torch.bfloat16, device=A.device)
    if kw.get("a_scale") is None:
        _bf(A), _bf(B)
    if "sk_work" not in kw and M >= 4096:
        kw = dict(kw, sk_work=streamk_workspace(A.device))
    d = make_gemm_desc(A, B, out, bias=bias, **kw)
    ctx.check(ctx.lib.utx_gemm_bf16(ctx.handle, C.byref(d), ctx.stream()))
    return out


def gemv(x, W, bias=None, silu_in=False, silu_out=False, out=None):
    ctx = get_ctx(x.device.index)
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    d = GemvDesc()
    d.x, d.ldx, d.W, d.ldw, d.bias = ptr(_bf(x)), x.stride(0), ptr(_bf(W)), W.stride(0), ptr(bias)
    d.y, d.ldy, d.M, d.N, d.K = ptr(out), out.stride(0), M, N, K
    d.silu_in, d.silu_out = int(silu_in), int(silu_out)
    ctx.check(ctx.lib.utx_gemv_bf16(ctx.handle, C.byref(d), ctx.stream()))
    return out


def qkv_post(qkv, q_col, k_col, v_col, wq, wk, cos, sin, Qh, Kh, Vt, n_tok, tok_off, H, eps=1e-6, q_scale=1.0,
             heads_per_group=0, group_stride=0, head_stride=None, row_stride_v=None, skip_qk=False, sub_heads=0, sub_stride=0):
    """heads_per_group > 0: Qh / Kh / Vt are flat bases of a grouped layout (the sequence-parallel send buffer, ulysses.py):
    head h at (h // g) * group_stride + (h % g) * head_stride, V^T rows row_stride_v apart; with sub_heads = s > 0 the heads of a
    group are cut once more: (h // g) * group_stride + ((h % g) // s) * sub_stride + (h % s) * head_stride."""
    ctx = get_ctx(qkv.device.index)
    d = QkvPostDesc()
    d.qkv, d.ld = ptr(_bf(qkv)), qkv.stride(0)
    d.q_col, d.k_col, d.v_col = q_col, k_col, v_col
    d.wq, d.wk = ptr(_bf(wq)), ptr(_bf(wk))
    assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous()
    d.cosb, d.sinb = ptr(cos), ptr(sin)
    d.Qh, d.Kh, d.Vt = ptr(Qh), ptr(Kh), ptr(Vt)
    if heads_per_group:
        d.hs_qk, d.hs_v, d.S_pad = head_stride, head_stride, row_stride_v
        d.heads_per_group, d.gs_qk, d.gs_v = heads_per_group, group_stride, group_stride
        d.sub_heads, d.gs2_qk, d.gs2_v = sub_heads, sub_stride, sub_stride
    else:
        d.hs_qk, d.hs_v, d.S_pad = Qh.stride(0), Vt.stride(0), Vt.shape[2]
    d.n_tok, d.tok_off, d.H, d.eps, d.q_scale = n_tok, tok_off, H, eps, q_scale
    d.skip_qk = int(bool(skip_qk))      # q / k came out of the GEMM's fused epilogue (make_gemm_desc(qk_post=...)): V transpose only
    ctx.check(ctx.lib.utx_qkv_post(ctx.handle, C.byref(d), ctx.stream()))


def ln_mod(x, shift, scale, out=None, eps=1e-6):
    ctx = get_ctx(x.device.index)
    if out is None:
        out = torch.empty_like(x)
    d = LnModDesc()
    d.x, d.ldx, d.shift, d.scale = ptr(_bf(x)), x.stride(0), ptr(_bf(shift)), ptr(_bf(scale))
    d.y, d.ldy, d.n_tok, d.D, d.eps = ptr(out), out.stride(0), x.shape[0], x.shape[1], eps
    ctx.check(ctx.lib.utx_ln_mod(ctx.handle, C.byref(d), ctx.stream()))
    return out


def sched_step(x, v, dsigma, n_noise_tokens=None, cond=None):
    """in-place: x[:n_noise] += dsigma * v[:n_noise] (fp32 math); x[n_noise:] = cond."""
    ctx = get_ctx(x.device.index)
    tok, ch = x.shape
    n_noise = tok if n_noise_tokens is None else n_noise_tokens
    d = SchedDesc()
    d.x, d.v, d.cond = ptr(_bf(x)), ptr(_bf(v)), ptr(cond)
    d.n_noise_elems, d.n_total_elems, d.dsigma = n_noise * ch, tok * ch, float(dsigma)
    ctx.check(ctx.lib.utx_sched_step(ctx.handle, C.byref(d), ctx.stream()))
    return x
