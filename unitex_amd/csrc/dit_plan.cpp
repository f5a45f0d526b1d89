// utx_dit_load: the FLUX.1-dev transformer step as a utx_plan, built in C (SURVEY 8b `utx_dit_load` / `utx_dit_step`).
//
// The same launch list unitex_amd/flux/transformer.py::FluxDiT._build produces for the bf16 single-GPU path -- descriptor for descriptor (the GPU test
// tests/test_dit_ops_gpu.py::test_c_built_dit_plan_equals_the_python_built_one compares the two plans entry by entry, byte by byte) -- from plain C structs: the
// caller owns every buffer (weights packed the way FluxDiT packs them: fused [q;k;v] / [q;k;v;proj_mlp] projections, all AdaLN linears concatenated, LoRA pairs
// concatenated along rank per output segment) and every workspace (sizes follow from utx_dit_config; nothing is allocated here).  What the step computes and
// which reference lines it replaces is documented at the kernels and in transformer.py: double blocks with the text half beside the image half on the plan's side
// stream, joint attention with the key multiplicity of de-duplicated text tokens, single blocks with the fused [q|k|v|mlp] projection, last-block pruning to the
// n_out rows whose prediction is read, AdaLayerNormContinuous + proj_out; with utx_dit_config.fp8 the MX fp8 form of the same step (FluxDiT fp8_weights: the big
// image-stream linears on e4m3 operands + E8M0 scales, LayerNorm-modulation and the GELU epilogues emitting the next operand as fp8, the pruned last block on a
// second activation scratch).  NOT here (Python builder only): sequence parallelism, the fused q / k epilogue option.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "kernels.h"

extern "C" {
int utx_plan_assign_sk(utx_plan* p, void* sk_work, size_t sk_work_bytes, int n_cus);      // plan.cpp
}

namespace {
typedef const char* cptr;
inline cptr off(const void* base, long elems) { return (cptr)base + elems * 2; }        // bf16 element offset

struct Builder {
    utx_plan* plan;
    const utx_dit_config& c;
    const utx_dit_workspace& w;
    int D, H, S, S_pad;
    long Kmax;      // widest activation = row length of the fp8 scratch: (1 + mlp_ratio) D
    int rc;
    Builder(utx_plan* p, const utx_dit_config& c_, const utx_dit_workspace& w_) : plan(p), c(c_), w(w_), rc(0) {
        H = c.num_heads; D = H * 128; S = c.S_txt + c.S_img; S_pad = (S + 63) / 64 * 64; Kmax = (long)(1 + c.mlp_ratio) * D;
    }
    void chk(int r) { if (r != 0 && rc == 0) rc = r; }

    utx_gemm_desc gemm_desc(const void* A, long lda, int M, int K, const void* B, int N, void* C, long ldc, const void* bias) const {
        utx_gemm_desc d; memset(&d, 0, sizeof(d));
        d.A = A; d.lda = lda; d.B = B; d.ldb = K;
        d.K2 = 0; d.lora_n_limit = 0; d.lora_seg_n = 128;
        d.M = M; d.N = N; d.K = K;
        d.alpha = 1.0f; d.bias = bias; d.gelu_from = N;
        d.C = C; d.ldc = ldc; d.n_split = N;
        return d;
    }
    void add_gemm(utx_gemm_desc d, bool main_stream) { (void)main_stream; chk(utx_plan_add_gemm(plan, &d)); }
    // (optional LoRA-down GEMM +) main GEMM, as FluxDiT._gemm: T = the LoRA temp of this stream; limit / seg = 0: the whole output, one segment
    void linear(const utx_dit_linear& L, const void* A, long lda, int M, int K, int N, void* C, long ldc, bool main_stream, const void* T, int lora_limit, int lora_seg,
                int gelu_from = -1, const void* gate = nullptr, const void* res = nullptr, long ldres = 0, int n_split = -1, void* C1 = nullptr, long ldc1 = 0) {
        utx_gemm_desc d = gemm_desc(A, lda, M, K, L.w, N, C, ldc, L.b);
        if (L.lora_A && L.lora_rp > 0) {
            const int nr = L.lora_nseg * L.lora_rp;
            utx_gemm_desc d0 = gemm_desc(A, lda, M, K, L.lora_A, nr, (void*)T, 3L * c.lora_rank_padded, nullptr);
            d0.alpha = L.lora_alpha;
            add_gemm(d0, main_stream);
            d.A2 = T; d.lda2 = 3L * c.lora_rank_padded; d.B2 = L.lora_B; d.ldb2 = L.lora_rp; d.K2 = L.lora_rp;
            d.lora_n_limit = lora_limit > 0 ? lora_limit : N;
            d.lora_seg_n = lora_seg > 0 ? lora_seg : N;
        }
        if (gelu_from >= 0) d.gelu_from = gelu_from;
        d.gate = gate;
        if (gate) { d.res = res; d.ldres = ldres; }
        if (n_split >= 0) d.n_split = n_split;
        if (C1) { d.C1 = C1; d.ldc1 = ldc1; }
        add_gemm(d, main_stream);
    }
    // row0 >= 0: the result leaves as MX fp8 into the activation scratch rows [row0, ...) columns [0, D) + tile-packed scales (FluxDiT._lnmod mxq)
    void lnmod(const void* x, void* y, const void* shift, const void* scale, int n_tok, int mxq_row0 = -1) {
        utx_ln_mod_desc d; memset(&d, 0, sizeof(d));
        d.x = x; d.ldx = D; d.shift = shift; d.scale = scale; d.y = y; d.ldy = D; d.n_tok = n_tok; d.D = D; d.eps = 1e-6f;
        if (mxq_row0 >= 0) { d.q = (char*)w.aq + (long)mxq_row0 * Kmax; d.ldq = Kmax; d.qs = w.asp; d.qs_row_blocks = w.asp_row_blocks; }
        chk(utx_plan_add_ln_mod(plan, &d));
    }
    // ---- MX fp8 (FluxDiT._build mx() / _gemm mx8): ops.mx8_uses_packed -- the one-wave-per-SIMD MX kernel's constraints and the bf16 dispatch's fill rule
    static bool shape_packed(int M, int N, int n_split = -1, int gelu_from = -1) {
        if (N % 256 || (n_split >= 0 && n_split < N && n_split % 256) || (gelu_from >= 0 && gelu_from < N && gelu_from % 256)) return false;
        return (long)((M + 255) / 256) * (N / 256) >= 192;
    }
    bool is_mx(const utx_dit_linear& L) const { return c.fp8 && L.q; }
    bool is_packed(const utx_dit_linear& L, int M, int N, int n_split = -1, int gelu_from = -1) const { return is_mx(L) && L.sp && shape_packed(M, N, n_split, gelu_from); }
    void quant(const void* x, long ldx, void* q, long ldq, void* s, long lds_or_rb, int M, int K, int packed) { chk(utx_plan_add_quant_mx8(plan, x, ldx, q, ldq, s, lds_or_rb, M, K, packed)); }
    struct MxOpt { int quant_cols; int a_col0; bool q_out; MxOpt() : quant_cols(-1), a_col0(0), q_out(false) {} };
    // linear() on fp8 operands: the activation [M, K] is quantised into scratch rows [row0, row0 + M) (columns [0, quant_cols); the rest was written by its
    // producer), adapters are part of L.q.  packed = tile-packed scales / one-wave-per-SIMD kernel, else row-major scales / 128^2-tile kernel.
    void linear_mx(const utx_dit_linear& L, bool packed, MxOpt o, int row0, const void* A, long lda, int M, int K, int N, void* C, long ldc, int gelu_from = -1,
                   const void* gate = nullptr, const void* res = nullptr, long ldres = 0, int n_split = -1, void* C1 = nullptr, long ldc1 = 0) {
        char* aqv = (char*)w.aq + (long)row0 * Kmax + (packed ? o.a_col0 : 0);
        utx_gemm_desc d = gemm_desc(aqv, Kmax, M, K, L.q, N, C, ldc, L.b);
        if (packed) {
            char* asv = (char*)w.asp + (long)(o.a_col0 / 128) * w.asp_row_blocks * 512;
            const int nq = o.quant_cols < 0 ? K : o.quant_cols;
            if (nq > 0) quant(A, lda, aqv, Kmax, asv, w.asp_row_blocks, M, nq, 1);
            d.a_scale = asv; d.lds_a = w.asp_row_blocks; d.b_scale = L.sp; d.lds_b = L.sp_row_blocks; d.mx8 = 2;
            if (o.q_out) { d.q_out = (char*)w.aq + (long)row0 * Kmax + D; d.ldq_out = Kmax; d.qs_out = w.asp; d.qs_out_rb = w.asp_row_blocks; d.q_out_kt0 = D / 128; }
        } else {
            char* asv = (char*)w.as_rm + (long)row0 * (Kmax / 32);
            quant(A, lda, aqv, Kmax, asv, Kmax / 32, M, K, 0);
            d.a_scale = asv; d.lds_a = Kmax / 32; d.b_scale = L.s; d.lds_b = L.lds_s; d.mx8 = 1;
        }
        if (gelu_from >= 0) d.gelu_from = gelu_from;
        d.gate = gate;
        if (gate) { d.res = res; d.ldres = ldres; }
        if (n_split >= 0) d.n_split = n_split;
        if (C1) { d.C1 = C1; d.ldc1 = ldc1; }
        add_gemm(d, true);
    }
    void qkvpost(const void* qkv, const void* wq, const void* wk, int n_tok, int tok_off) {
        utx_qkv_post_desc d; memset(&d, 0, sizeof(d));
        d.qkv = qkv; d.ld = 3L * D; d.q_col = 0; d.k_col = D; d.v_col = 2 * D;
        d.wq = wq; d.wk = wk; d.cosb = w.cos; d.sinb = w.sin;
        d.Qh = w.Qh; d.Kh = w.Kh; d.Vt = w.Vt;
        d.hs_qk = (long)S_pad * 128; d.hs_v = 128L * S_pad; d.S_pad = S_pad;
        d.n_tok = n_tok; d.tok_off = tok_off; d.H = H; d.eps = 1e-6f;
        d.q_scale = (float)((1.0 / sqrt(128.0)) * 1.4426950408889634);
        chk(utx_plan_add_qkv_post(plan, &d));
    }
    void gemv(const void* x, int K, const utx_dit_linear& L, int N, void* y, int silu_in, int silu_out) {
        utx_gemv_desc d; memset(&d, 0, sizeof(d));
        d.x = x; d.ldx = K; d.W = L.w; d.ldw = K; d.bias = L.b; d.y = y; d.ldy = N; d.M = 1; d.N = N; d.K = K; d.silu_in = silu_in; d.silu_out = silu_out;
        chk(utx_plan_add_gemv(plan, &d));
    }
    void attn(void* out, long o_ss, int r0, int r1) {      // queries = token rows [r0, r1); out starts at row r0
        chk(utx_plan_add_attn(plan, off(w.Qh, (long)r0 * 128), w.Kh, w.Vt, out, (long)S_pad * 128, 128, (long)S_pad * 128, 128, 128L * S_pad, S_pad, o_ss, H, r1 - r0, S, 0.0f,
                              c.key_bias_log2, c.key_bias_period, w.attn_work, w.attn_work_bytes));
    }
};
}  // namespace

extern "C" int utx_dit_load(utx_ctx* ctx, const utx_dit_config* cfg, const utx_dit_weights* wt, const utx_dit_workspace* ws, utx_plan** out) {
    if (!cfg || !wt || !ws || !out || !wt->dbl || !wt->sgl) return -2;
    const utx_dit_config& c = *cfg;
    if (c.num_heads <= 0 || c.S_txt <= 0 || c.S_img <= 0 || c.num_double < 0 || c.num_single < 1 || c.mlp_ratio <= 0) return -2;
    if (c.fp8 && (!ws->aq || !ws->as_rm || !ws->asp || ws->asp_row_blocks < (c.S_txt + c.S_img + 127) / 128)) return -2;
    utx_plan* plan = nullptr;
    int rc = utx_plan_create(ctx, &plan);
    if (rc) return rc;
    Builder b(plan, c, *ws);
    const int D = b.D, S = b.S, S_txt = c.S_txt, S_img = c.S_img, MR = c.mlp_ratio;
    const int n_out = (c.n_out > 0 && c.n_out < S_img) ? c.n_out : S_img;
    const int ncu = c.n_cus > 0 ? c.n_cus : 256;
    const void* mod = ws->mod;
    auto chunk = [&](int o, int j) { return (const void*)off(mod, (long)o + (long)j * D); };
    cptr h = (cptr)ws->h, xn = (cptr)ws->xn, qkv = (cptr)ws->qkv, cat = (cptr)ws->cat, attn = (cptr)ws->attn;
    void* h_c = (void*)h; void* h_x = (void*)off(h, (long)S_txt * D);
    void* xn_c = (void*)xn; void* xn_x = (void*)off(xn, (long)S_txt * D);
    const long ldcat = (long)(1 + MR) * D;
    // ---- conditioning embeddings
    b.gemv(ws->tproj, 256, wt->t_lin1, D, ws->e1, 0, 1);
    b.gemv(ws->e1, D, wt->t_lin2, D, ws->e_t, 0, 0);
    if (c.guidance_embeds) {
        b.gemv(ws->gproj, 256, wt->g_lin1, D, ws->e1, 0, 1);
        b.gemv(ws->e1, D, wt->g_lin2, D, ws->e_g, 0, 0);
    }
    b.gemv(ws->pooled, c.pooled_dim, wt->p_lin1, D, ws->e1, 0, 1);
    b.gemv(ws->e1, D, wt->p_lin2, D, ws->e_p, 0, 0);
    b.chk(utx_plan_add_add3(plan, ws->e_t, c.guidance_embeds ? ws->e_g : nullptr, ws->e_p, ws->temb, D));
    b.gemv(ws->temb, D, wt->mod, wt->n_mod, ws->mod, 1, 0);
    // ---- embedders
    b.linear(wt->x_embedder, ws->lat, c.in_channels, S_img, c.in_channels, D, h_x, D, true, nullptr, 0, 0);
    b.linear(wt->context_embedder, ws->enc, c.joint_dim, S_txt, c.joint_dim, D, h_c, D, true, nullptr, 0, 0);
    const void* T = ws->T;
    const void* Tc = c.two_streams ? ws->Tc : ws->T;
    void* ff_x = (void*)off(cat, (long)S_txt * ldcat);      // double-block MLP hidden aliases the single-block cat buffer (its first 4D columns)
    void* ff_c = (void*)cat;
    const bool fuse = c.fp8 && c.fp8_fuse_quant;
    auto section = [&](auto&& side, auto&& mainf) {          // FluxDiT._par: two op lists side by side (plan fork / main / join) or one after the other
        if (c.two_streams) { b.chk(utx_plan_fork(plan)); side(false); b.chk(utx_plan_main(plan)); mainf(true); b.chk(utx_plan_join(plan)); }
        else { mainf(true); side(true); }
    };
    for (int i = 0; i < c.num_double; ++i) {
        const utx_dit_double_block& B = wt->dbl[i];
        const int ox = B.mod_x, oc = B.mod_c;
        const bool x_qkv = b.is_mx(B.qkv_x), x_ff1 = b.is_mx(B.ff1_x), x_ff2 = b.is_mx(B.ff2_x);
        const bool p_qkv = b.is_packed(B.qkv_x, S_img, 3 * D), p_ff1 = b.is_packed(B.ff1_x, S_img, MR * D), p_ff2 = b.is_packed(B.ff2_x, S_img, D);
        const bool f_qkv = fuse && p_qkv, f_ff = fuse && p_ff1 && p_ff2;
        Builder::MxOpt o_qkv, o_ff1, o_ff2;
        if (f_qkv) o_qkv.quant_cols = 0;
        if (f_ff) { o_ff1.quant_cols = 0; o_ff1.q_out = true; o_ff2.quant_cols = 0; o_ff2.a_col0 = D; }
        section(
            [&](bool ms) {
                b.lnmod(h_c, xn_c, chunk(oc, 0), chunk(oc, 1), S_txt);
                b.linear(B.qkv_c, xn_c, D, S_txt, D, 3 * D, (void*)qkv, 3L * D, ms, Tc, 3 * D, D);
                b.qkvpost(qkv, B.naq, B.nak, S_txt, 0);
            },
            [&](bool ms) {
                b.lnmod(h_x, xn_x, chunk(ox, 0), chunk(ox, 1), S_img, f_qkv ? S_txt : -1);
                if (x_qkv) b.linear_mx(B.qkv_x, p_qkv, o_qkv, S_txt, xn_x, D, S_img, D, 3 * D, (void*)off(qkv, (long)S_txt * 3 * D), 3L * D);
                else b.linear(B.qkv_x, xn_x, D, S_img, D, 3 * D, (void*)off(qkv, (long)S_txt * 3 * D), 3L * D, ms, T, 3 * D, D);
                b.qkvpost(off(qkv, (long)S_txt * 3 * D), B.nq, B.nk, S_img, S_txt);
            });
        b.attn((void*)attn, D, 0, S);
        section(
            [&](bool ms) {
                b.linear(B.out_c, attn, D, S_txt, D, D, h_c, D, ms, Tc, 0, 0, -1, chunk(oc, 2), h_c, D);
                b.lnmod(h_c, xn_c, chunk(oc, 3), chunk(oc, 4), S_txt);
                b.linear(B.ff1_c, xn_c, D, S_txt, D, MR * D, ff_c, ldcat, ms, Tc, 0, 0, 0);
                b.linear(B.ff2_c, ff_c, ldcat, S_txt, MR * D, D, h_c, D, ms, Tc, 0, 0, -1, chunk(oc, 5), h_c, D);
            },
            [&](bool ms) {
                b.linear(B.out_x, off(attn, (long)S_txt * D), D, S_img, D, D, h_x, D, ms, T, 0, 0, -1, chunk(ox, 2), h_x, D);
                b.lnmod(h_x, xn_x, chunk(ox, 3), chunk(ox, 4), S_img, f_ff ? S_txt : -1);
                if (x_ff1) b.linear_mx(B.ff1_x, p_ff1, o_ff1, S_txt, xn_x, D, S_img, D, MR * D, ff_x, ldcat, 0);
                else b.linear(B.ff1_x, xn_x, D, S_img, D, MR * D, ff_x, ldcat, ms, T, 0, 0, 0);
                if (x_ff2) b.linear_mx(B.ff2_x, p_ff2, o_ff2, S_txt, ff_x, ldcat, S_img, MR * D, D, h_x, D, -1, chunk(ox, 5), h_x, D);
                else b.linear(B.ff2_x, ff_x, ldcat, S_img, MR * D, D, h_x, D, ms, T, 0, 0, -1, chunk(ox, 5), h_x, D);
            });
    }
    for (int i = 0; i < c.num_single; ++i) {
        const utx_dit_single_block& B = wt->sgl[i];
        const int o = B.mod;
        const bool pruned = (i == c.num_single - 1) && n_out < S_img;
        const bool x_qkvm = !pruned && b.is_mx(B.qkvm), x_out = !pruned && b.is_mx(B.out);
        const bool p_qkvm = !pruned && b.is_packed(B.qkvm, S, (3 + MR) * D, 3 * D, 3 * D), p_out = !pruned && b.is_packed(B.out, S, D);
        const bool f_sgl = fuse && p_qkvm && p_out;
        b.lnmod(h, (void*)xn, chunk(o, 0), chunk(o, 1), S, f_sgl ? 0 : -1);
        if (pruned && c.fp8 && B.qkvm.q && B.qkvm.sp && B.out.q && B.out.sp && Builder::shape_packed(n_out, D) && Builder::shape_packed(S, 2 * D)) {
            // the pruned block on MX fp8 operands: x_n quantised once for k | v over all rows and once more, as a matrix of its own (tile-packed scales are
            // addressed from a 128-row-aligned origin; r0 = the text rows is not one), for the rows that keep a query -- second scratch aq2 / asp2
            if (!ws->aq2 || !ws->asp2 || ws->asp2_row_blocks < (n_out + 127) / 128) { b.chk(-2); break; }
            const int r0 = S_txt, r1 = S_txt + n_out;
            const utx_dit_linear& L = B.qkvm;
            cptr Wq = (cptr)L.q, bm = (cptr)L.b;
            const long Kmax = b.Kmax;
            auto sp_rows = [&](const utx_dit_linear& X, int row) { return (const void*)((cptr)X.sp + (long)(row / 128) * 512); };      // PackedScales.row_slice
            auto mx_desc = [&](const void* A, int M, int K, const void* Bq, int N, void* Cc, long ldc, const void* bias, const void* as_, int a_rb, const void* bs_, int b_rb) {
                utx_gemm_desc d = b.gemm_desc(A, Kmax, M, K, Bq, N, Cc, ldc, bias);
                d.a_scale = as_; d.lds_a = a_rb; d.b_scale = bs_; d.lds_b = b_rb; d.mx8 = 2;
                return d;
            };
            b.quant(xn, D, ws->aq, Kmax, ws->asp, ws->asp_row_blocks, S, D, 1);
            b.add_gemm(mx_desc(ws->aq, S, D, Wq + (long)D * D, 2 * D, (void*)off(qkv, D), 3L * D, off(bm, D), ws->asp, ws->asp_row_blocks, sp_rows(L, D), L.sp_row_blocks), true);
            b.quant(off(xn, (long)r0 * D), D, ws->aq2, Kmax, ws->asp2, ws->asp2_row_blocks, n_out, D, 1);
            b.add_gemm(mx_desc(ws->aq2, n_out, D, Wq, D, (void*)off(qkv, (long)r0 * 3 * D), 3L * D, bm, ws->asp2, ws->asp2_row_blocks, sp_rows(L, 0), L.sp_row_blocks), true);
            {
                utx_gemm_desc d = mx_desc(ws->aq2, n_out, D, Wq + 3L * D * D, MR * D, (void*)off(cat, (long)r0 * ldcat + D), ldcat, off(bm, 3L * D), ws->asp2, ws->asp2_row_blocks,
                                          sp_rows(L, 3 * D), L.sp_row_blocks);
                d.gelu_from = 0;
                b.add_gemm(d, true);
            }
            b.qkvpost(qkv, B.nq, B.nk, S, 0);
            b.attn((void*)off(cat, (long)r0 * ldcat), ldcat, r0, r1);
            b.quant(off(cat, (long)r0 * ldcat), ldcat, ws->aq2, Kmax, ws->asp2, ws->asp2_row_blocks, n_out, (int)Kmax, 1);
            {
                utx_gemm_desc d = mx_desc(ws->aq2, n_out, (int)Kmax, B.out.q, D, (void*)off(h, (long)r0 * D), D, B.out.b, ws->asp2, ws->asp2_row_blocks, B.out.sp, B.out.sp_row_blocks);
                d.gate = chunk(o, 2); d.res = off(h, (long)r0 * D); d.ldres = D;
                b.add_gemm(d, true);
            }
            continue;
        }
        if (pruned) {
            // LAST block: keys / values for every token, query / MLP / output projection for rows [r0, r1) only (transformer.py, "set_output_rows")
            const int r0 = S_txt, r1 = S_txt + n_out;
            const utx_dit_linear& L = B.qkvm;
            cptr Wm = (cptr)L.w, bm = (cptr)L.b;
            const bool lora = L.lora_A && L.lora_rp > 0;
            const int R = L.lora_rp;
            const long ldT = 3L * c.lora_rank_padded;
            if (lora) {
                utx_gemm_desc d0 = b.gemm_desc(xn, D, S, D, L.lora_A, 3 * R, (void*)T, ldT, nullptr);
                d0.alpha = L.lora_alpha;
                b.add_gemm(d0, true);
            }
            {   // k | v over all rows
                utx_gemm_desc d = b.gemm_desc(xn, D, S, D, off(Wm, (long)D * D), 2 * D, (void*)off(qkv, D), 3L * D, off(bm, D));
                if (lora) { d.A2 = off(T, R); d.lda2 = ldT; d.B2 = off(L.lora_B, (long)D * R); d.ldb2 = R; d.K2 = R; d.lora_n_limit = 2 * D; d.lora_seg_n = D; }
                b.add_gemm(d, true);
            }
            {   // q for the rows that are read
                utx_gemm_desc d = b.gemm_desc(off(xn, (long)r0 * D), D, r1 - r0, D, Wm, D, (void*)off(qkv, (long)r0 * 3 * D), 3L * D, bm);
                if (lora) { d.A2 = off(T, (long)r0 * ldT); d.lda2 = ldT; d.B2 = L.lora_B; d.ldb2 = R; d.K2 = R; d.lora_n_limit = D; d.lora_seg_n = D; }
                b.add_gemm(d, true);
            }
            {   // GELU(proj_mlp) for those rows
                utx_gemm_desc d = b.gemm_desc(off(xn, (long)r0 * D), D, r1 - r0, D, off(Wm, 3L * D * D), MR * D, (void*)off(cat, (long)r0 * ldcat + D), ldcat, off(bm, 3L * D));
                d.gelu_from = 0;
                b.add_gemm(d, true);
            }
            b.qkvpost(qkv, B.nq, B.nk, S, 0);
            b.attn((void*)off(cat, (long)r0 * ldcat), ldcat, r0, r1);
            {
                utx_gemm_desc d = b.gemm_desc(off(cat, (long)r0 * ldcat), ldcat, r1 - r0, (1 + MR) * D, B.out.w, D, (void*)off(h, (long)r0 * D), D, B.out.b);
                d.gate = chunk(o, 2); d.res = off(h, (long)r0 * D); d.ldres = D;
                b.add_gemm(d, true);
            }
            continue;
        }
        Builder::MxOpt o_qkvm, o_out;
        if (f_sgl) { o_qkvm.quant_cols = 0; o_qkvm.q_out = true; o_out.quant_cols = D; }      // GELU(mlp) leaves as fp8 (columns D..); only the attention output is quantised
        if (x_qkvm) b.linear_mx(B.qkvm, p_qkvm, o_qkvm, 0, xn, D, S, D, (3 + MR) * D, (void*)qkv, 3L * D, 3 * D, nullptr, nullptr, 0, 3 * D, (void*)off(cat, D), ldcat);
        else b.linear(B.qkvm, xn, D, S, D, (3 + MR) * D, (void*)qkv, 3L * D, true, T, 3 * D, D, 3 * D, nullptr, nullptr, 0, 3 * D, (void*)off(cat, D), ldcat);
        b.qkvpost(qkv, B.nq, B.nk, S, 0);
        b.attn((void*)cat, ldcat, 0, S);
        if (x_out) b.linear_mx(B.out, p_out, o_out, 0, cat, ldcat, S, (1 + MR) * D, D, (void*)h, D, -1, chunk(o, 2), h, D);
        else b.linear(B.out, cat, ldcat, S, (1 + MR) * D, D, (void*)h, D, true, T, 0, 0, -1, chunk(o, 2), h, D);
    }
    // AdaLayerNormContinuous (scale, shift) + proj_out on the rows that are read
    b.lnmod(h_x, xn_x, chunk(wt->mod_out, 1), chunk(wt->mod_out, 0), n_out);
    b.linear(wt->proj_out, xn_x, D, n_out, D, c.in_channels, ws->out, c.in_channels, true, nullptr, 0, 0);
    if (b.rc) { utx_plan_free(plan); return b.rc; }
    utx_plan_assign_sk(plan, ws->sk_work, ws->sk_work_bytes, ncu);      // FluxDiT._assign_streamk
    *out = plan;
    return 0;
}

extern "C" int utx_dit_step(utx_plan* plan, utx_stream stream, int* failed_entry) { return utx_plan_run(plan, stream, failed_entry); }
