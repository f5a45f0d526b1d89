/*
 * libunitex_hip.so -- C ABI of the MI355X (gfx950) hot path of UniTEX.
 *
 * Drop-in boundary for: CustomRGBTextureFullPipeline's FLUX-DiT denoise loop and the TextureTools
 * render / UV back-projection.  Every entry point below names the reference
 * interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *  - plain C: raw device pointers + sizes, no torch / C++ types.  All device memory is owned by the
 *    caller (PyTorch-ROCm allocations in the Python host); the library borrows pointers for the
 *    duration of a call and never frees or retains them, except opaque handles it allocates itself
 *    (utx_ctx, utx_bvh) that the caller releases with utx_free / utx_bvh_free.
 *  - every launch is stream-ordered on the caller's `stream` (a hipStream_t passed as void*), no
 *    hidden synchronisation unless stated.
 *  - return 0 on success, negative error code otherwise; utx_last_error(ctx) gives a message.
 *    The reference signals errors by Python exceptions/asserts (e.g.
 *    TextureTools/texturetools/render/nvdiffrast/renderer_inverse.py:171,256-261); the Python shim
 *    (unitex_amd/_lib.py) turns non-zero codes into RuntimeError.
 *  - bf16 tensors are passed as uint16_t storage.
 */
#ifndef UNITEX_HIP_H
#define UNITEX_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported (tests/test_abi.py compares `nm -D` with
 * this file); its internal launchers are not part of the ABI. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define UTX_VERSION 100

typedef struct utx_ctx utx_ctx;
typedef void* utx_stream;   /* hipStream_t */

int utx_version(void);
int utx_init(int device, utx_ctx** ctx);
void utx_free(utx_ctx* ctx);
const char* utx_last_error(utx_ctx* ctx);

/* Launch options (kernel selection / scheduling A/B).  Every one is RESULT-PRESERVING: identical bits, except that the three
 * tail-split switches (UTX_ATTN_TAILSPLIT, UTX_GEMM_TAILSPLIT, UTX_GEMM_STREAMK) change the fp32 summation order inside the tiles
 * they split -- the same result up to rounding, deterministic from launch to launch.  The first utx_init reads the environment
 * variables of the same names once; later changes go through utx_set_option only (no getenv on the launch path).  Names:
 * UTX_ATTN_GLDS, UTX_ATTN_FAST, UTX_ATTN_Q64, UTX_ATTN_TPB, UTX_ATTN_TAILSPLIT, UTX_GEMM_GROUP_M, UTX_GEMM_TILE, UTX_GEMM_TAILSPLIT,
 * UTX_GEMM_STREAMK, UTX_GEMM_PERS_GRID, UTX_GEMM_PERS_SCHED, UTX_BVH_STACK_WALK, UTX_BVH_PACKET, UTX_ATTN_PEEL, UTX_ATTN8_PEEL, UTX_NN_GRID.  UTX_ATTN_Q64 (default 1: the
 * 4 x 64 attention kernel for the launches it takes) keeps the bits wherever the 8 x 32 kernel does not re-centre its running maximum behind the first 32 keys; where it does, the two
 * kernels return two roundings of the same softmax.  The timing-ablation switches that compute wrong
 * results (UTX_ATTN_VAR, UTX_ATTN_DEBUG, UTX_GEMM_DEBUG) exist only in the separately built libunitex_hip_ablate.so (tools/ only): in
 * this library they return -7 and the environment variables are ignored.  Returns -2 for an unknown name. */
int utx_set_option(const char* name, int value);
int utx_get_option(const char* name, int* value);
int utx_is_ablation_build(void);

/* ------------------------------------------------------------------------------------------------
 * FLUX DiT denoise step  (flux_piplines/texturing/pipeline.py:634-681; transformer blocks are
 * diffusers' FluxTransformer2DModel [3p], attention core restated in
 * flux_piplines/texturing/attention_processor.py:31-110)
 * ---------------------------------------------------------------------------------------------- */

/* Flash attention forward, bf16 in/out, fp32 softmax + accumulation, non-causal, head_dim 128.
 * Replaces F.scaled_dot_product_attention at attention_processor.py:89-91.
 *   q, k : element (h, s, d) at base + h*{q,k}_hs + s*{q,k}_ss + d; k rows must be readable (finite)
 *          up to the next multiple of 64 past S (keys >= S are masked, never used)
 *   vt   : V transposed, element (h, d, s) at base + h*vt_hs + d*vt_ds + s; every row must be readable
 *          (finite) up to the next multiple of 64 past S
 *   o    : element (s, h, d) at base + s*o_ss + h*128 + d
 * Strides in elements; q/k/vt base pointers 16-byte aligned; q_hs, k_hs, vt_hs, q_ss, k_ss, vt_ds multiples of 8,
 * o_ss multiple of 4.
 * softmax_scale > 0: the reference's scale (1/sqrt(128)); softmax_scale == 0: Q was pre-multiplied by
 * scale*log2(e) by utx_qkv_post (q_scale) and scores are used as base-2 exponents directly.
 * One call may issue up to five stream-ordered operations (a memset of the flag bytes, full rounds of workgroups, the key-split tail round, its merge, the repair pass).
 * Since round 6 the launches that qualify -- softmax_scale == 0, S a multiple of 64, contiguous operands, key_bias_period == 0, scratch present -- run the 4 x 64 kernel
 * (attention_q64.hip: one wave per SIMD, generated instruction stream) followed by its repair pass; all others the 8 x 32 kernel (attention_glds.hip).
 * The tail round needs scratch.  utx_attn_fwd_bf16_ws takes it from the caller (utx_attn_workspace_bytes: nothing is allocated on the
 * launch path, re-entrant per stream / buffer).  The entry points without a workspace argument use a buffer owned by the CONTEXT, grown
 * with hipMalloc by the first call that needs more (never while the stream is being captured: such a launch stays unsplit -- same
 * result up to one bf16 rounding of the tail rows); calls through one context must not run concurrently on different streams. */
int utx_attn_fwd_bf16(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                      long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                      int H, int S, float softmax_scale, utx_stream stream);
/* The same with KEY MULTIPLICITY: every key of tile 0 (keys 0..63) -- and, with key_bias_period = n > 0, of every 64-key tile
 * whose index is a multiple of n -- stands for 2^key_bias_log2 identical keys: key_bias_log2 is added to its base-2 scores, which
 * is exactly softmax over the repeated keys.  Used by the text-token dedup of unitex_amd/flux/transformer.py: the reference feeds
 * 512 all-zero text embeddings with all-zero position ids (flux_piplines/texturing/pipeline.py:538-543), i.e. 512 IDENTICAL tokens
 * at every layer (last bullet); 64 of them are carried, each counting 8-fold.  key_bias_log2 = 0 is utx_attn_fwd_bf16. */
int utx_attn_fwd_bf16_kb(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                         long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                         int H, int S, float softmax_scale, float key_bias_log2, int key_bias_period, utx_stream stream);
/* The same with a QUERY COUNT OF ITS OWN: rows 0 .. S_q-1 of q are the queries (o gets S_q rows), keys / values are all S_kv tokens; q and k / vt are separate arrays and
 * S_q may be smaller or (since round 6: the sequence-parallel launch over de-duplicated text keys, utx_sp_unpack_qkv_dedup) larger than S_kv.  S_q < S_kv is used by
 * the last transformer block: the pipeline discards the prediction of the condition tokens (flux_piplines/texturing/pipeline.py:645,660,684:
 * the condition tail of the latents is re-pinned before every transformer call and cut off at the end), so only the noise tokens need
 * a query in the block whose output nobody attends to any more (unitex_amd/flux/transformer.py, `set_output_rows`). */
int utx_attn_fwd_bf16_kbq(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                          long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                          int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period, utx_stream stream);
/* The same with CALLER-OWNED scratch for the key-split tail round: work >= utx_attn_workspace_bytes(ctx, H, S_q, S_kv) bytes, 16-byte aligned, used only
 * by the launches of this call (stream-ordered); layout [key-split scratch | one flag byte per 64-query group (the 4 x 64 kernel's headroom record)].  work == NULL or too small:
 * the tail round is not split and the 8 x 32 kernel runs.  Both sizing calls follow the QUERY count and accept S_q > S_kv like the launch itself.  utx_attn_plan (pure host arithmetic,
 * no device): out = {workgroups, workgroups in full rounds of n_cus, key ranges per tail workgroup (1 = not split), 64-key tiles per range}. */
int utx_attn_fwd_bf16_ws(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                         long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                         int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period,
                         void* work, size_t work_bytes, utx_stream stream);
size_t utx_attn_workspace_bytes(utx_ctx* ctx, int H, int S_q, int S_kv);

/* MX fp8 attention -- OPT-IN (BASELINE configs[4] "fp8 MFMA"; the reference has no fp8 path: the contract is this library's own and is a different numerics
 * contract from the bf16 kernel above, with its own stated tolerance: tests/test_attention_fp8_gpu.py).  Same joint attention as utx_attn_fwd_bf16_kbq
 * (attention_processor.py:31-110: softmax(Q K^T) V per head over the joint sequence; Q arrives pre-scaled by scale * log2(e), so scores are base-2 exponents)
 * with Q K^T and P V on the fp8 matrix pipe: Q8 / K8 [H][S_pad][128] e4m3 bytes + qs / ks [H][S_pad] dwords (four E8M0 bytes per row: the 32-channel blocks
 * of d) -- what utx_quant_mx8 makes of the head-major Q / K viewed as [H * S_pad, 128] with lds = 4; V8^T [H][128][S_pad] e4m3 + vs [H][S_pad / 32][32][4]
 * E8M0 bytes per (block of 32 keys, channel d % 32, d / 32) -- what utx_quant_vt_mx8 makes of V^T [H][128][S_pad] bf16.  P is quantised to e4m3 with the unit
 * scale inside the kernel (p <= 256 by the kernel's re-centring rule; p < 2^-9 flush to zero).  Output o [S_q][o_ss] bf16, head h in columns 128 h ..; S_q <=
 * S_kv queries (the first S_q rows of Q8), S_pad = rows allocated (multiple of 64, pad rows zero); key_bias_* as utx_attn_fwd_bf16_kb. */
int utx_quant_vt_mx8(utx_ctx* ctx, const void* vt, void* v8, void* vs, int H, int S_pad, utx_stream stream);
int utx_attn_fwd_fp8(utx_ctx* ctx, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                     int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, utx_stream stream);
/* The same with CALLER-OWNED scratch for the key-split tail round (round 6: the bf16 kernel's plan, work items and merge): work >= utx_attn_workspace_bytes(ctx, H, S_q, S_kv)
 * bytes, 16-byte aligned; null / too small: the launch stays unsplit.  utx_attn_fwd_fp8 uses the context's per-stream scratch (none while the stream is capturing). */
int utx_attn_fwd_fp8_ws(utx_ctx* ctx, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                        int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes, utx_stream stream);
int utx_attn_plan(int H, int S_q, int S_kv, int n_cus, int out[4]);
/* The same on BLOCK-STRIDED operands: the S_kv tokens (queries and keys alike) come in blocks of blk_rows (a multiple of 64 that divides S_kv); block b of Q / K /
 * V^T of a head starts q_bs / k_bs / vt_bs elements (multiples of 8) behind block b - 1, rows inside a block are q_ss / k_ss apart, V^T rows vt_ds (each holding the
 * blk_rows columns of its block).  This is the receive buffer of the sequence-parallel Q / K / V all-to-all, [source rank][q | k | v][head][S_loc x 128]
 * (unitex_amd/flux/ulysses.py), consumed where RCCL put it instead of behind a relayout pass; the output rows stay in token order (block-major), o_ss apart.
 * Same tiles in the same order as the contiguous call: bit-identical results.  Replaces nothing in the reference (it is single-GPU); beyond it, SURVEY 8e. */
int utx_attn_fwd_bf16_blk(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                          long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                          int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period,
                          void* work, size_t work_bytes, int blk_rows, long q_bs, long k_bs, long vt_bs, utx_stream stream);

/* C = epi(alpha * (A B^T + A2 B2^T) + bias): bf16 GEMM, fp32 accumulate, fused epilogues.
 * Replaces every nn.Linear (+ peft LoRA branch, + GELU, + gated residual) inside the FLUX blocks
 * (attention_processor.py:42-45,60-63,101-106; pipeline.py:108-112 LoRA adapters;
 * trainer.py:282-295 LoRA targets). See unitex_amd/csrc/gemm.hip for the epilogue definitions. */
typedef struct utx_gemm_desc {
    const void* A; long lda;      /* [M,K]  bf16, K contiguous */
    const void* B; long ldb;      /* [N,K]  bf16 (torch Linear weight) */
    const void* A2; long lda2;    /* optional LoRA segment: [M, nseg*K2] = s * x A_lora^T */
    const void* B2; long ldb2;    /* [N,K2] LoRA up weight */
    int M, N, K, K2;              /* K, K2 multiples of 64; N multiple of 8 */
    int lora_n_limit;             /* output columns < limit take the LoRA segment */
    int lora_seg_n;               /* A2 column block for column n is (n / lora_seg_n) * K2 */
    float alpha;
    const void* bias;             /* [N] bf16 or NULL */
    int gelu_from;                /* columns >= gelu_from get GELU(tanh); N for none */
    const void* gate;             /* [N] bf16 or NULL: out = res + gate * y */
    const void* res; long ldres;  /* residual [M,N] bf16 (may alias C) */
    void* C; long ldc;
    int n_split; void* C1; long ldc1;  /* columns >= n_split go to C1[m, n - n_split]; N for none */
    int ntn;                      /* internal */
    /* implicit 3x3 convolution (conv_Wo > 0): A is an NHWC activation [Hi*Wi, Cin] (Cin = 1 << conv_cin_log2 >= 64),
     * B the weight [N, 9*Cin] with K order (ky, kx, cin), M = Ho*Wo output pixels, K = 9*Cin.  Tap (ky,kx) of output
     * pixel (oy,ox) reads input (oy*stride + ky - pad, ox*stride + kx - pad) of the (conv_up ? 2x nearest-upsampled
     * : plain) image; out-of-range taps read zero_page (>= 128 zero bytes).  Replaces torch Conv2d(3x3) (+ the
     * F.pad(0,1,0,1) stride-2 downsampler and the nearest-2x upsampler) of the FLUX VAE [3p]. */
    int conv_Hi, conv_Wi, conv_Wo, conv_cin_log2, conv_stride, conv_pad, conv_up;
    const void* zero_page;
    /* OCP MX fp8 base segment (mx8 = 1; BASELINE configs[4] "fp8 MFMA weights"): A [M,K] and B [N,K] hold e4m3 bytes (lda / ldb
     * in bytes, K a multiple of 128) and a_scale [M][K/32] / b_scale [N][K/32] one E8M0 byte per 32 elements along K (row strides
     * lds_a / lds_b bytes, multiples of 4); element value = e4m3 * 2^(scale - 127).  Accumulation in fp32 by
     * v_mfma_scale_f32_32x32x64_f8f6f4; the LoRA segment (A2 / B2) stays bf16; epilogues unchanged.  Quantise activations with
     * utx_quant_mx8; weights once at load (unitex_amd/flux/mx8.py).  Replaces the same nn.Linear as the bf16 form.
     * mx8 = 2: the same operands with TILE-PACKED scales (utx_quant_mx8_packed; lds_a / lds_b = row blocks of 128 per K-tile slab of a_scale /
     * b_scale) on the persistent one-wave-per-SIMD kernel: N and every column boundary (n_split, gelu_from) multiples of 256, no LoRA segment
     * (K2 = 0: merge the adapters into the weight before quantising), no fused q / k epilogue; anything else is refused (-2), never dropped. */
    const void* a_scale; long lds_a;
    const void* b_scale; long lds_b;
    int mx8;
    /* Fused q / k post-processing (qk_cols > 0; the "fused QKV + RMSNorm + RoPE" of the hot path; one-wave-per-SIMD kernel only -- a wave owns
     * a whole 128-column head there): output columns n < qk_cols are NOT written to C; each 128-column head slab gets what utx_qkv_post does
     * to q (columns < qk_cols / 2) or k (the rest): per-head RMSNorm with qk_wq / qk_wk, RoPE from qk_cos / qk_sin ([tok][64] fp32, token =
     * qk_tok_off + row), q scaled by qk_q_scale, head-major store to qk_Qh / qk_Kh ([H][S_pad][128], head stride qk_hs elements) -- same
     * arithmetic, same rounding points, same summation order, bit-identical to GEMM -> utx_qkv_post (attention_processor.py:42-87).  Columns
     * >= qk_cols (v, the MLP half of a single block) take the plain epilogue; v is transposed by utx_qkv_post with skip_qk = 1. */
    int qk_cols;                       /* 0 = off; else 2 * H * 128 */
    int qk_tok_off;
    float qk_eps, qk_q_scale;
    const void* qk_wq; const void* qk_wk;
    const float* qk_cos; const float* qk_sin;
    void* qk_Qh; void* qk_Kh; long qk_hs;
    /* Optional scratch for the balanced tail of the one-wave-per-SIMD kernel (UTX_GEMM_STREAMK, default on): when the 256 x 256 tiles do not
     * fill the last round of workgroups, the K loops of that round's tiles are cut into equal ranges over all CUs, the fp32 partial tiles
     * pass through sk_work and a second small kernel sums them in K order and runs the epilogue (deterministic; differs from the unsplit
     * result only by fp32 summation order).  Caller-owned, >= utx_gemm_streamk_workspace_bytes(), one buffer per stream that launches
     * GEMMs concurrently; NULL = never split. */
    void* sk_work; size_t sk_work_bytes;
    /* mx8 == 2 only, optional: the GELU part of the output (columns n >= gelu_from) leaves as OCP MX fp8 -- the A operand of the NEXT mx8 = 2 GEMM
     * (FLUX: GELU(ff.net.0) feeds ff.net.2; GELU(proj_mlp) feeds proj_out) -- instead of bf16: bytes q_out [M][ldq_out] at column n - gelu_from,
     * tile-packed scales qs_out (qs_out_rb row blocks per K-tile slab), K-tile q_out_kt0 + (n - gelu_from) / 128.  Bit-identical to the bf16
     * epilogue followed by utx_quant_mx8_packed (rounded to bf16, then quantised per block of 32 columns); C / C1 are not written for those
     * columns.  (N - gelu_from) % 128 == 0, ldq_out % 8 == 0.  The split tail round is not used by such a launch. */
    void* q_out; long ldq_out;
    void* qs_out; long qs_out_rb;
    int q_out_kt0;
} utx_gemm_desc;
int utx_gemm_bf16(utx_ctx* ctx, const utx_gemm_desc* d, utx_stream stream);
size_t utx_gemm_streamk_workspace_bytes(utx_ctx* ctx);   /* size of utx_gemm_desc.sk_work for this device (2 partial tiles per CU) */
/* What utx_gemm_bf16 would do with this descriptor on a device of n_cus compute units under the current launch options -- the library's own dispatch
 * arithmetic, no device needed (the host uses it to decide which projections can take the fused q / k epilogue, and bench.py to report the launch census):
 * out[0] = kernel (0 128x128 tiles, 1 one wave per SIMD, 2 persistent 8-wave, 3 per-tile 8-phase, 4 two-barrier 256^2), out[1] = output tiles of that kernel (128^2 for kernel 0, else 256^2), out[2] = tiles
 * of the last round that are cut along K (0: unsplit; needs d->sk_work != NULL), out[3] = K ranges per such tile. */
int utx_gemm_plan(const utx_gemm_desc* d, int n_cus, int out[4]);

/* OCP MX fp8 quantisation of a bf16 matrix along its rows (the activation operand of an mx8 GEMM):
 *   per block of 32 consecutive elements: e = floor(log2(max|x|)) - 8 (clamped to [-127, 127]; -127 for an all-zero block),
 *   scale byte = e + 127, q = e4m3_rne(clamp(x * 2^-e, -448, 448)).
 * x [M][ldx] bf16 (K % 32 == 0, ldx % 8 == 0) -> q [M][ldq] bytes, s [M][lds] bytes (K/32 per row).  oracle/mx8_ref.py restates it. */
int utx_quant_mx8(utx_ctx* ctx, const void* x, long ldx, void* q, long ldq, void* s, long lds, int M, int K, utx_stream stream);
/* The same quantiser (identical q bytes and scale values) with the scales TILE-PACKED for utx_gemm_desc.mx8 == 2 (the one-wave-per-SIMD kernel,
 * unitex_amd/csrc/gemm_w4.hip): s is an array of dwords [K/128][row_blocks][32][4]; dword [kt][rb][l][im] holds the four E8M0 bytes of the
 * 32-element blocks 4 kt .. 4 kt + 3 of row 128 rb + 32 im + l (byte j = block 4 kt + j).  row_blocks >= ceil(M / 128); K % 128 == 0; s 16-byte
 * aligned, K/128 * row_blocks * 512 bytes.  Rows >= M of the last row block are not written.  A lane of the GEMM then fetches the scale dwords of
 * its four fragment rows for a K-tile with ONE 16-byte load (512 contiguous bytes per wave) instead of four strided dword gathers. */
int utx_quant_mx8_packed(utx_ctx* ctx, const void* x, long ldx, void* q, long ldq, void* s, long row_blocks, int M, int K, utx_stream stream);

/* y[m,n] = act_out(sum_k act_in(x[m,k]) W[n,k] + b[n]) for M <= 8 (embedders, AdaLN modulation). */
typedef struct utx_gemv_desc {
    const void* x; long ldx;
    const void* W; long ldw;
    const void* bias;
    void* y; long ldy;
    int M, N, K;
    int silu_in, silu_out;
} utx_gemv_desc;
int utx_gemv_bf16(utx_ctx* ctx, const utx_gemv_desc* d, utx_stream stream);

/* ---- FLUX AutoencoderKL pieces (diffusers [3p]; encode at flux_piplines/texturing/pipeline.py:226-238, decode at
 * :683-692).  Activations are NHWC bf16 [H*W, C].  Convolutions with Cin >= 64 are utx_gemm_bf16 in conv mode. */
/* GroupNorm(32 groups, C % 128 == 0) + affine (+ SiLU): work = utx_group_norm_workspace_bytes() bytes. */
size_t utx_group_norm_workspace_bytes(void);
int utx_group_norm(utx_ctx* ctx, const void* x, long npix, int C, const void* gamma, const void* beta, float eps, int silu,
                   void* y, void* work, utx_stream stream);
/* in-place row softmax of a bf16 matrix [nrow, ncol] (row stride ld elements; ncol, ld multiples of 8). */
int utx_softmax_rows(utx_ctx* ctx, void* s, long nrow, long ld, int ncol, utx_stream stream);
/* 3x3 / stride 1 / pad 1 convolution for thin inputs (Cin = 3 or 16): wt is [9*Cin][Cout] bf16 (tap-major). */
int utx_conv3x3_thin(utx_ctx* ctx, const void* x, int H, int W, int Cin, const void* wt, const void* bias, int Cout, void* y,
                     utx_stream stream);

/* per-head RMSNorm(q,k) + RoPE + head-major relayout + V transpose
 * (attention_processor.py:50-57,76-87). */
typedef struct utx_qkv_post_desc {
    const void* qkv; long ld;
    int q_col, k_col, v_col;
    const void* wq; const void* wk;          /* [128] bf16 RMSNorm weights */
    const float* cosb; const float* sinb;    /* [S_total][64] fp32 rotary tables */
    void* Qh; void* Kh; void* Vt;            /* [H][S_pad][128], [H][S_pad][128], [H][128][S_pad] */
    long hs_qk, hs_v, S_pad;
    int n_tok, tok_off, H;
    float eps;
    float q_scale;                           /* multiplies Q before its bf16 rounding (1 = reference layout;
                                                scale*log2(e) feeds utx_attn_fwd_bf16(softmax_scale = 0)) */
    /* grouped head addressing (sequence-parallel send layout, unitex_amd/flux/ulysses.py): with heads_per_group = g > 0 head h
     * lives at (h / g) * gs + (h % g) * hs instead of h * hs, i.e. the kernel writes straight into the all-to-all send buffer
     * [P][3][H/P][...] (one group per destination rank) and no pack pass exists.  0 = plain [H][...] layout. */
    int heads_per_group;
    long gs_qk, gs_v;
    int skip_qk;                             /* 1: q and k were produced by the GEMM's fused epilogue (utx_gemm_desc.qk_cols): only V is transposed */
    /* second level of the grouped addressing (0 = off): the g heads of a destination rank are cut into head groups of sub_heads heads whose
     * exchanges are pipelined with attention (send layout [group][P][3][sub_heads][...]): head hi = h % g of a rank lives at
     * (h / g) * gs + (hi / sub_heads) * gs2 + (hi % sub_heads) * hs.  g % sub_heads == 0. */
    int sub_heads;
    long gs2_qk, gs2_v;
} utx_qkv_post_desc;
int utx_qkv_post(utx_ctx* ctx, const utx_qkv_post_desc* d, utx_stream stream);

/* Sequence-parallel ("Ulysses") exchange, receive side (unitex_amd/flux/ulysses.py; no counterpart in the single-GPU reference,
 * flux_piplines/texturing/pipeline.py:633-681 -- this is SURVEY 8(e)'s parity-preserving split of the ONE joint sequence).
 * Both are pure 16-byte-vector copies (HBM-bound), one launch each; S = P * S_loc, S_loc % 64 == 0, E = S_loc * 128.
 *   utx_sp_unpack_qkv: recv [P src][3][Hp][E]  ->  q, k [Hp][S][128] (row = src*S_loc + tok),  vt [Hp][128][S]
 *                      (recv[src][0|1][hp] is [S_loc][128]; recv[src][2][hp] is [128][S_loc])
 *   utx_sp_unpack_qkv_dedup: the same for q; k [Hp][S_k][128] and vt [Hp][128][S_k] with S_k = text_rows + P (S_loc - text_rows): the first text_rows tokens of
 *                      EVERY source rank's block are the same rows (the caller's guarantee: the identical text tokens every rank carries, flux/transformer.py) and
 *                      are kept ONCE, from source rank 0, followed by the other tokens of rank 0 .. P-1 -- the key order of the single-GPU sequence, so that a rank's
 *                      attention launch (S_q = S queries over S_kv = S_k keys) is the single-GPU launch over H / P heads.  text_rows % 64 == 0, < S_loc
 *   utx_sp_unpack_o  : recv [P src][S_loc][Hp*128]  ->  out [S_loc][ld]: columns src*Hp*128 .. of row tok
 *   utx_sp_unpack_o_cols: the same for ONE head group of several (Hp = heads of the group): source rank src lands at column src * src_cols
 *                      of `out` (src_cols = all heads of a rank x 128; `out` points at the group's first column) */
int utx_sp_unpack_qkv(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* q, void* k, void* vt, utx_stream stream);
int utx_sp_unpack_qkv_dedup(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, int text_rows, void* q, void* k, void* vt, utx_stream stream);
int utx_sp_unpack_o(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* out, long ld, utx_stream stream);
int utx_sp_unpack_o_cols(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* out, long ld, long src_cols, utx_stream stream);

/* LayerNorm(no affine) + AdaLN modulation (AdaLayerNormZero/ZeroSingle/Continuous [3p]). */
typedef struct utx_ln_mod_desc {
    const void* x; long ldx;
    const void* shift; const void* scale;    /* [D] bf16 */
    void* y; long ldy;
    int n_tok, D;
    float eps;
    /* q != NULL (D % 128 == 0): the result leaves as OCP MX fp8 instead of bf16 -- exactly what utx_quant_mx8_packed makes of the bf16 result
     * (the value is rounded to bf16 first, then quantised per block of 32): bytes q [n_tok][ldq], tile-packed scales qs (qs_row_blocks row
     * blocks per K-tile slab, see utx_quant_mx8_packed).  The activation operand of an mx8 = 2 GEMM without the bf16 round trip through HBM;
     * y is not written. */
    void* q; long ldq;
    void* qs; long qs_row_blocks;
} utx_ln_mod_desc;
int utx_ln_mod(utx_ctx* ctx, const utx_ln_mod_desc* d, utx_stream stream);

/* Flow-match Euler step + condition re-pin (flux_piplines/texturing/pipeline.py:644-645,660). */
typedef struct utx_sched_desc {
    void* x; const void* v; const void* cond;
    long n_noise_elems, n_total_elems;
    float dsigma;
} utx_sched_desc;
int utx_sched_step(utx_ctx* ctx, const utx_sched_desc* d, utx_stream stream);


/* ------------------------------------------------------------------------------------------------
 * TextureTools render / UV back-projection
 * (TextureTools/texturetools/render/nvdiffrast/renderer_inverse.py:159-365,574-633,
 *  raytracing/rt_aprmis, texture/stitching/mip.py, image/lens_blur.py)
 * ---------------------------------------------------------------------------------------------- */

/* clip[n][v] = [x y z 1] . mvp[n]^T ; ndc (optional, may be NULL) = clip.xy / clip.w
 * (renderer_inverse.py:263-265).  verts [V][3], mvp [n][4][4] row-major, clip [n][V][4], ndc [n][V][2]. */
int utx_transform_points(utx_ctx* ctx, const float* verts, int V, const float* mvp, int n_views,
                         float* clip, float* ndc, utx_stream stream);

/* dr.rasterize replacement (renderer_inverse.py:183,273): pos [V][4] clip space, tri [F][3] int32,
 * rast [H][W][4] = (u, v, z/w, triangle_id + 1), 0 = empty.  work: utx_rasterize_workspace_bytes. */
long utx_rasterize_workspace_bytes(int F, int H, int W);
int utx_rasterize(utx_ctx* ctx, const float* pos, const int* tri, int F, int H, int W, float* rast,
                  void* work, utx_stream stream);

/* dr.interpolate replacement (renderer_inverse.py:188,277,288): attr [V][C] -> out [npix][C]. */
int utx_interpolate(utx_ctx* ctx, const float* attr, int C, const float* rast, const int* tri, long npix,
                    float* out, utx_stream stream);

/* geometry-condition shading of VideoExporter.export_condition (video/export_nvdiffrast_video.py:956-989):
 * rast [npix][4], interpolated vertex normals / positions [npix][3] -> uint8 normal / ccm images [npix][3]
 * and alpha [npix], composited on bg3_host (HOST array of 3 floats), uint8 by truncation. */
int utx_condition_shade(utx_ctx* ctx, const float* rast, const float* nrm, const float* pos, const float* bg3_host,
                        long npix, void* out_normal, void* out_ccm, void* out_alpha, utx_stream stream);

/* per-face unit normals of PBRMesh (mesh/structure_v2.py:49-50): verts [V][3], faces [F][3] -> out [F][3]. */
int utx_face_normals(utx_ctx* ctx, const float* verts, const int* faces, int F, float* out, utx_stream stream);

/* textured shading of the orbit video (video/export_nvdiffrast_video.py:141-256 -> renderer_base.py:289-336
 * uv_rendering): rast [npix][4], per-vertex uv [V][2] in [0,1], tex [Ht][Wt][3] fp32 in UV-raster orientation (row
 * grows with v) -> uint8 RGB [npix][3]: bilinear fetch (wrap), background bg3_host where empty, truncation to uint8. */
int utx_texture_shade(utx_ctx* ctx, const float* rast, const float* uv, const int* tri, const float* tex, int Ht, int Wt,
                      const float* bg3_host, long npix, void* out, utx_stream stream);

/* LBVH ray-mesh intersector (raytracing/__init__.py:12-83 RayTracing / rt_aprmis APRMISRayTracing; the build the
 * reference runs per mesh: raytracing/rt_aprmis/bvhhelpers.py:20-84).  verts/faces are borrowed and must stay alive while
 * the handle is used.
 * utx_bvh_build: the handle owns its node arrays (ONE hipMalloc inside build, released by utx_bvh_free); the tree depth is
 *   known on return (one stream synchronisation).
 * utx_bvh_build_ws (what NVDiffRendererInverse.infer uses): every array lives in the caller's `work` (256-byte aligned,
 *   >= utx_bvh_workspace_bytes(F); it must outlive the handle) -- nothing is allocated, freed or waited for on the path:
 *   the build is enqueued and returns; the first launch that needs the tree depth (utx_bvh_trace*, utx_backproject,
 *   utx_bvh_depth) waits for it, later ones do not.  utx_bvh_free releases the handle only. */
typedef struct utx_bvh utx_bvh;
int utx_bvh_build(utx_ctx* ctx, const float* verts, int V, const int* faces, int F, utx_bvh** out, utx_stream stream);
size_t utx_bvh_workspace_bytes(int F);
int utx_bvh_build_ws(utx_ctx* ctx, const float* verts, int V, const int* faces, int F, void* work, size_t work_bytes, utx_bvh** out, utx_stream stream);
void utx_bvh_free(utx_bvh* bvh);
/* device pointers to the node arrays (tests / diagnostics): info [2F-1][3], aabb [2F-1][6], sorted Morton
 * codes [F], sorted element ids [F]; returns F. */
int utx_bvh_arrays(utx_bvh* bvh, int** info, float** aabb, unsigned** codes_sorted, int** idx_sorted);
/* intersects_closest (rt_aprmis/__init__.py:40-86): tid [R] int32, -1 = miss. */
int utx_bvh_trace(utx_ctx* ctx, utx_bvh* bvh, const float* rays_o, const float* rays_d, long R, int* tid, utx_stream stream);
/* the same trace + the number of tree nodes the rays visited, ADDED to *visited (device counter; measurement of SURVEY 8d's
 * nodes-visited-per-ray, bench.py); longest root-to-leaf path of the tree (the stackless packed traversal is used up to 60, the
 * reference's 64-entry stack walk above: intersect_test2.slang:63-146). */
int utx_bvh_trace_count(utx_ctx* ctx, utx_bvh* bvh, const float* rays_o, const float* rays_d, long R, int* tid,
                        unsigned long long* visited, utx_stream stream);
int utx_bvh_depth(utx_bvh* bvh);

/* fused per-(view, texel) gather + visibility of uv_to_pcd (renderer_inverse.py:277-298,316-325) */
typedef struct utx_backproject_desc {
    const void* rast2d;                 /* [T][4] f32 UV-space raster */
    const void* verts; const void* faces; const void* fnormal;   /* [V][3] f32, [F][3] i32, [F][3] f32 */
    const void* vndc;                   /* [n_views][V][2] f32 */
    const void* dirs;                   /* [n_views][3] f32 = -c2w[:, :3, 2] (orthographic) */
    const void* images;                 /* [n_views][H][W][4] f32 (rgb + alpha) */
    void* color; void* rayvis; void* alphaok;   /* [n_views][T][3] f32, [n_views][T] u8, [n_views][T] u8 */
    int T_h, T_w, V, n_views, H, W;
    int view_begin, view_count;         /* views handled by this launch (per-GPU view sharding) */
    float cos_thresh, two_sqrt3;
} utx_backproject_desc;
int utx_backproject(utx_ctx* ctx, const utx_backproject_desc* d, utx_bvh* bvh, utx_stream stream);

/* visibility hole filling k=3,5 + AND coverage + AND alpha>0.999 (renderer_inverse.py:326-343).
 * rayvis/alphaok/vis_out/tmp: [n_views][H][W] u8. */
int utx_dilate_visibility(utx_ctx* ctx, const void* rayvis, const void* alphaok, const float* rast2d, int n_views,
                          int H, int W, void* tmp, void* vis_out, utx_stream stream);

/* first-come-wins priority composite (renderer_inverse.py:595-602).  order: HOST array of view ids.
 * colors [n][T][3] f32, vis [n][T] u8 -> atlas [T][3] f32, winner [T] int8 (-1 = unseen). */
int utx_composite(utx_ctx* ctx, const float* colors, const void* vis, const int* order_host, int n_order, long T,
                  float* atlas, void* winner, utx_stream stream);

/* seam mask (renderer_inverse.py:602-604): winner [H][W] int8 -> seam [H][W] u8; tmp [H][W] u8. */
int utx_seam_mask(utx_ctx* ctx, const void* winner, const float* rast2d, int H, int W, void* tmp, void* seam, utx_stream stream);

/* exact 3-D nearest-seen-texel fill of unseen covered texels, in place on atlas (renderer_inverse.py:606-615).
 * pos [T][3] f32; nn_index [T] int32 (optional): chosen source texel or -1. */
long utx_nn_fill_workspace_bytes(long T);
int utx_nn_fill(utx_ctx* ctx, const float* pos, const void* winner, const float* rast2d, long T, float* atlas,
                int* nn_index, void* work, long work_bytes, utx_stream stream);

/* view-space visibility with the gradient filter (mv_to_pcd, filt_gradient_points=True; renderer_inverse.py:189-209).
 * attr6 [n][H][W][6] f32 interpolated (position, vertex normal), rast [n][H][W][4], fnormal [F][3], dirs [n][3] ray directions.
 * tmp: 2*n*H*W bytes.  vis [n][H][W] u8, alpha [n][H][W] f32 (optional).  radius = 15 (the reference's 31-wide pool, along W). */
int utx_view_visibility(utx_ctx* ctx, const float* attr6, const float* rast, const float* fnormal, const float* dirs, int n, int H, int W,
                        float grad_thr, float cos_thr, int radius, void* tmp, void* vis, float* alpha, utx_stream stream);

/* exact k-NN gather in 3-D (bake_mv_to_uv_kdtree, renderer_inverse.py:367-433; search = torch_kdtree [3p], pcd/knn/__init__.py:103-113).
 * Sources and queries are dense arrays with optional byte masks (view pixels / atlas texels); k <= 32.
 * mode 0: out_attr = mean of the k neighbours' attributes; mode 1: MVPaint weighting (needs normals).
 * out_idx / out_d2 ([M][k], optional): neighbour ids (ascending squared distance, ties -> lower id), -1 / inf when fewer exist. */
typedef struct utx_knn_desc {
    const float* src_pos; const float* src_attr; const float* src_nrm; const unsigned char* src_mask; long N;
    const float* dst_pos; const float* dst_nrm; const unsigned char* dst_mask; long M;
    int k, C, mode;
    float* out_attr; int* out_idx; float* out_d2;
} utx_knn_desc;
long utx_knn_workspace_bytes(long N);
int utx_knn(utx_ctx* ctx, const utx_knn_desc* d, void* work, long work_bytes, utx_stream stream);

/* lens blur consumed on the seam only (image/lens_blur.py:260-280; renderer_inverse.py:620-624).
 * k49_host: HOST array, the collapsed real 7x7 kernel. src/dst [H][W][3] f32. */
int utx_lens_blur_seam(utx_ctx* ctx, const float* src, const void* seam, int H, int W, const float* k49_host, float* dst, utx_stream stream);

/* pull-push hole filling (texture/stitching/mip.py:51-95). kd/out [H][W][3] f32, mask [H][W] u8. */
long utx_pull_push_workspace_bytes(int H, int W);
int utx_pull_push(utx_ctx* ctx, const float* kd, const void* mask, int H, int W, float* out, void* work, utx_stream stream);

/* tensor_to_image (renderer_utils.py:62-83): clamp*255 -> u8 by truncation, optional vertical flip. */
int utx_to_u8(utx_ctx* ctx, const float* src, long n_rows, long row_elems, int flip, void* dst, utx_stream stream);

/* sizeof() of the descriptor structs above, in declaration order (ABI self-check for FFI mirrors). */

/* Chart labelling of the blank-mesh UV unwrap (pipeline.py:171-179 preprocess_blank_mesh -> uv_atlas.py:131-175: open3d / UVAtlas
 * [3p], replaced by a builder-defined chart unwrap, unitex_amd/texturetools/meshes.py `unwrap_charts`): connected components of
 * the face adjacency graph restricted to faces of equal `bucket`.  adj [F][3] = face across edge e or -1, bucket [F], chart [F]
 * out = smallest face index of the component (deterministic), flag = one device int of scratch.  Synchronises the stream
 * (convergence test).  Returns the number of sweeps (> 0) or a negative error. */
int utx_chart_flood(utx_ctx* ctx, const int* adj, const int* bucket, int F, int* chart, int* flag, utx_stream stream);

/* ---- utx_plan: one denoise step as a replayable list of launches (SURVEY 8b `utx_dit_step`) ------------------------------------------------
 * The host builds a step of the FLUX DiT once as a list of descriptors over fixed workspaces (unitex_amd/flux/transformer.py is the builder this
 * repo ships; any caller that can fill the descriptors can be one); utx_plan_add_* COPY them, utx_plan_run replays them on `stream` with the
 * same launchers as the single-call entry points above -- same kernels, same order, bit-identical -- in ONE C call per step, nothing allocated,
 * nothing synchronised (capturable into a HIP graph).  Inputs that change between steps (latents, timestep projection, conditioning) live in
 * the device buffers the descriptors point at.  Two-stream sections: utx_plan_fork; entries for the plan's side stream; utx_plan_main; entries for
 * the caller's stream; utx_plan_join (events recorded / awaited inside utx_plan_run).  utx_plan_add_add3: out = bf16(bf16(a + b) + c), n bf16
 * elements (b may be NULL) -- the conditioning-embedding sum of CombinedTimestepGuidanceTextProjEmbeddings [3p].
 * utx_plan_run returns 0 or the failing launcher's code (entry index in *failed_entry when not NULL). */
typedef struct utx_plan utx_plan;
int utx_plan_create(utx_ctx* ctx, utx_plan** out);
void utx_plan_free(utx_plan* plan);
int utx_plan_size(const utx_plan* plan);
int utx_plan_add_gemm(utx_plan* plan, const utx_gemm_desc* d);
int utx_plan_add_gemv(utx_plan* plan, const utx_gemv_desc* d);
int utx_plan_add_ln_mod(utx_plan* plan, const utx_ln_mod_desc* d);
int utx_plan_add_qkv_post(utx_plan* plan, const utx_qkv_post_desc* d);
int utx_plan_add_attn(utx_plan* plan, const void* q, const void* k, const void* vt, void* o, long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs,
                      long vt_ds, long o_ss, int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period, void* work,
                      size_t work_bytes);
int utx_plan_add_quant_mx8(utx_plan* plan, const void* x, long ldx, void* q, long ldq, void* s, long lds_or_row_blocks, int M, int K, int packed);
int utx_plan_add_add3(utx_plan* plan, const void* a, const void* b, const void* c, void* out, int n);
/* the opt-in MX fp8 attention as plan entries: arguments of utx_quant_vt_mx8 / utx_attn_fwd_fp8 (Q8 / K8 come from utx_plan_add_quant_mx8 over the
 * [H * S_pad, 128] row view: ldx = ldq = 128, lds = 4, packed = 0) */
int utx_plan_add_quant_vt_mx8(utx_plan* plan, const void* vt, void* v8, void* vs, int H, int S_pad);
int utx_plan_add_attn_fp8(utx_plan* plan, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                          int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period);
int utx_plan_add_attn_fp8_ws(utx_plan* plan, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                             int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes);
int utx_plan_fork(utx_plan* plan);
int utx_plan_main(utx_plan* plan);
int utx_plan_join(utx_plan* plan);
int utx_plan_run(utx_plan* plan, utx_stream stream, int* failed_entry);
/* entries [begin, end) only (whole two-stream sections): a host that interleaves its own work with the step -- the sequence-parallel host issues its
 * all-to-alls between ranges -- replays each range with one call.  A launcher error inside a forked section still joins the side stream before returning. */
int utx_plan_run_range(utx_plan* plan, int begin, int end, utx_stream stream, int* failed_entry);
int utx_plan_assign_sk(utx_plan* plan, void* sk_work, size_t sk_work_bytes, int n_cus);   /* split-tail scratch for the caller-stream GEMMs when any has more 256^2 tiles than CUs */
int utx_plan_entry(const utx_plan* plan, int i, int* kind, int* side, void* buf, size_t cap);   /* read an entry back: kind 0 gemm 1 gemv 2 ln_mod 3 qkv_post 4 attn 5 quant 6 add3 7 fork 8 join */

/* ---- utx_dit_load / utx_dit_step: the FLUX.1-dev transformer step built in C -------------------------------------------------------------
 * Replaces FluxTransformer2DModel.forward as the reference calls it (flux_piplines/texturing/pipeline.py:646-656; block arithmetic = diffusers [3p]) for the
 * bf16 single-GPU path: utx_dit_load builds the SAME utx_plan unitex_amd/flux/transformer.py::FluxDiT builds (compared entry by entry, byte by byte, in
 * tests/test_dit_ops_gpu.py::test_c_built_dit_plan_equals_the_python_built_one) from plain structs.  The caller owns everything: weights PACKED as FluxDiT packs
 * them -- double block: qkv_x = [to_q; to_k; to_v] (9216 x 3072), qkv_c = [add_q; add_k; add_v]; single block: qkvm = [to_q; to_k; to_v; proj_mlp] (21504 x
 * 3072); `mod` = every AdaLN modulation Linear concatenated (mod_x / mod_c / mod / mod_out = row offset of a block's slice); LoRA (optional per linear,
 * un-merged, peft semantics): lora_A = s x A of the switched-on adapters concatenated along rank, one block of lora_rp (multiple of 64) rows per output
 * segment (lora_nseg: 3 for the fused q|k|v projections, else 1), lora_B [N_lora, lora_rp], lora_alpha -- and every workspace (bf16 unless noted; S = S_txt +
 * S_img, S_pad = S rounded up to 64, D = 128 heads): lat [S_img, in], enc [S_txt, joint], pooled [1, pooled], tproj / gproj [1, 256] (sinusoidal timestep /
 * guidance projections, filled by the caller per step), e1 / e_t / e_g / e_p / temb [1, D], mod [n_mod], h / xn / attn [S, D], qkv [S, 3D], cat [S, 5D], out
 * [S_img, in], cos / sin [S, 64] fp32 (RoPE tables), Qh / Kh [H, S_pad, 128], Vt [H, 128, S_pad] (zero-filled once: pad rows stay zero), T [S, 3 R] and Tc
 * [S_txt, 3 R] (LoRA temps, R = lora_rank_padded = the largest lora_rp), sk_work (utx_gemm_streamk_workspace_bytes) and attn_work (utx_attn_workspace_bytes)
 * optional.  n_out < S_img: last-block pruning (only the first n_out image rows of `out` are defined); key_bias_*: multiplicity of de-duplicated text keys;
 * two_streams: the text half of the double blocks on the plan's side stream.  A step: write lat / tproj (and enc / pooled / gproj / cos / sin when they change),
 * utx_dit_step(plan, stream) = utx_plan_run, read out.  fp8 = 1: the MX fp8 step (BASELINE configs[4] numerics) -- the linears that carry q / s / sp run
 * on fp8 operands (their LoRA is ignored: merged by the caller), shapes that fill the chip on the one-wave-per-SIMD kernel with tile-packed scales and, with
 * fp8_fuse_quant, activations written as fp8 by LayerNorm-modulation / the GELU epilogues; the rest on the 128^2-tile kernel behind a quantiser pass.  Not in
 * the C builder: sequence parallelism (the collectives are the host's). */
typedef struct utx_dit_linear {
    const void* w; const void* b;
    const void* lora_A; const void* lora_B;
    float lora_alpha; int lora_rp; int lora_nseg;
    /* fp8 mode (utx_dit_config.fp8), the five big image-stream linears only: the weight once more as OCP MX fp8 -- adapters MERGED in before quantising
     * (W + sum s B A, rounded to bf16, then utx_quant_mx8) -- q = e4m3 bytes [N, K], s = row-major E8M0 scales [N, lds_s >= K / 32], sp = the same scales
     * tile-packed (utx_quant_mx8_packed; sp_row_blocks = row blocks per K-tile slab; NULL when N % 256) */
    const void* q; const void* s; long lds_s; const void* sp; int sp_row_blocks;
} utx_dit_linear;
typedef struct utx_dit_double_block {
    utx_dit_linear qkv_x, qkv_c, out_x, out_c, ff1_x, ff2_x, ff1_c, ff2_c;
    const void* nq; const void* nk; const void* naq; const void* nak;      /* RMSNorm weights [128] */
    int mod_x, mod_c;
} utx_dit_double_block;
typedef struct utx_dit_single_block {
    utx_dit_linear qkvm, out;
    const void* nq; const void* nk;
    int mod;
} utx_dit_single_block;
typedef struct utx_dit_weights {
    utx_dit_linear x_embedder, context_embedder, proj_out, t_lin1, t_lin2, g_lin1, g_lin2, p_lin1, p_lin2, mod;
    const utx_dit_double_block* dbl; const utx_dit_single_block* sgl;
    int mod_out, n_mod;
} utx_dit_weights;
typedef struct utx_dit_config {
    int num_heads, num_double, num_single, in_channels, joint_dim, pooled_dim, mlp_ratio, guidance_embeds;
    int S_txt, S_img, n_out;
    float key_bias_log2; int key_bias_period;
    int two_streams, n_cus, lora_rank_padded;
    int fp8, fp8_fuse_quant;      /* MX fp8 linears (qkv_x / ff1_x / ff2_x of the double blocks, qkvm / out of the single blocks); producers write fp8 operands directly */
} utx_dit_config;
typedef struct utx_dit_workspace {
    void *lat, *enc, *pooled, *tproj, *gproj, *e1, *e_t, *e_g, *e_p, *temb, *mod, *h, *xn, *qkv, *cat, *attn, *out;
    float *cos, *sin;
    void *Qh, *Kh, *Vt, *T, *Tc;
    void* sk_work; size_t sk_work_bytes;
    void* attn_work; size_t attn_work_bytes;
    /* fp8 mode: activation scratch aq [S, 5D] e4m3 bytes, as_rm [S, 5D / 32] row-major scales (shapes on the 128^2-tile MX kernel), asp = tile-packed scales
     * [5D / 128][asp_row_blocks >= ceil(S / 128)][512]; aq2 [n_out, 5D] / asp2 the same for the rows a pruned last block keeps (n_out < S_img only) */
    void *aq, *as_rm, *asp; int asp_row_blocks;
    void *aq2, *asp2; int asp2_row_blocks;
} utx_dit_workspace;
int utx_dit_load(utx_ctx* ctx, const utx_dit_config* cfg, const utx_dit_weights* weights, const utx_dit_workspace* ws, utx_plan** out);
int utx_dit_step(utx_plan* plan, utx_stream stream, int* failed_entry);

/* HOST-side mesh preparation (no device work, no context): quadric-error-metric edge-collapse decimation to at most target_faces triangles.
 * Replaces open3d's simplify_quadric_decimation in preprocess_blank_mesh_o3d (uv_atlas.py:155-163; open3d / VTK [3p]) with the published
 * algorithm (Garland & Heckbert 1997): area-weighted face quadrics, boundary edges held by perpendicular constraint planes
 * (boundary_weight, 1.0 in the reference's legacy call), optimal placement, flip rejection.  verts [V][3] f32, faces [F][3] i32 ->
 * verts_out (capacity V), faces_out (capacity F), counts in *V_out / *F_out.  Deterministic.  A collapse is admissible when no surviving face flips, the
 * edge is not a non-manifold fan and the link condition holds (the common neighbours of its ends are exactly the apexes of its faces).  Returns 0, or 1 when
 * no admissible collapse was left above target_faces (the output is valid, *F_out > target_faces), negative on invalid arguments. */
int utx_mesh_decimate_qem(const float* verts, int V, const int* faces, int F, int target_faces, double boundary_weight,
                          float* verts_out, int* faces_out, int* V_out, int* F_out);

int utx_abi_sizes(int* out, int n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
