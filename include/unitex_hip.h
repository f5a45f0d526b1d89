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

#define UTX_VERSION 100

typedef struct utx_ctx utx_ctx;
typedef void* utx_stream;   /* hipStream_t */

int utx_version(void);
int utx_init(int device, utx_ctx** ctx);
void utx_free(utx_ctx* ctx);
const char* utx_last_error(utx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * FLUX DiT denoise step  (flux_piplines/texturing/pipeline.py:634-681; transformer blocks are
 * diffusers' FluxTransformer2DModel [3p], attention core restated in
 * flux_piplines/texturing/attention_processor.py:31-110)
 * ---------------------------------------------------------------------------------------------- */

/* Flash attention forward, bf16 in/out, fp32 softmax + accumulation, non-causal, head_dim 128.
 * Replaces F.scaled_dot_product_attention at attention_processor.py:89-91.
 *   q, k : element (h, s, d) at base + h*{q,k}_hs + s*{q,k}_ss + d
 *   vt   : V transposed, element (h, d, s) at base + h*vt_hs + d*vt_ds + s; every row must be readable
 *          (finite) up to the next multiple of 64 past S
 *   o    : element (s, h, d) at base + s*o_ss + h*128 + d
 * Strides in elements; q_ss, k_ss, vt_ds multiples of 8, o_ss multiple of 4. */
int utx_attn_fwd_bf16(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                      long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                      int H, int S, float softmax_scale, utx_stream stream);

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
} utx_gemm_desc;
int utx_gemm_bf16(utx_ctx* ctx, const utx_gemm_desc* d, utx_stream stream);

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
} utx_qkv_post_desc;
int utx_qkv_post(utx_ctx* ctx, const utx_qkv_post_desc* d, utx_stream stream);

/* LayerNorm(no affine) + AdaLN modulation (AdaLayerNormZero/ZeroSingle/Continuous [3p]). */
typedef struct utx_ln_mod_desc {
    const void* x; long ldx;
    const void* shift; const void* scale;    /* [D] bf16 */
    void* y; long ldy;
    int n_tok, D;
    float eps;
} utx_ln_mod_desc;
int utx_ln_mod(utx_ctx* ctx, const utx_ln_mod_desc* d, utx_stream stream);

/* Flow-match Euler step + condition re-pin (flux_piplines/texturing/pipeline.py:644-645,660). */
typedef struct utx_sched_desc {
    void* x; const void* v; const void* cond;
    long n_noise_elems, n_total_elems;
    float dsigma;
} utx_sched_desc;
int utx_sched_step(utx_ctx* ctx, const utx_sched_desc* d, utx_stream stream);

/* sizeof() of the descriptor structs above, in declaration order (ABI self-check for FFI mirrors). */
int utx_abi_sizes(int* out, int n);

#ifdef __cplusplus
}
#endif
#endif
