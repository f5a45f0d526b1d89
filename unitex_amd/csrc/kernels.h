// Internal launcher prototypes shared by the .hip files and capi.cpp.  The parameter structs ARE
// the public descriptors of include/unitex_hip.h (single source of truth for the layout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unitex_hip.h"
#ifndef UTX_BF16_T
#define UTX_BF16_T
typedef uint16_t bf16_t;
#endif

struct AttnParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* vt;
    bf16_t* o;
    long q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss;  // element strides
    int H, S, nqb;
    int Sq;            // query rows (0 = S): attention_glds.hip only
    int dbg;           // perf ablation only (UTX_ATTN_DEBUG bits): 1 no staging, 2 no exp, 4 no barrier, 8 no PV, 16 no QK
    float scale_log2;  // softmax_scale * log2(e)
    unsigned char* flags;  // per (head, 64-query group) overflow marks: written by the 4 x 64 kernel, read by the repair pass (else null)
    int flag_hs;           // flags per head
    // tail split (attention_glds.hip): work items [w_base, ...) of this launch, each cut into nsplit key ranges of
    // tiles_per_split 64-key tiles; a split writes its normalised partial output + log2-sum-exp instead of the final rows
    int w_base, nsplit, tiles_per_split;
    bf16_t* part_o;        // [items][nsplit][256][128] bf16
    float* part_lse;       // [items][nsplit][256] f32: running maximum + log2(row sum)
    // key multiplicity (text-token dedup, flux/transformer.py): the keys of tile 0 -- and of every tile whose index is a multiple of
    // key_bias_period when that is > 0 -- stand for 2^key_bias_log2 identical keys each: key_bias_log2 is added to their scores
    float key_bias_log2;
    int key_bias_period;
    // caller-owned scratch of the tail split (utx_attn_workspace_bytes); null / too small: the launch stays unsplit
    void* work;
    size_t work_bytes;
    // block-strided operands (attention_glds.hip, BLK): tokens in blocks of blk_rows (0 = contiguous) lying q_bs / k_bs / vt_bs elements apart
    int blk_rows;
    long q_bs, k_bs, vt_bs;
};
// MX fp8 attention (attention_fp8.hip; opt-in): e4m3 operands with E8M0 block scales
struct Attn8Params {
    const uint8_t *q8, *k8;      // [H][S_pad][128] e4m3
    const uint8_t* v8t;          // [H][128][S_pad] e4m3
    const uint32_t *qs, *ks;     // [H][S_pad]: four E8M0 bytes per row (32-channel blocks of d)
    const uint32_t* vs;          // [H][S_pad / 32][32]: dword of row d % 32 = the E8M0 bytes of channels d % 32 + {0, 32, 64, 96} for that block of 32 keys
    bf16_t* o;
    long o_ss;
    int H, S, Sq, S_pad, nqb;
    float key_bias_log2;
    int key_bias_period;
    // key-split tail round (round 6; the bf16 kernel's plan, work items and merge -- AttnParams above): filled by the launcher from `work`
    int w_base, nsplit, tiles_per_split;
    bf16_t* part_o;
    float* part_lse;
    void* work;            // caller-owned scratch (utx_attn_workspace_bytes); null / too small: the launch stays unsplit
    size_t work_bytes;
};
// Launch options.  Every field is result-preserving (kernel selection / scheduling A/B): read ONCE from the environment by
// the first utx_init (UTX_ATTN_*, UTX_GEMM_* variables of the same names), afterwards changed only through utx_set_option.
// The three `*_abl` fields switch timing ablations that compute WRONG results; they exist only in the UTX_ABLATION build
// (libunitex_hip_ablate.so, used by tools/ -- never by the product, the tests or bench.py).
struct UtxOptions {
    int attn_glds;        // 1 (default): LDS-DMA staged attention kernel; 0: register-staged variants
    int attn_fast;        // register-staged kernel only: 2 block-pipelined sum-checked softmax, 1 sum-checked, 0 per-tile max
    int attn_q64;         // 1 (default since round 6): launches the 4 x 64 kernel takes (attention_q64.hip: pre-scaled Q, whole 64-key tiles, contiguous operands, caller scratch) run it
                          // + its repair pass; 0: the 8 x 32 kernel everywhere (A/B; bit-identical wherever the 8 x 32 kernel does not re-centre behind the first block)
    int attn_tpb;         // tiles per barrier of the LDS-DMA kernel (1 | 2)
    int attn_tailsplit;   // 1 (default): key-split tail round
    int gemm_group_m;     // 0 = built-in GROUP_M
    int gemm_tile;        // 0 auto, 128, 256 (per-tile 8-phase), 2560 (persistent), 2562 (2-barrier 256^2), 2564 (one wave per SIMD)
    int gemm_tailsplit;   // 1: K-split tail round of the 8-phase GEMM (off by default)
    int gemm_pers_grid;   // persistent GEMM: number of workgroups (0 = one per CU)
    int gemm_pers_sched;  // persistent GEMM: DMA placement over the phases of a K-tile: 0 = by shape, 1 = force SCHED 0, 2 = force SCHED 1
    int gemm_streamk;     // 1 (default): one-wave-per-SIMD GEMM balances the K loops of its last, partly filled round over all CUs (needs utx_gemm_desc.sk_work)
    int bvh_stack_walk;   // 1: the reference's stack walk over the unpacked tree instead of the stackless packed walk (A/B; same results)
    int attn_var_abl, attn_debug_abl, gemm_debug_abl;
    int bvh_packet;       // 1 (default): back-projection rays walk the tree as wave-wide packets over 8 x 8 texel tiles (bvh_trace_packet); 0: one thread per ray (A/B; same results)
    int attn_peel;        // 1 (default since round 5): the pre-scaled LDS-DMA attention launch runs its fast loop (attention_glds.hip, FAST: first / ragged tile outside the loop, the tile's
                          // barrier between S2 and S3, next tile's first K fragments read under S3); 0: the general loop (the default until round 4).  Same bits either way.
    int attn8_peel;       // 1 (default since round 5): MX fp8 attention with tile 0 / a ragged last tile outside the loop and the loop's exponentials in quarters under the PV MFMAs
                          // (attn_fwd_fp8_kernel<1>, attention_fp8.hip); 0: the general loop.  Same bits either way.
    int gemm_fastk;       // one-wave-per-SIMD GEMM (bf16): 1 (default since round 6: +2.3 ... +3.7 % on the FLUX shapes, profiles/r06_gemm_fastk_check_v0.log) = the steady-state K loop runs the generated instruction stream (gemm_w4_loop_asm.inc), 0 = hipcc's loop (round 2-5).  Same bits.
    int nn_grid;          // 0 (default): the NN fill's cell grid follows the atlas size; 64 | 128 | 256 force one (A/B and the grid-independence test; same results)
};
extern UtxOptions g_utx_opt;

typedef utx_gemm_desc GemmParams;
typedef utx_knn_desc KnnParams;
typedef utx_gemv_desc GemvParams;
typedef utx_qkv_post_desc QkvPostParams;
typedef utx_ln_mod_desc LnModParams;
typedef utx_sched_desc SchedParams;

extern "C" {
int utx_launch_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                        long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                        long o_ss, int H, int S, int Sq, float scale, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes, hipStream_t stream);
void utx_attn_split_plan_impl(int H, int Sq, int S, int ncu, int out[4]);     // attention_glds.hip (pure): {workgroups, in full rounds, key ranges per tail workgroup, tiles per range}
size_t utx_attn_workspace_bytes_impl(int H, int Sq, int S, int ncu);      // [tail-split scratch | 4 x 64 kernel's headroom flags]
size_t utx_attn_split_bytes_impl(int H, int Sq, int S, int ncu);
size_t utx_attn_q64_flag_bytes(int H, int Sq, int S);                    // attention_q64.hip
int utx_attn_q64_takes(const AttnParams* p, int presc);                  // attention_q64.hip: shape, layout and scratch fit the 4 x 64 kernel
int utx_launch_attn_merge(const AttnParams* t, int n_items, hipStream_t stream);
int utx_launch_attn_fwd_glds(const AttnParams* p, int presc, hipStream_t stream);
int utx_launch_attn_fwd_fp8(const Attn8Params* p, hipStream_t stream);                                   // attention_fp8.hip
int utx_launch_quant_vt_mx8(const void* vt, void* v8, void* vs, int H, int S_pad, hipStream_t stream);   // attention_fp8.hip
int utx_launch_attn_fwd_blk(const void* q, const void* k, const void* vt, void* o, long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                            long o_ss, int H, int S, int Sq, float scale, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes,
                            int blk_rows, long q_bs, long k_bs, long vt_bs, hipStream_t stream);
int utx_launch_attn_fwd_q64(const AttnParams* p, int presc, hipStream_t stream);
int utx_launch_gemm_bf16(const GemmParams* p, hipStream_t stream);
size_t utx_gemm_streamk_workspace_bytes_impl(void);
void utx_gemm_plan_impl(const GemmParams* p, int ncu, int sk_has_work, int out[4]);          // gemm.hip: kernel choice + split of the last round (pure)
void utx_gemm_w4_split_plan(const GemmParams* p, int tiles, int grid, int has_work, int* T, int* S);   // gemm_w4.hip (pure)
int utx_launch_gemm_w4(GemmParams p, hipStream_t stream);     // gemm_w4.hip: persistent 256x256 kernel, one wave per SIMD
int utx_launch_gemm_pers(GemmParams p, hipStream_t stream);   // gemm_pers.hip: persistent 256x256 kernel (large-M linears)
int utx_launch_gemv_bf16(const GemvParams* p, hipStream_t stream);
int utx_launch_quant_mx8(const void* x, long ldx, void* q, long ldq, void* s, long lds, int M, int K, hipStream_t stream);
int utx_launch_quant_mx8_packed(const void* x, long ldx, void* q, long ldq, void* s, long row_blocks, int M, int K, hipStream_t stream);
int utx_launch_qkv_post(const QkvPostParams* p, hipStream_t stream);
int utx_launch_sp_unpack_qkv(const void* recv, int P, int Hp, int S_loc, int text_rows, void* q, void* k, void* vt, hipStream_t stream);
int utx_launch_sp_unpack_o(const void* recv, int P, int Hp, int S_loc, void* out, long ld, long src_cols, hipStream_t stream);
size_t utx_group_norm_workspace_bytes_impl(void);
int utx_launch_group_norm(const void* x, long npix, int C, const void* gamma, const void* beta, float eps, int silu, void* y, void* work, hipStream_t stream);
int utx_launch_softmax_rows(void* s, long nrow, long ld, int ncol, hipStream_t stream);
int utx_launch_conv3x3_thin(const void* x, int H, int W, int Cin, const void* wt, const void* bias, int Cout, void* y, hipStream_t stream);
int utx_launch_ln_mod(const LnModParams* p, hipStream_t stream);
int utx_launch_sched_step(const SchedParams* p, hipStream_t stream);
int utx_launch_transform(const float* verts, int V, const float* mvp, int n_views, float* clip, float* ndc, hipStream_t stream);
int utx_launch_rasterize(const float* pos, const int* tri, int F, int H, int W, float* rast, void* work, hipStream_t stream);
int utx_launch_interpolate(const float* attr, int C, const float* rast, const int* tri, long npix, float* out, hipStream_t stream);
int utx_launch_condition_shade(const float* rast, const float* nrm, const float* pos, const float* bg3_host, long npix, void* out_normal, void* out_ccm, void* out_alpha, hipStream_t stream);
int utx_launch_face_normals(const float* verts, const int* faces, int F, float* out, hipStream_t stream);
int utx_launch_view_visibility(const float* attr6, const float* rast, const float* fnormal, const float* dirs, int n, int H, int W,
                               float grad_thr, float cos_thr, int radius, void* tmp, void* vis, float* alpha, hipStream_t stream);
size_t utx_knn_workspace_bytes_impl(long N);
int utx_launch_knn(const KnnParams* p, void* work, size_t work_bytes, hipStream_t stream);
int utx_launch_texture_shade(const float* rast, const float* uv, const int* tri, const float* tex, int Ht, int Wt, const float* bg3_host, long npix, void* out, hipStream_t stream);
int utx_bvh_build_impl(const float* verts, int V, const int* faces, int F, utx_bvh** out, hipStream_t stream);
size_t utx_bvh_workspace_bytes_impl(int F);
int utx_bvh_build_ws_impl(const float* verts, int V, const int* faces, int F, void* work, size_t work_bytes, utx_bvh** out, hipStream_t stream);
void utx_bvh_free_impl(utx_bvh* b);
int utx_bvh_arrays_impl(utx_bvh* b, int** info, float** aabb, unsigned** codes_sorted, int** idx_sorted);
int utx_bvh_trace_impl(utx_bvh* b, const float* ro, const float* rd, long R, int* tid, unsigned long long* visited, int force_stack, hipStream_t stream);
int utx_bvh_depth_impl(utx_bvh* b);
int utx_launch_backproject(const utx_backproject_desc* p, const utx_bvh* bvh, hipStream_t stream);
int utx_launch_dilate_visibility(const void* rayvis, const void* alphaok, const void* rast2d, int n_views, int Hh, int Ww, void* tmp, void* vis_out, hipStream_t stream);
int utx_launch_composite(const float* colors, const void* vis, const int* order, int n_order, long T, float* atlas, void* winner, hipStream_t stream);
int utx_launch_seam_mask(const void* winner, const float* rast2d, int Hh, int Ww, void* tmp, void* seam, hipStream_t stream);
size_t utx_nn_fill_workspace_bytes_impl(long T);
int utx_launch_nn_fill(const float* pos, const void* winner, const float* rast2d, long T, float* atlas, int* nn_index, void* work, size_t work_bytes, hipStream_t stream);
int utx_launch_lens_blur_seam(const float* src, const void* seam, int Hh, int Ww, const float* k49_host, float* dst, hipStream_t stream);
size_t utx_pull_push_workspace_bytes_impl(int Hh, int Ww);
int utx_launch_pull_push(const float* kd, const void* mask, int Hh, int Ww, float* out, void* work, hipStream_t stream);
int utx_launch_chart_flood(const int* adj, const int* bucket, int F, int* chart, int* flag, hipStream_t stream);
int utx_launch_to_u8(const float* src, long n_rows, long row_elems, int flip, void* dst, hipStream_t stream);
}
