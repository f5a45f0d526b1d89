// C-ABI entry points of libunitex_hip.so (declared in include/unitex_hip.h).
// Thin: argument validation, error bookkeeping, launch.  No torch, no hidden syncs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <stdlib.h>
#include <mutex>
#include <map>
#include <vector>
#include <utility>
#include "kernels.h"
#include "common.h"

UtxOptions g_utx_opt = {1, 2, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0};      // (the last two: gemm_fastk, nn_grid)

struct OptName { const char* name; int UtxOptions::*field; bool ablation; };
static const OptName kOptions[] = {
    {"UTX_ATTN_GLDS", &UtxOptions::attn_glds, false},       {"UTX_ATTN_FAST", &UtxOptions::attn_fast, false},
    {"UTX_ATTN_Q64", &UtxOptions::attn_q64, false},         {"UTX_ATTN_TPB", &UtxOptions::attn_tpb, false},
    {"UTX_ATTN_TAILSPLIT", &UtxOptions::attn_tailsplit, false},
    {"UTX_GEMM_GROUP_M", &UtxOptions::gemm_group_m, false}, {"UTX_GEMM_TILE", &UtxOptions::gemm_tile, false},
    {"UTX_GEMM_TAILSPLIT", &UtxOptions::gemm_tailsplit, false}, {"UTX_GEMM_PERS_GRID", &UtxOptions::gemm_pers_grid, false},
    {"UTX_GEMM_PERS_SCHED", &UtxOptions::gemm_pers_sched, false}, {"UTX_GEMM_STREAMK", &UtxOptions::gemm_streamk, false},
    {"UTX_BVH_STACK_WALK", &UtxOptions::bvh_stack_walk, false}, {"UTX_BVH_PACKET", &UtxOptions::bvh_packet, false},
    {"UTX_ATTN_PEEL", &UtxOptions::attn_peel, false},             {"UTX_ATTN8_PEEL", &UtxOptions::attn8_peel, false},
    {"UTX_NN_GRID", &UtxOptions::nn_grid, false},                 {"UTX_GEMM_FASTK", &UtxOptions::gemm_fastk, false},
    {"UTX_ATTN_VAR", &UtxOptions::attn_var_abl, true},      {"UTX_ATTN_DEBUG", &UtxOptions::attn_debug_abl, true},
    {"UTX_GEMM_DEBUG", &UtxOptions::gemm_debug_abl, true},
};
#ifdef UTX_ABLATION
static const bool kAblationBuild = true;
#else
static const bool kAblationBuild = false;
#endif

// The environment is read exactly once, by whichever entry point touches the options first (utx_init, utx_set_option, utx_get_option,
// utx_gemm_plan): a utx_set_option made before the first utx_init must not be overwritten by a later read of the same UTX_* variable.
static void options_from_env_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (const OptName& o : kOptions) {
            if (o.ablation && !kAblationBuild) continue;      // wrong-result switches do not exist in the product library
            const char* e = getenv(o.name);
            if (e && *e) g_utx_opt.*(o.field) = atoi(e);
        }
    });
}

struct utx_ctx {
    int device;
    std::string err;
    // scratch of the attention tail split for the entry points that take no workspace argument: owned by THIS context (not process-static), one buffer PER
    // STREAM (two streams of a context may run attention side by side) and GROW-ONLY: a buffer that was handed to a launch is never freed before utx_free
    // -- a replaced (smaller) buffer is parked in `retired`, so nothing that still holds its pointer (a launch in flight, a captured graph) can be left
    // dangling.  A CAPTURING stream gets no context scratch at all (the launch stays unsplit): a graph must not bake in a pointer whose other users
    // this library cannot order against the replays.  Callers that want the split under capture, or no allocation on the launch path, pass their own
    // buffer through utx_attn_fwd_bf16_ws (what FluxDiT does).
    std::mutex ws_mu;
    std::map<hipStream_t, std::pair<void*, size_t>> attn_ws;
    std::vector<void*> retired;
};

static void* ctx_attn_ws(utx_ctx* ctx, int H, int Sq, int S, hipStream_t stream, size_t* bytes) {
    *bytes = 0;
    if (!ctx) return nullptr;
    const size_t need = utx_attn_workspace_bytes_impl(H, Sq == S ? 0 : Sq, S, utx_ncu());
    if (need == 0) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(ctx->ws_mu);
    std::pair<void*, size_t>& slot = ctx->attn_ws[stream];
    if (slot.second < need) {
        void* nb = nullptr;
        if (hipMalloc(&nb, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }      // no scratch: the launch stays unsplit
        if (slot.first) ctx->retired.push_back(slot.first);
        slot = {nb, need};
    }
    *bytes = slot.second;
    return slot.first;
}

static int fail(utx_ctx* ctx, int code, const char* what) {
    if (ctx) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s failed (code %d%s)", what, code,
                 code == -2 ? ": invalid argument / alignment" :
                 code == -3 ? ": hipFuncSetAttribute" :
                 code == -4 ? ": launch error" : "");
        ctx->err = buf;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ctx->err += " hip: "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}
#define UTX_CALL(ctx, name, expr) do { int rc_ = (expr); if (rc_ != 0) return fail(ctx, rc_, name); return 0; } while (0)

extern "C" {

int utx_version(void) { return UTX_VERSION; }

int utx_init(int device, utx_ctx** out) {
    if (!out) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -5;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -5;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return -6;  // MI355X only, by design
    options_from_env_once();
    utx_ctx* c = new utx_ctx();
    c->device = device;
    *out = c;
    return 0;
}

int utx_set_option(const char* name, int value) {
    if (!name) return -2;
    options_from_env_once();
    for (const OptName& o : kOptions)
        if (strcmp(name, o.name) == 0) {
            if (o.ablation && !kAblationBuild) return -7;
            g_utx_opt.*(o.field) = value;
            return 0;
        }
    return -2;
}

int utx_get_option(const char* name, int* value) {
    if (!name || !value) return -2;
    options_from_env_once();
    for (const OptName& o : kOptions)
        if (strcmp(name, o.name) == 0) {
            if (o.ablation && !kAblationBuild) return -7;
            *value = g_utx_opt.*(o.field);
            return 0;
        }
    return -2;
}

int utx_is_ablation_build(void) { return kAblationBuild ? 1 : 0; }

void utx_free(utx_ctx* ctx) {
    if (ctx) {
        for (auto& kv : ctx->attn_ws) if (kv.second.first) (void)hipFree(kv.second.first);
        for (void* p : ctx->retired) (void)hipFree(p);
    }
    delete ctx;
}

const char* utx_last_error(utx_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int utx_attn_fwd_bf16(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                      long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                      int H, int S, float softmax_scale, utx_stream stream) {
    if (!q || !k || !vt || !o) return fail(ctx, -2, "utx_attn_fwd_bf16");
    size_t wsb = 0; void* const ws = ctx_attn_ws(ctx, H, S, S, (hipStream_t)stream, &wsb);
    UTX_CALL(ctx, "utx_attn_fwd_bf16",
             utx_launch_attn_fwd(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S, S,
                                 softmax_scale, 0.f, 0, ws, wsb, (hipStream_t)stream));
}

int utx_attn_fwd_bf16_kb(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                         long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                         int H, int S, float softmax_scale, float key_bias_log2, int key_bias_period, utx_stream stream) {
    if (!q || !k || !vt || !o) return fail(ctx, -2, "utx_attn_fwd_bf16_kb");
    size_t wsb = 0; void* const ws = ctx_attn_ws(ctx, H, S, S, (hipStream_t)stream, &wsb);
    UTX_CALL(ctx, "utx_attn_fwd_bf16_kb",
             utx_launch_attn_fwd(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S, S,
                                 softmax_scale, key_bias_log2, key_bias_period, ws, wsb, (hipStream_t)stream));
}

int utx_attn_fwd_bf16_kbq(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                          long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                          int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period, utx_stream stream) {
    if (!q || !k || !vt || !o) return fail(ctx, -2, "utx_attn_fwd_bf16_kbq");
    size_t wsb = 0; void* const ws = (S_q > 0) ? ctx_attn_ws(ctx, H, S_q, S_kv, (hipStream_t)stream, &wsb) : nullptr;
    UTX_CALL(ctx, "utx_attn_fwd_bf16_kbq",
             utx_launch_attn_fwd(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S_kv, S_q,
                                 softmax_scale, key_bias_log2, key_bias_period, ws, wsb, (hipStream_t)stream));
}

int utx_attn_fwd_bf16_ws(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                         long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                         int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period,
                         void* work, size_t work_bytes, utx_stream stream) {
    if (!q || !k || !vt || !o) return fail(ctx, -2, "utx_attn_fwd_bf16_ws");
    UTX_CALL(ctx, "utx_attn_fwd_bf16_ws",
             utx_launch_attn_fwd(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S_kv, S_q,
                                 softmax_scale, key_bias_log2, key_bias_period, work, work_bytes, (hipStream_t)stream));
}

int utx_attn_fwd_bf16_blk(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                          long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                          int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period,
                          void* work, size_t work_bytes, int blk_rows, long q_bs, long k_bs, long vt_bs, utx_stream stream) {
    if (!q || !k || !vt || !o || blk_rows <= 0) return fail(ctx, -2, "utx_attn_fwd_bf16_blk");
    UTX_CALL(ctx, "utx_attn_fwd_bf16_blk",
             utx_launch_attn_fwd_blk(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S_kv, S_q,
                                     softmax_scale, key_bias_log2, key_bias_period, work, work_bytes, blk_rows, q_bs, k_bs, vt_bs, (hipStream_t)stream));
}

int utx_quant_vt_mx8(utx_ctx* ctx, const void* vt, void* v8, void* vs, int H, int S_pad, utx_stream stream) {
    if (!vt || !v8 || !vs) return fail(ctx, -2, "utx_quant_vt_mx8");
    UTX_CALL(ctx, "utx_quant_vt_mx8", utx_launch_quant_vt_mx8(vt, v8, vs, H, S_pad, (hipStream_t)stream));
}

int utx_attn_fwd_fp8(utx_ctx* ctx, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                     int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, utx_stream stream) {
    if (!q8 || !qs || !k8 || !ks || !v8t || !vs || !o) return fail(ctx, -2, "utx_attn_fwd_fp8");
    Attn8Params p;
    p.q8 = (const uint8_t*)q8; p.k8 = (const uint8_t*)k8; p.v8t = (const uint8_t*)v8t;
    p.qs = (const uint32_t*)qs; p.ks = (const uint32_t*)ks; p.vs = (const uint32_t*)vs;
    p.o = (bf16_t*)o; p.o_ss = o_ss; p.H = H; p.S = S_kv; p.Sq = S_q; p.S_pad = S_pad; p.nqb = 0;
    p.key_bias_log2 = key_bias_log2; p.key_bias_period = key_bias_period;
    // the key-split tail round runs on the context's per-stream scratch (none while the stream is capturing: the launch then stays unsplit); utx_attn_fwd_fp8_ws takes the caller's
    p.work = (S_q > 0 && S_q <= S_kv) ? ctx_attn_ws(ctx, H, S_q, S_kv, (hipStream_t)stream, &p.work_bytes) : nullptr;
    if (!p.work) p.work_bytes = 0;
    UTX_CALL(ctx, "utx_attn_fwd_fp8", utx_launch_attn_fwd_fp8(&p, (hipStream_t)stream));
}

int utx_attn_fwd_fp8_ws(utx_ctx* ctx, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                        int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes, utx_stream stream) {
    if (!q8 || !qs || !k8 || !ks || !v8t || !vs || !o) return fail(ctx, -2, "utx_attn_fwd_fp8_ws");
    Attn8Params p;
    p.q8 = (const uint8_t*)q8; p.k8 = (const uint8_t*)k8; p.v8t = (const uint8_t*)v8t;
    p.qs = (const uint32_t*)qs; p.ks = (const uint32_t*)ks; p.vs = (const uint32_t*)vs;
    p.o = (bf16_t*)o; p.o_ss = o_ss; p.H = H; p.S = S_kv; p.Sq = S_q; p.S_pad = S_pad; p.nqb = 0;
    p.key_bias_log2 = key_bias_log2; p.key_bias_period = key_bias_period;
    p.work = work; p.work_bytes = work ? work_bytes : 0;
    UTX_CALL(ctx, "utx_attn_fwd_fp8_ws", utx_launch_attn_fwd_fp8(&p, (hipStream_t)stream));
}

size_t utx_attn_workspace_bytes(utx_ctx* ctx, int H, int S_q, int S_kv) {
    (void)ctx;
    // S_q > S_kv is a plain case since round 6 (the sequence-parallel launch over de-duplicated text keys); refusing it here left exactly that launch without scratch -- unsplit
    // and on the 8 x 32 kernel -- while the launcher itself accepted it
    if (H <= 0 || S_kv <= 0 || S_q <= 0) return 0;
    return utx_attn_workspace_bytes_impl(H, S_q == S_kv ? 0 : S_q, S_kv, utx_ncu());
}

int utx_attn_plan(int H, int S_q, int S_kv, int n_cus, int out[4]) {
    if (!out || H <= 0 || S_kv <= 0 || S_q <= 0 || n_cus <= 0) return -2;
    options_from_env_once();
    utx_attn_split_plan_impl(H, S_q == S_kv ? 0 : S_q, S_kv, n_cus, out);
    return 0;
}

size_t utx_gemm_streamk_workspace_bytes(utx_ctx*) { return utx_gemm_streamk_workspace_bytes_impl(); }
int utx_gemm_plan(const utx_gemm_desc* d, int n_cus, int out[4]) {
    if (!d || !out || n_cus <= 0 || d->M <= 0 || d->N <= 0 || d->K <= 0) return -2;
    options_from_env_once();
    utx_gemm_plan_impl(d, n_cus, d->sk_work != nullptr, out);
    return 0;
}
int utx_gemm_bf16(utx_ctx* ctx, const utx_gemm_desc* d, utx_stream stream) {
    if (!d || !d->A || !d->B || !d->C) return fail(ctx, -2, "utx_gemm_bf16");
    UTX_CALL(ctx, "utx_gemm_bf16", utx_launch_gemm_bf16(d, (hipStream_t)stream));
}

int utx_quant_mx8(utx_ctx* ctx, const void* x, long ldx, void* q, long ldq, void* sc, long lds, int M, int K, utx_stream stream) {
    if (!x || !q || !sc) return fail(ctx, -2, "utx_quant_mx8");
    UTX_CALL(ctx, "utx_quant_mx8", utx_launch_quant_mx8(x, ldx, q, ldq, sc, lds, M, K, (hipStream_t)stream));
}

int utx_quant_mx8_packed(utx_ctx* ctx, const void* x, long ldx, void* q, long ldq, void* sc, long row_blocks, int M, int K, utx_stream stream) {
    if (!x || !q || !sc) return fail(ctx, -2, "utx_quant_mx8_packed");
    UTX_CALL(ctx, "utx_quant_mx8_packed", utx_launch_quant_mx8_packed(x, ldx, q, ldq, sc, row_blocks, M, K, (hipStream_t)stream));
}

int utx_gemv_bf16(utx_ctx* ctx, const utx_gemv_desc* d, utx_stream stream) {
    if (!d || !d->x || !d->W || !d->y) return fail(ctx, -2, "utx_gemv_bf16");
    UTX_CALL(ctx, "utx_gemv_bf16", utx_launch_gemv_bf16(d, (hipStream_t)stream));
}

size_t utx_group_norm_workspace_bytes(void) { return utx_group_norm_workspace_bytes_impl(); }

int utx_group_norm(utx_ctx* ctx, const void* x, long npix, int C, const void* gamma, const void* beta, float eps, int silu,
                   void* y, void* work, utx_stream stream) {
    if (!x || !gamma || !beta || !y || !work) return fail(ctx, -2, "utx_group_norm");
    UTX_CALL(ctx, "utx_group_norm", utx_launch_group_norm(x, npix, C, gamma, beta, eps, silu, y, work, (hipStream_t)stream));
}

int utx_softmax_rows(utx_ctx* ctx, void* s, long nrow, long ld, int ncol, utx_stream stream) {
    if (!s) return fail(ctx, -2, "utx_softmax_rows");
    UTX_CALL(ctx, "utx_softmax_rows", utx_launch_softmax_rows(s, nrow, ld, ncol, (hipStream_t)stream));
}

int utx_conv3x3_thin(utx_ctx* ctx, const void* x, int H, int W, int Cin, const void* wt, const void* bias, int Cout, void* y,
                     utx_stream stream) {
    if (!x || !wt || !bias || !y) return fail(ctx, -2, "utx_conv3x3_thin");
    UTX_CALL(ctx, "utx_conv3x3_thin", utx_launch_conv3x3_thin(x, H, W, Cin, wt, bias, Cout, y, (hipStream_t)stream));
}

int utx_qkv_post(utx_ctx* ctx, const utx_qkv_post_desc* d, utx_stream stream) {
    if (!d || !d->qkv || !d->Qh || !d->Kh || !d->Vt || !d->cosb || !d->sinb || !d->wq || !d->wk)
        return fail(ctx, -2, "utx_qkv_post");
    UTX_CALL(ctx, "utx_qkv_post", utx_launch_qkv_post(d, (hipStream_t)stream));
}

int utx_sp_unpack_qkv(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* q, void* k, void* vt, utx_stream stream) {
    if (!recv || !q || !k || !vt) return fail(ctx, -2, "utx_sp_unpack_qkv");
    UTX_CALL(ctx, "utx_sp_unpack_qkv", utx_launch_sp_unpack_qkv(recv, P, Hp, S_loc, 0, q, k, vt, (hipStream_t)stream));
}

int utx_sp_unpack_qkv_dedup(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, int text_rows, void* q, void* k, void* vt, utx_stream stream) {
    if (!recv || !q || !k || !vt) return fail(ctx, -2, "utx_sp_unpack_qkv_dedup");
    UTX_CALL(ctx, "utx_sp_unpack_qkv_dedup", utx_launch_sp_unpack_qkv(recv, P, Hp, S_loc, text_rows, q, k, vt, (hipStream_t)stream));
}

int utx_sp_unpack_o(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* out, long ld, utx_stream stream) {
    if (!recv || !out) return fail(ctx, -2, "utx_sp_unpack_o");
    UTX_CALL(ctx, "utx_sp_unpack_o", utx_launch_sp_unpack_o(recv, P, Hp, S_loc, out, ld, 0, (hipStream_t)stream));
}

int utx_sp_unpack_o_cols(utx_ctx* ctx, const void* recv, int P, int Hp, int S_loc, void* out, long ld, long src_cols, utx_stream stream) {
    if (!recv || !out || src_cols <= 0) return fail(ctx, -2, "utx_sp_unpack_o_cols");
    UTX_CALL(ctx, "utx_sp_unpack_o_cols", utx_launch_sp_unpack_o(recv, P, Hp, S_loc, out, ld, src_cols, (hipStream_t)stream));
}

int utx_ln_mod(utx_ctx* ctx, const utx_ln_mod_desc* d, utx_stream stream) {
    if (!d || !d->x || (!d->y && !d->q) || !d->shift || !d->scale) return fail(ctx, -2, "utx_ln_mod");
    UTX_CALL(ctx, "utx_ln_mod", utx_launch_ln_mod(d, (hipStream_t)stream));
}

int utx_sched_step(utx_ctx* ctx, const utx_sched_desc* d, utx_stream stream) {
    if (!d || !d->x || !d->v) return fail(ctx, -2, "utx_sched_step");
    UTX_CALL(ctx, "utx_sched_step", utx_launch_sched_step(d, (hipStream_t)stream));
}


// ---- geometry ----
int utx_transform_points(utx_ctx* ctx, const float* verts, int V, const float* mvp, int n_views, float* clip, float* ndc, utx_stream stream) {
    if (!verts || !mvp || !clip) return fail(ctx, -2, "utx_transform_points");
    UTX_CALL(ctx, "utx_transform_points", utx_launch_transform(verts, V, mvp, n_views, clip, ndc, (hipStream_t)stream));
}
long utx_rasterize_workspace_bytes(int F, int H, int W) { return 8L * H * W + 16 + 4L * F + 64; }
int utx_rasterize(utx_ctx* ctx, const float* pos, const int* tri, int F, int H, int W, float* rast, void* work, utx_stream stream) {
    if (!pos || !tri || !rast || !work) return fail(ctx, -2, "utx_rasterize");
    UTX_CALL(ctx, "utx_rasterize", utx_launch_rasterize(pos, tri, F, H, W, rast, work, (hipStream_t)stream));
}
int utx_interpolate(utx_ctx* ctx, const float* attr, int C, const float* rast, const int* tri, long npix, float* out, utx_stream stream) {
    if (!attr || !rast || !tri || !out) return fail(ctx, -2, "utx_interpolate");
    UTX_CALL(ctx, "utx_interpolate", utx_launch_interpolate(attr, C, rast, tri, npix, out, (hipStream_t)stream));
}
int utx_face_normals(utx_ctx* ctx, const float* verts, const int* faces, int F, float* out, utx_stream stream) {
    if (!verts || !faces || !out) return fail(ctx, -2, "utx_face_normals");
    UTX_CALL(ctx, "utx_face_normals", utx_launch_face_normals(verts, faces, F, out, (hipStream_t)stream));
}

int utx_texture_shade(utx_ctx* ctx, const float* rast, const float* uv, const int* tri, const float* tex, int Ht, int Wt,
                      const float* bg3_host, long npix, void* out, utx_stream stream) {
    if (!rast || !uv || !tri || !tex || !out) return fail(ctx, -2, "utx_texture_shade");
    UTX_CALL(ctx, "utx_texture_shade", utx_launch_texture_shade(rast, uv, tri, tex, Ht, Wt, bg3_host, npix, out, (hipStream_t)stream));
}

int utx_condition_shade(utx_ctx* ctx, const float* rast, const float* nrm, const float* pos, const float* bg3_host, long npix,
                        void* out_normal, void* out_ccm, void* out_alpha, utx_stream stream) {
    if (!rast || !nrm || !pos || !bg3_host || !out_normal || !out_ccm || !out_alpha) return fail(ctx, -2, "utx_condition_shade");
    UTX_CALL(ctx, "utx_condition_shade", utx_launch_condition_shade(rast, nrm, pos, bg3_host, npix, out_normal, out_ccm, out_alpha, (hipStream_t)stream));
}
int utx_bvh_build(utx_ctx* ctx, const float* verts, int V, const int* faces, int F, utx_bvh** out, utx_stream stream) {
    if (!verts || !faces || !out) return fail(ctx, -2, "utx_bvh_build");
    UTX_CALL(ctx, "utx_bvh_build", utx_bvh_build_impl(verts, V, faces, F, out, (hipStream_t)stream));
}
size_t utx_bvh_workspace_bytes(int F) { return utx_bvh_workspace_bytes_impl(F); }
int utx_bvh_build_ws(utx_ctx* ctx, const float* verts, int V, const int* faces, int F, void* work, size_t work_bytes, utx_bvh** out, utx_stream stream) {
    if (!verts || !faces || !out || !work) return fail(ctx, -2, "utx_bvh_build_ws");
    UTX_CALL(ctx, "utx_bvh_build_ws", utx_bvh_build_ws_impl(verts, V, faces, F, work, work_bytes, out, (hipStream_t)stream));
}
void utx_bvh_free(utx_bvh* bvh) { utx_bvh_free_impl(bvh); }
int utx_bvh_arrays(utx_bvh* bvh, int** info, float** aabb, unsigned** codes_sorted, int** idx_sorted) {
    return utx_bvh_arrays_impl(bvh, info, aabb, codes_sorted, idx_sorted);
}
int utx_bvh_trace(utx_ctx* ctx, utx_bvh* bvh, const float* ro, const float* rd, long R, int* tid, utx_stream stream) {
    if (!bvh || !ro || !rd || !tid) return fail(ctx, -2, "utx_bvh_trace");
    UTX_CALL(ctx, "utx_bvh_trace", utx_bvh_trace_impl(bvh, ro, rd, R, tid, nullptr, g_utx_opt.bvh_stack_walk, (hipStream_t)stream));
}
int utx_bvh_trace_count(utx_ctx* ctx, utx_bvh* bvh, const float* ro, const float* rd, long R, int* tid, unsigned long long* visited, utx_stream stream) {
    if (!bvh || !ro || !rd || !tid || !visited) return fail(ctx, -2, "utx_bvh_trace_count");
    UTX_CALL(ctx, "utx_bvh_trace_count", utx_bvh_trace_impl(bvh, ro, rd, R, tid, visited, 0, (hipStream_t)stream));
}
int utx_bvh_depth(utx_bvh* bvh) { return bvh ? utx_bvh_depth_impl(bvh) : -2; }
int utx_backproject(utx_ctx* ctx, const utx_backproject_desc* d, utx_bvh* bvh, utx_stream stream) {
    if (!d || !bvh || !d->rast2d || !d->verts || !d->faces || !d->fnormal || !d->vndc || !d->dirs || !d->images ||
        !d->color || !d->rayvis || !d->alphaok || d->view_begin < 0 || d->view_begin + d->view_count > d->n_views)
        return fail(ctx, -2, "utx_backproject");
    UTX_CALL(ctx, "utx_backproject", utx_launch_backproject(d, bvh, (hipStream_t)stream));
}
int utx_dilate_visibility(utx_ctx* ctx, const void* rayvis, const void* alphaok, const float* rast2d, int n_views, int H, int W,
                          void* tmp, void* vis_out, utx_stream stream) {
    if (!rayvis || !alphaok || !rast2d || !tmp || !vis_out) return fail(ctx, -2, "utx_dilate_visibility");
    UTX_CALL(ctx, "utx_dilate_visibility", utx_launch_dilate_visibility(rayvis, alphaok, rast2d, n_views, H, W, tmp, vis_out, (hipStream_t)stream));
}
int utx_composite(utx_ctx* ctx, const float* colors, const void* vis, const int* order_host, int n_order, long T, float* atlas,
                  void* winner, utx_stream stream) {
    if (!colors || !vis || !order_host || !atlas || !winner) return fail(ctx, -2, "utx_composite");
    UTX_CALL(ctx, "utx_composite", utx_launch_composite(colors, vis, order_host, n_order, T, atlas, winner, (hipStream_t)stream));
}
int utx_seam_mask(utx_ctx* ctx, const void* winner, const float* rast2d, int H, int W, void* tmp, void* seam, utx_stream stream) {
    if (!winner || !rast2d || !tmp || !seam) return fail(ctx, -2, "utx_seam_mask");
    UTX_CALL(ctx, "utx_seam_mask", utx_launch_seam_mask(winner, rast2d, H, W, tmp, seam, (hipStream_t)stream));
}
int utx_view_visibility(utx_ctx* ctx, const float* attr6, const float* rast, const float* fnormal, const float* dirs, int n, int H, int W,
                        float grad_thr, float cos_thr, int radius, void* tmp, void* vis, float* alpha, utx_stream stream) {
    if (!attr6 || !rast || !fnormal || !dirs || !tmp || !vis) return fail(ctx, -2, "utx_view_visibility");
    UTX_CALL(ctx, "utx_view_visibility", utx_launch_view_visibility(attr6, rast, fnormal, dirs, n, H, W, grad_thr, cos_thr, radius, tmp, vis, alpha, (hipStream_t)stream));
}
long utx_knn_workspace_bytes(long N) { return (long)utx_knn_workspace_bytes_impl(N); }
int utx_knn(utx_ctx* ctx, const utx_knn_desc* d, void* work, long work_bytes, utx_stream stream) {
    if (!d || !d->src_pos || !d->dst_pos || !work) return fail(ctx, -2, "utx_knn");
    UTX_CALL(ctx, "utx_knn", utx_launch_knn(d, work, (size_t)work_bytes, (hipStream_t)stream));
}
long utx_nn_fill_workspace_bytes(long T) { return (long)utx_nn_fill_workspace_bytes_impl(T); }
int utx_nn_fill(utx_ctx* ctx, const float* pos, const void* winner, const float* rast2d, long T, float* atlas, int* nn_index,
                void* work, long work_bytes, utx_stream stream) {
    if (!pos || !winner || !rast2d || !atlas || !work) return fail(ctx, -2, "utx_nn_fill");
    UTX_CALL(ctx, "utx_nn_fill", utx_launch_nn_fill(pos, winner, rast2d, T, atlas, nn_index, work, (size_t)work_bytes, (hipStream_t)stream));
}
int utx_lens_blur_seam(utx_ctx* ctx, const float* src, const void* seam, int H, int W, const float* k49_host, float* dst, utx_stream stream) {
    if (!src || !seam || !k49_host || !dst) return fail(ctx, -2, "utx_lens_blur_seam");
    UTX_CALL(ctx, "utx_lens_blur_seam", utx_launch_lens_blur_seam(src, seam, H, W, k49_host, dst, (hipStream_t)stream));
}
long utx_pull_push_workspace_bytes(int H, int W) { return (long)utx_pull_push_workspace_bytes_impl(H, W); }
int utx_pull_push(utx_ctx* ctx, const float* kd, const void* mask, int H, int W, float* out, void* work, utx_stream stream) {
    if (!kd || !mask || !out || !work) return fail(ctx, -2, "utx_pull_push");
    UTX_CALL(ctx, "utx_pull_push", utx_launch_pull_push(kd, mask, H, W, out, work, (hipStream_t)stream));
}
int utx_chart_flood(utx_ctx* ctx, const int* adj, const int* bucket, int F, int* chart, int* flag, utx_stream stream) {
    if (!adj || !bucket || !chart || !flag) return fail(ctx, -2, "utx_chart_flood");
    const int rc = utx_launch_chart_flood(adj, bucket, F, chart, flag, (hipStream_t)stream);
    if (rc < 0) return fail(ctx, rc, "utx_chart_flood");
    return rc;
}
int utx_to_u8(utx_ctx* ctx, const float* src, long n_rows, long row_elems, int flip, void* dst, utx_stream stream) {
    if (!src || !dst) return fail(ctx, -2, "utx_to_u8");
    UTX_CALL(ctx, "utx_to_u8", utx_launch_to_u8(src, n_rows, row_elems, flip, dst, (hipStream_t)stream));
}

}  // extern "C"

// Layout self-description, so the ctypes mirror in unitex_amd/_lib.py can be checked without a GPU.
extern "C" int utx_abi_sizes(int* out, int n) {
    const int v[] = {(int)sizeof(utx_gemm_desc), (int)sizeof(utx_gemv_desc), (int)sizeof(utx_qkv_post_desc),
                     (int)sizeof(utx_ln_mod_desc), (int)sizeof(utx_sched_desc), (int)sizeof(utx_backproject_desc),
                     (int)sizeof(utx_knn_desc), (int)sizeof(utx_dit_linear), (int)sizeof(utx_dit_double_block), (int)sizeof(utx_dit_single_block),
                     (int)sizeof(utx_dit_weights), (int)sizeof(utx_dit_config), (int)sizeof(utx_dit_workspace)};
    const int m = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < m && i < n; ++i) out[i] = v[i];
    return m;
}
