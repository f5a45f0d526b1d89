// C-ABI entry points of libunitex_hip.so (declared in include/unitex_hip.h).
// Thin: argument validation, error bookkeeping, launch.  No torch, no hidden syncs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "kernels.h"

struct utx_ctx {
    int device;
    std::string err;
};

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
    utx_ctx* c = new utx_ctx();
    c->device = device;
    *out = c;
    return 0;
}

void utx_free(utx_ctx* ctx) { delete ctx; }

const char* utx_last_error(utx_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int utx_attn_fwd_bf16(utx_ctx* ctx, const void* q, const void* k, const void* vt, void* o,
                      long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds, long o_ss,
                      int H, int S, float softmax_scale, utx_stream stream) {
    if (!q || !k || !vt || !o) return fail(ctx, -2, "utx_attn_fwd_bf16");
    UTX_CALL(ctx, "utx_attn_fwd_bf16",
             utx_launch_attn_fwd(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S,
                                 softmax_scale, (hipStream_t)stream));
}

int utx_gemm_bf16(utx_ctx* ctx, const utx_gemm_desc* d, utx_stream stream) {
    if (!d || !d->A || !d->B || !d->C) return fail(ctx, -2, "utx_gemm_bf16");
    UTX_CALL(ctx, "utx_gemm_bf16", utx_launch_gemm_bf16(d, (hipStream_t)stream));
}

int utx_gemv_bf16(utx_ctx* ctx, const utx_gemv_desc* d, utx_stream stream) {
    if (!d || !d->x || !d->W || !d->y) return fail(ctx, -2, "utx_gemv_bf16");
    UTX_CALL(ctx, "utx_gemv_bf16", utx_launch_gemv_bf16(d, (hipStream_t)stream));
}

int utx_qkv_post(utx_ctx* ctx, const utx_qkv_post_desc* d, utx_stream stream) {
    if (!d || !d->qkv || !d->Qh || !d->Kh || !d->Vt || !d->cosb || !d->sinb || !d->wq || !d->wk)
        return fail(ctx, -2, "utx_qkv_post");
    UTX_CALL(ctx, "utx_qkv_post", utx_launch_qkv_post(d, (hipStream_t)stream));
}

int utx_ln_mod(utx_ctx* ctx, const utx_ln_mod_desc* d, utx_stream stream) {
    if (!d || !d->x || !d->y || !d->shift || !d->scale) return fail(ctx, -2, "utx_ln_mod");
    UTX_CALL(ctx, "utx_ln_mod", utx_launch_ln_mod(d, (hipStream_t)stream));
}

int utx_sched_step(utx_ctx* ctx, const utx_sched_desc* d, utx_stream stream) {
    if (!d || !d->x || !d->v) return fail(ctx, -2, "utx_sched_step");
    UTX_CALL(ctx, "utx_sched_step", utx_launch_sched_step(d, (hipStream_t)stream));
}

}  // extern "C"

// Layout self-description, so the ctypes mirror in unitex_amd/_lib.py can be checked without a GPU.
extern "C" int utx_abi_sizes(int* out, int n) {
    const int v[] = {(int)sizeof(utx_gemm_desc), (int)sizeof(utx_gemv_desc), (int)sizeof(utx_qkv_post_desc),
                     (int)sizeof(utx_ln_mod_desc), (int)sizeof(utx_sched_desc)};
    const int m = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < m && i < n; ++i) out[i] = v[i];
    return m;
}
