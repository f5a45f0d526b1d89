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
    float scale_log2;  // softmax_scale * log2(e)
};
typedef utx_gemm_desc GemmParams;
typedef utx_gemv_desc GemvParams;
typedef utx_qkv_post_desc QkvPostParams;
typedef utx_ln_mod_desc LnModParams;
typedef utx_sched_desc SchedParams;

extern "C" {
int utx_launch_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                        long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                        long o_ss, int H, int S, float scale, hipStream_t stream);
int utx_launch_gemm_bf16(const GemmParams* p, hipStream_t stream);
int utx_launch_gemv_bf16(const GemvParams* p, hipStream_t stream);
int utx_launch_qkv_post(const QkvPostParams* p, hipStream_t stream);
int utx_launch_ln_mod(const LnModParams* p, hipStream_t stream);
int utx_launch_sched_step(const SchedParams* p, hipStream_t stream);
}
