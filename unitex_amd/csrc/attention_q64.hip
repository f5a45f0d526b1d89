// bf16 MFMA flash-attention forward, 64 queries per wave (gfx950).  Same contract, maths and LDS-DMA staging as
// attention_glds.hip; what differs is the work split: 4 waves x 64 queries per workgroup instead of 8 x 32, so every K / Vt
// fragment read from LDS feeds TWO MFMAs (one per 32-query half) -- the per-MFMA ds_read_b128 traffic is halved.
// Why: with half of the fragment reads removed (ablation UTX_ATTN_VAR=3, wrong results by design) the 32-query kernel runs
// 25.9 -> 19.9 ms per launch at S = 50 688 (profiles/r01_perf_attn_ablation.log): the LDS fragment stream, not the MFMA
// pipe, is what the waves wait on.  One wave per SIMD with the full 512-entry register file (accumulators for both halves:
// 2 x 64 output + 4 x 16 score registers per lane, Q fragments for both halves).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define AQ_KVB 64
#define AQ_KTILE 16384
#define AQ_VTILE 16384
#define AQ_LDS (2 * (AQ_KTILE + AQ_VTILE))

__device__ __forceinline__ void aq_glds16(const bf16_t* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// Output accumulators, running-max blocks and nothing else live in AGPRs for the whole loop.  The two rare operations that
// touch them with the VALU (rescale by alpha, re-splat of -m) go through opaque asm so that the register allocator never
// sees a VALU use and keeps them in place (a visible VALU use makes it copy all 128 accumulators through VGPRs every tile).
// MFMA -> VALU and VALU -> MFMA hazards around the asm are covered by explicit s_nop (the hazard recogniser does not look
// inside inline asm); this is cold code.
__device__ __forceinline__ void aq_scale_acc(f32x16& a, float alpha) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = a[r], t;
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x), "=&v"(t) : "v"(alpha));
        a[r] = x;
    }
}
__device__ __forceinline__ void aq_splat_acc(f32x16& a, float v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(x) : "v"(v));
        a[r] = x;
    }
}

// VAR: timing ablations (wrong results by design): 1 = no v_exp, 2 = no softmax VALU at all, 3 = block-1 fragments reuse the
// block-0 registers (half the ds_reads), 4 = no DMA staging after the first tile, 5 = fragments read once (no ds_reads in
// steady state), 6 = 5 + 4, 7 = 6 + no barrier, 8 = no barrier only.
template <int PRESC, int VAR>
__global__ __launch_bounds__(256) void attn_fwd_q64_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kring = smem;
    char* const vring = smem + 2 * AQ_KTILE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;

    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int head = w / p.nqb;
    const int qb = w - head * p.nqb;
    const int S = p.S;
    const bf16_t* kbase = p.k + (long)head * p.k_hs;
    const bf16_t* vbase = p.vt + (long)head * p.vt_hs;

    // ---- Q fragments of both 32-query halves
    const int q0 = qb * 256 + wave * 64;
    bf16x8 qf0[8], qf1[8];
    {
        int r0 = q0 + lq, r1 = q0 + 32 + lq;
        if (r0 > S - 1) r0 = S - 1;
        if (r1 > S - 1) r1 = S - 1;
        const bf16_t* qp0 = p.q + (long)head * p.q_hs + (long)r0 * p.q_ss + lh * 8;
        const bf16_t* qp1 = p.q + (long)head * p.q_hs + (long)r1 * p.q_ss + lh * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            qf0[kk] = *reinterpret_cast<const bf16x8*>(qp0 + kk * 16);
            qf1[kk] = *reinterpret_cast<const bf16x8*>(qp1 + kk * 16);
        }
    }

    // ---- DMA sources (4 waves: wave-instruction (wave, j), j = 0..3, fills LDS bytes [(4*wave + j) * 1024, +1024))
    //   K : slot s = (4*wave+j)*64 + lane -> row s>>4, LDS chunk s&15 <- global chunk (s&15) ^ (row&15)
    //   Vt: slot s                        -> row s>>3, LDS chunk s&7  <- global chunk (s&7) ^ ((row>>1)&7)
#define AQ_SRC(j_)                                                                                          \
    const int ks##j_ = (4 * wave + (j_)) * 64 + lane;                                                       \
    const bf16_t* ksrc##j_ = kbase + (long)(ks##j_ >> 4) * p.k_ss + (((ks##j_ & 15) ^ ((ks##j_ >> 4) & 15)) << 3); \
    const bf16_t* vsrc##j_ = vbase + (long)(ks##j_ >> 3) * p.vt_ds + (((ks##j_ & 7) ^ ((ks##j_ >> 4) & 7)) << 3);
    AQ_SRC(0) AQ_SRC(1) AQ_SRC(2) AQ_SRC(3)
    const int dma_off = (4 * wave) * 1024;
#define AQ_STAGE1(j_, slot_)                                                                               \
    aq_glds16(ksrc##j_ + kadv_, kring + (slot_) * AQ_KTILE + dma_off + (j_) * 1024);                       \
    aq_glds16(vsrc##j_ + vadv_, vring + (slot_) * AQ_VTILE + dma_off + (j_) * 1024);
#define AQ_STAGE(t_, slot_)                                                                                \
    do {                                                                                                   \
        const long kadv_ = (long)(t_) * AQ_KVB * p.k_ss;                                                   \
        const int vadv_ = (t_) * AQ_KVB;                                                                   \
        AQ_STAGE1(0, slot_) AQ_STAGE1(1, slot_) AQ_STAGE1(2, slot_) AQ_STAGE1(3, slot_)                    \
    } while (0)

    // ---- fragment read offsets (kappa permutation and swizzles as in attention_glds.hip)
    const int ka = lq >> 3, khp = (lq >> 2) & 1, kc = lq & 3;
    const int krow = 16 * (ka >> 1) + 8 * khp + 4 * (ka & 1) + kc;
    const int kswz = krow & 15, vswz = (lq >> 1) & 7;
    int kx[8], vx[4];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kx[kk] = krow * 256 + (((2 * kk + lh) ^ kswz) << 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) vx[s] = lq * 128 + (((2 * s + lh) ^ vswz) << 4);

    f32x16 oacc0[4], oacc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc0[i][r] = 0.f; oacc1[i][r] = 0.f; }
    float m_run0 = 0.f, l_run0 = 0.f, m_run1 = 0.f, l_run1 = 0.f;
    const float c2 = p.scale_log2;
    bf16x8 pb0[4], pb1[4];
    f32x16 negm0, negm1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm0[r] = 0.f; negm1[r] = 0.f; }

    const int nt = (S + AQ_KVB - 1) / AQ_KVB;
    AQ_STAGE(0, 0);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { asm volatile("" : "+v"(qf0[kk])); asm volatile("" : "+v"(qf1[kk])); }

    bf16x8 kfa[8], kfb[8], vfa[8], vfb[8];
    constexpr bool NOFRAG = (VAR >= 5 && VAR <= 7);
    if (NOFRAG) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            kfa[kk] = *reinterpret_cast<const bf16x8*>(kring + kx[kk]);
            kfb[kk] = *reinterpret_cast<const bf16x8*>(kring + 8192 + kx[kk]);
            vfa[kk] = *reinterpret_cast<const bf16x8*>(vring + (kk & 3) * 4096 + vx[kk >> 2]);
            vfb[kk] = *reinterpret_cast<const bf16x8*>(vring + (kk & 3) * 4096 + vx[2 + (kk >> 2)]);
        }
    }
    for (int t = 0; t < nt; ++t) {
        const int slot = t & 1;
        if (NOFRAG) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) { asm volatile("" : "+v"(kfa[kk]), "+v"(kfb[kk]), "+v"(vfa[kk]), "+v"(vfb[kk])); }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { asm volatile("" : "+v"(qf0[kk])); asm volatile("" : "+v"(qf1[kk])); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+a"(oacc0[i])); asm volatile("" : "+a"(oacc1[i])); }
        asm volatile("" : "+a"(negm0)); asm volatile("" : "+a"(negm1));
        if (t + 1 < nt && ((VAR != 4 && VAR != 6 && VAR != 7) || t == 0)) AQ_STAGE(t + 1, slot ^ 1);
        const char* kb = kring + slot * AQ_KTILE;
        const char* vb = vring + slot * AQ_VTILE;
        const bool ragged = (t == nt - 1) && (S & (AQ_KVB - 1));
        const int lim = S - t * AQ_KVB - 8 * lh;

        f32x16 sa00, sa01, sa10, sa11;     // sa{block}{half}
#define AQ_EXPB(sa_, p0_, p1_, ps_)                                                                  \
        if (VAR != 2) {                                                                              \
            f32x2 acc2_ = {0.f, 0.f};                                                                \
            _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                      \
                f32x2 pv_;                                                                           \
                pv_[0] = VAR == 1 ? sa_[r] : __builtin_amdgcn_exp2f(PRESC ? sa_[r] : sa_[r] * c2);   \
                pv_[1] = VAR == 1 ? sa_[r + 1] : __builtin_amdgcn_exp2f(PRESC ? sa_[r + 1] : sa_[r + 1] * c2); \
                acc2_ += pv_;                                                                        \
                if (r < 8) { p0_[r] = (__bf16)pv_[0]; p0_[r + 1] = (__bf16)pv_[1]; }                 \
                else { p1_[r - 8] = (__bf16)pv_[0]; p1_[r - 7] = (__bf16)pv_[1]; }                   \
            }                                                                                        \
            ps_ = acc2_[0] + acc2_[1];                                                               \
        }
        // slow path for (block, half H): true max, move that half's running max, rescale what exists of that half
#define AQ_SLOW(sa_, other_, fix_other_, kblk_, boff_, first_, H)                                    \
        {                                                                                            \
            _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                       \
                const bf16x8 kf_ = *reinterpret_cast<const bf16x8*>(kb + (kblk_) + kx[kk]);          \
                sa_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf##H[kk], kk == 0 ? negm##H : sa_, 0, 0, 0); \
            }                                                                                        \
            if (ragged) {                                                                            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                       \
                    if ((boff_) + 16 * (r >> 3) + (r & 7) >= lim) sa_[r] = -INFINITY;                \
            }                                                                                        \
            float mx_ = sa_[0];                                                                      \
            _Pragma("unroll") for (int r = 1; r < 16; ++r) mx_ = fmaxf(mx_, sa_[r]);                 \
            mx_ = fmaxf(mx_, __shfl_xor(mx_, 32, 64));                                               \
            const float d_ = (first_) ? mx_ : fmaxf(mx_, 0.f);                                       \
            const float alpha_ = (first_) ? 1.0f : __builtin_amdgcn_exp2f(PRESC ? -d_ : -d_ * c2);   \
            m_run##H += d_;                                                                          \
            l_run##H *= alpha_;                                                                      \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) sa_[r] -= d_;                             \
            if (fix_other_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) other_[r] -= d_; }      \
            asm volatile("s_nop 15\n\ts_nop 7");                                                     \
            aq_splat_acc(negm##H, -m_run##H);                                                        \
            if (!(first_)) { _Pragma("unroll") for (int i = 0; i < 4; ++i) aq_scale_acc(oacc##H[i], alpha_); } \
            asm volatile("s_nop 4");                                                                 \
        }
        float ps00 = 0.f, ps01 = 0.f, ps10 = 0.f, ps11 = 0.f;
        // S0: QK(block 0) for both halves from one set of K fragments; block-1 fragments stream in behind the MFMAs
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) if (!NOFRAG) kfa[kk] = *reinterpret_cast<const bf16x8*>(kb + kx[kk]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (VAR == 3) kfb[kk] = kfa[kk]; else if (!NOFRAG) kfb[kk] = *reinterpret_cast<const bf16x8*>(kb + 8192 + kx[kk]);
            sa00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[kk], qf0[kk], kk == 0 ? negm0 : sa00, 0, 0, 0);
            sa01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[kk], qf1[kk], kk == 0 ? negm1 : sa01, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // S1: QK(block 1) || exp(block 0); V fragments of block 0 stream in
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (!NOFRAG) vfa[kk] = *reinterpret_cast<const bf16x8*>(vb + (kk & 3) * 4096 + vx[kk >> 2]);
            sa10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[kk], qf0[kk], kk == 0 ? negm0 : sa10, 0, 0, 0);
            sa11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[kk], qf1[kk], kk == 0 ? negm1 : sa11, 0, 0, 0);
        }
        AQ_EXPB(sa00, pb0[0], pb0[1], ps00)
        AQ_EXPB(sa01, pb1[0], pb1[1], ps01)
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            if (i_ & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x402, 9, 1);
        }
        if (t == 0 || ragged || !__all(fmaxf(ps00, ps01) <= 8192.0f)) {      // one branch for both halves (keeps the exps above it)
            if (t == 0 || ragged || !__all(ps00 <= 8192.0f)) {
                AQ_SLOW(sa00, sa10, true, 0, 0, t == 0, 0)
                AQ_EXPB(sa00, pb0[0], pb0[1], ps00)
            }
            if (t == 0 || ragged || !__all(ps01 <= 8192.0f)) {
                AQ_SLOW(sa01, sa11, true, 0, 0, t == 0, 1)
                AQ_EXPB(sa01, pb1[0], pb1[1], ps01)
            }
        }
        l_run0 += ps00; l_run1 += ps01;
        // S2: PV(block 0) for both halves from one set of Vt fragments || exp(block 1); block-1 Vt fragments stream in
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VAR == 3) vfb[i] = vfa[i]; else if (!NOFRAG) vfb[i] = *reinterpret_cast<const bf16x8*>(vb + (i & 3) * 4096 + vx[2 + (i >> 2)]);
            oacc0[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[i], pb0[i >> 2], oacc0[i & 3], 0, 0, 0);
            oacc1[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[i], pb1[i >> 2], oacc1[i & 3], 0, 0, 0);
        }
        AQ_EXPB(sa10, pb0[2], pb0[3], ps10)
        AQ_EXPB(sa11, pb1[2], pb1[3], ps11)
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
            if (i_ & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
            __builtin_amdgcn_sched_group_barrier(0x402, 9, 2);
        }
        if (ragged || !__all(fmaxf(ps10, ps11) <= 8192.0f)) {
            if (ragged || !__all(ps10 <= 8192.0f)) {
                AQ_SLOW(sa10, sa00, false, 8192, 32, false, 0)
                AQ_EXPB(sa10, pb0[2], pb0[3], ps10)
            }
            if (ragged || !__all(ps11 <= 8192.0f)) {
                AQ_SLOW(sa11, sa01, false, 8192, 32, false, 1)
                AQ_EXPB(sa11, pb1[2], pb1[3], ps11)
            }
        }
        l_run0 += ps10; l_run1 += ps11;
        // S3: PV(block 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            oacc0[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[i], pb0[2 + (i >> 2)], oacc0[i & 3], 0, 0, 0);
            oacc1[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[i], pb1[2 + (i >> 2)], oacc1[i & 3], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (VAR != 7 && VAR != 8) __syncthreads();
    }

    // ---- epilogue, one 32-query half at a time: lane (q, h) holds O[q][32db + 8a + 4h + c], r = 4a + c
#define AQ_STORE(H, qoff_)                                                                           \
    {                                                                                                \
        const float l_tot = l_run##H + __shfl_xor(l_run##H, 32, 64);                                 \
        const float inv = 1.0f / l_tot;                                                              \
        const int qrow = q0 + (qoff_) + lq;                                                          \
        if (qrow < S) {                                                                              \
            bf16_t* op = p.o + (long)qrow * p.o_ss + head * 128 + 4 * lh;                            \
            _Pragma("unroll") for (int db = 0; db < 4; ++db)                                         \
                _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                      \
                    uint2 v;                                                                         \
                    v.x = pack2bf(oacc##H[db][4 * a + 0] * inv, oacc##H[db][4 * a + 1] * inv);       \
                    v.y = pack2bf(oacc##H[db][4 * a + 2] * inv, oacc##H[db][4 * a + 3] * inv);       \
                    *reinterpret_cast<uint2*>(op + 32 * db + 8 * a) = v;                             \
                }                                                                                    \
        }                                                                                            \
    }
    AQ_STORE(0, 0)
    AQ_STORE(1, 32)
}

template <int PRESC, int VAR>
static int launch_q64(AttnParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_q64_kernel<PRESC, VAR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, AQ_LDS) != hipSuccess) return -3;
        attr_set = true;
    }
    p.nqb = (p.S + 255) / 256;
    hipLaunchKernelGGL((attn_fwd_q64_kernel<PRESC, VAR>), dim3(p.nqb * p.H), dim3(256), AQ_LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int utx_launch_attn_fwd_q64(const AttnParams* p, int presc, hipStream_t stream) {
    static int var = -1;
    if (var < 0) { const char* e = getenv("UTX_ATTN_VAR"); var = e ? atoi(e) : 0; }
    if (presc && var == 1) return launch_q64<1, 1>(*p, stream);
    if (presc && var == 2) return launch_q64<1, 2>(*p, stream);
    if (presc && var == 3) return launch_q64<1, 3>(*p, stream);
    if (presc && var == 4) return launch_q64<1, 4>(*p, stream);
    if (presc && var == 5) return launch_q64<1, 5>(*p, stream);
    if (presc && var == 6) return launch_q64<1, 6>(*p, stream);
    if (presc && var == 7) return launch_q64<1, 7>(*p, stream);
    if (presc && var == 8) return launch_q64<1, 8>(*p, stream);
    return presc ? launch_q64<1, 0>(*p, stream) : launch_q64<0, 0>(*p, stream);
}
