// bf16 MFMA flash-attention forward for the FLUX joint [text; image] sequence (gfx950).
//
// Replaces: torch SDPA call in NativeFluxAttnProcessor2_0.__call__
//   (/root/reference/flux_piplines/texturing/attention_processor.py:89-91): non-causal, no mask,
//   no dropout, head_dim 128, softmax scale 1/sqrt(128).
//
// Layout (produced by qkv_post in dit_elementwise.hip):
//   Q, K : [H][S_pad][128] bf16 (d contiguous)       -- strides passed explicitly
//   Vt   : [H][128][S_pad] bf16 (keys contiguous)    -- V pre-transposed so that the PV product's
//                                                       MFMA A-operand is a 16-byte LDS read
//   O    : [S][H*128] bf16 (token-major; feeds the out-projection GEMM directly)
//
// Structure: workgroup = NW waves x 32 query rows; KV tile = 64 keys.
//   * S^T = K Q^T is computed "swapped" (MFMA A = K, B = Q) so each lane owns one query column and the
//     softmax row reductions are in-register (+ one cross-half exchange).  K rows inside each 32-key block
//     are read through a permutation kappa() chosen so that the C-layout of the 32x32x16 MFMA leaves, in each
//     lane, exactly the 8 consecutive keys the PV MFMA's B-operand wants: P goes to bf16 in registers and
//     straight into the second MFMA (no LDS round trip, no permlane).  O^T = Vt P^T accumulates in 4 x f32x16.
//   * K tile [64][128] / Vt tile [128][64] staged global -> regs -> LDS (issue early, write late) into a 2-deep
//     ring, one s_barrier per tile.  LDS rows are PADDED by 16 B (272 / 144-byte strides) instead of
//     XOR-swizzled: conflict-free for the ds_read_b128 lane groups AND every fragment address is
//     base + immediate, which removes ~20 address VALU ops per tile.
//   * softmax is VALU-issue-bound beside the MFMAs (round-1 PMC: MFMA busy 40 %, VALU busy 40 %, no overlap),
//     so the VALU stream per tile is minimised: the score accumulators are INITIALISED to -m_run, i.e. the MFMA
//     chain itself performs the max subtraction; with Q pre-scaled by scale*log2(e) upstream (qkv_post) a
//     probability is ONE v_exp_f32.  The running max is only moved when some row would exceed 2^8
//     ("defer-max", guide T13); that re-centring path is wave-uniform and rare.
//   * deep operand prefetch: the 8 K fragments of key-block 0 are fetched up front, then one ds_read per MFMA
//     (key-block 1, then the first 8 V fragments under the second half of QK^T).
//
// Algorithmic FLOPs: 4 * S^2 * 128 per head (QK^T + PV, non-causal).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define ATT_KVB 64
#define ATT_D 128
#define ATT_KSTR 272                       // K tile row stride (256 + 16 pad)
#define ATT_VSTR 144                       // Vt tile row stride (128 + 16 pad)
#define ATT_KTILE (64 * ATT_KSTR)          // 17408
#define ATT_VTILE (128 * ATT_VSTR)         // 18432
#define ATT_STAGE (ATT_KTILE + ATT_VTILE)  // 35840
#define ATT_LDS_BYTES (2 * ATT_STAGE)      // 71680

template <int NW, int PRESC>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(AttnParams p) {
    constexpr int ATT_QB = 32 * NW;          // queries per workgroup
    constexpr int NPASS = 1024 / (64 * NW);  // staging passes: 1024 16-byte chunks per K (and per V) tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31;   // query column owned by this lane (MFMA C column)
    const int lh = lane >> 5;   // lane half

    // XCD-aware work mapping: consecutive logical ids (= same head) stay on one XCD's L2.
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int head = w / p.nqb;
    const int qb = w - head * p.nqb;
    const int S = p.S;

    const bf16_t* kbase = p.k + (long)head * p.k_hs;
    const bf16_t* vbase = p.vt + (long)head * p.vt_hs;

    // ---- Q fragments (MFMA B operand): lane (q, h) holds Q[q][16kk + 8h .. +7], kk = 0..7
    const int q0 = qb * ATT_QB + wave * 32;
    bf16x8 qf[8];
    {
        int qrow = q0 + lq;
        if (qrow > S - 1) qrow = S - 1;
        const bf16_t* qp = p.q + (long)head * p.q_hs + (long)qrow * p.q_ss + lh * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
    }

    // ---- staging registers (global -> regs -> LDS).  Named scalars (not arrays) so they stay in VGPRs:
    // hipcc demotes small arrays touched under a runtime branch to scratch.
    uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;
    const int k_slot = tid & 15, v_slot = tid & 7;
#define ATT_SETUP(i_)                                                                            \
    const int krow##i_ = (tid >> 4) + (i_) * (4 * NW);                                           \
    const int vd##i_ = (tid >> 3) + (i_) * (8 * NW);                                             \
    const int k_lds##i_ = krow##i_ * ATT_KSTR + (k_slot << 4);                                   \
    const int v_lds##i_ = ATT_KTILE + vd##i_ * ATT_VSTR + (v_slot << 4);                         \
    const bf16_t* vsrc##i_ = vbase + (long)vd##i_ * p.vt_ds + v_slot * 8;
    ATT_SETUP(0) ATT_SETUP(1) ATT_SETUP(2) ATT_SETUP(3)
#define ATT_LOAD1(i_, kv0_)                                                                      \
    {                                                                                            \
        int r_ = (kv0_) + krow##i_;                                                              \
        if (r_ > S - 1) r_ = S - 1; /* tail rows are masked to -inf later; stay in-bounds */     \
        kreg##i_ = *reinterpret_cast<const uint4*>(kbase + (long)r_ * p.k_ss + k_slot * 8);      \
        vreg##i_ = *reinterpret_cast<const uint4*>(vsrc##i_ + (kv0_));                           \
    }
#define ATT_LOAD_TILE(t_)                                                                        \
    do {                                                                                         \
        const int kv0__ = (t_) * ATT_KVB;                                                        \
        ATT_LOAD1(0, kv0__) ATT_LOAD1(1, kv0__)                                                  \
        if constexpr (NPASS > 2) { ATT_LOAD1(2, kv0__) ATT_LOAD1(3, kv0__) }                     \
    } while (0)
#define ATT_STORE1(i_)                                                                           \
    *reinterpret_cast<uint4*>(st_ + k_lds##i_) = kreg##i_;                                       \
    *reinterpret_cast<uint4*>(st_ + v_lds##i_) = vreg##i_;
#define ATT_STORE_TILE(buf_)                                                                     \
    do {                                                                                         \
        char* st_ = smem + (buf_) * ATT_STAGE;                                                   \
        ATT_STORE1(0) ATT_STORE1(1)                                                              \
        if constexpr (NPASS > 2) { ATT_STORE1(2) ATT_STORE1(3) }                                 \
    } while (0)

    // ---- per-lane LDS fragment bases (everything else is an immediate offset)
    // kappa: MFMA row i = 8a + 4h' + c  ->  key 16(a>>1) + 8h' + 4(a&1) + c   (within a 32-key block)
    const int ka = lq >> 3, khp = (lq >> 2) & 1, kc = lq & 3;
    const int krow = 16 * (ka >> 1) + 8 * khp + 4 * (ka & 1) + kc;
    const int k_off = krow * ATT_KSTR + lh * 16;             // + b*32*ATT_KSTR + kk*32
    const int v_off = ATT_KTILE + lq * ATT_VSTR + lh * 16;   // + db*32*ATT_VSTR + s*32

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = 0.f;     // running (deferred) max, raw score units; set from tile 0
    float l_run = 0.f;     // this lane-half's partial row sum
    const float c2 = p.scale_log2;

    const int nt = (S + ATT_KVB - 1) / ATT_KVB;
    ATT_LOAD_TILE(0);
    ATT_STORE_TILE(0);
    __syncthreads();
    // Retire the Q loads HERE, in the compiler's own scoreboard: otherwise hipcc guards every QK^T MFMA of the
    // loop with vmcnt(7)..vmcnt(0) for "possibly still pending" Q fragments, which drains the NEXT tile's
    // prefetch (issued at the top of the iteration) inside the QK^T phase and serialises HBM latency per tile.
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) ATT_LOAD_TILE(t + 1);
        const char* st = smem + buf * ATT_STAGE;

        // ---- S'^T = K Q^T - m_run : accumulators start at -m_run, so the MFMA chain does the max subtraction
        f32x16 sacc[2];
        const float neg_m = -m_run;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[b][r] = neg_m;
        bf16x8 vf[8];   // first 8 V fragments, fetched under the second half of QK^T
        __builtin_amdgcn_s_setprio(1);
        {
            bf16x8 kf0[8], kf1[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                kf0[kk] = *reinterpret_cast<const bf16x8*>(st + k_off + kk * 32);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                kf1[kk] = *reinterpret_cast<const bf16x8*>(st + k_off + 32 * ATT_KSTR + kk * 32);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[kk], qf[kk], sacc[0], 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                vf[kk] = *reinterpret_cast<const bf16x8*>(st + v_off + (kk & 3) * 32 * ATT_VSTR + (kk >> 2) * 32);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[kk], qf[kk], sacc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i_ = 0; i_ < 16; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // lane (q, h): sacc[b][r] = score(key = kv0 + 32b + 16(r>>3) + 8h + (r&7), query q) - m_run

        if (t == nt - 1 && (S & (ATT_KVB - 1))) {
            const int kv0 = t * ATT_KVB;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * b + 16 * (r >> 3) + 8 * lh + (r & 7);
                    if (key >= S) sacc[b][r] = -INFINITY;
                }
        }

        // ---- online softmax (per query column; partner half = lane ^ 32), deferred max
        float mx = sacc[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // tile max relative to m_run
        const float mxs = PRESC ? mx : mx * c2;      // ... in exp2 units
        if (t == 0 || !__all(mxs <= 8.0f)) {
            // re-centre (wave-uniform, rare after the first tiles): move the running max to the true max
            const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
            // tile 0: O = l = 0, nothing to rescale (and exp2(-d) may overflow for very negative first maxima)
            const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(PRESC ? -d : -d * c2);
            m_run += d;
            l_run *= alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[b][r] -= d;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        float psum = 0.f;
        bf16x8 pb[4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(PRESC ? sacc[b][r] : sacc[b][r] * c2);
                psum += pv;
                pb[2 * b + (r >> 3)][r & 7] = (__bf16)pv;
            }
        l_run += psum;

        // ---- O^T += Vt P^T   (4 blocks of 32 d, 4 k-steps of 16 keys)
        __builtin_amdgcn_s_setprio(1);
        {
            bf16x8 vg[8];   // fragments of PV k-steps 2,3, fetched under k-steps 0,1
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vg[i] = *reinterpret_cast<const bf16x8*>(st + v_off + (i & 3) * 32 * ATT_VSTR + (2 + (i >> 2)) * 32);
                oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pb[i >> 2], oacc[i & 3], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vg[i], pb[2 + (i >> 2)], oacc[i & 3], 0, 0, 0);
#pragma unroll
            for (int i_ = 0; i_ < 8; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (t + 1 < nt) ATT_STORE_TILE(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, convert, store.  lane (q, h) holds O[q][32db + 8a + 4h + c], r = 4a + c
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + lq;
    if (qrow < S) {
        bf16_t* op = p.o + (long)qrow * p.o_ss + head * ATT_D + 4 * lh;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                uint2 v;
                v.x = pack2bf(oacc[db][4 * a + 0] * inv, oacc[db][4 * a + 1] * inv);
                v.y = pack2bf(oacc[db][4 * a + 2] * inv, oacc[db][4 * a + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * db + 8 * a) = v;
            }
    }
}

template <int NW, int PRESC>
static int launch_variant(const AttnParams& p0, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<NW, PRESC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES) != hipSuccess) return -3;
        attr_set = true;
    }
    AttnParams p = p0;
    p.nqb = (p.S + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((attn_fwd_kernel<NW, PRESC>), dim3(p.nqb * p.H), dim3(64 * NW), ATT_LDS_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// UTX_ATTN_WAVES=4 selects the 4-wave / 2-workgroups-per-CU geometry (A/B knob; default 8 waves, 1 per CU)
static int attn_waves() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("UTX_ATTN_WAVES"); v = (e && atoi(e) == 4) ? 4 : 8; }
    return v;
}

// softmax_scale > 0: scores are multiplied by softmax_scale (natural-exp softmax, the reference's SDPA).
// softmax_scale == 0: Q was pre-multiplied by scale*log2(e) upstream (utx_qkv_post q_scale) -> scores are
// already base-2 exponents and a probability is a single v_exp_f32.
extern "C" int utx_launch_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                                   long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                                   long o_ss, int H, int S, float scale, hipStream_t stream) {
    if (S <= 0 || H <= 0 || scale < 0.f) return -1;
    if ((vt_ds & 7) || (q_ss & 7) || (k_ss & 7) || (o_ss & 3)) return -2;
    AttnParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
    p.q_hs = q_hs; p.q_ss = q_ss; p.k_hs = k_hs; p.k_ss = k_ss; p.vt_hs = vt_hs; p.vt_ds = vt_ds;
    p.o_ss = o_ss; p.H = H; p.S = S; p.nqb = 0;
    p.scale_log2 = scale * 1.4426950408889634f;
    const bool presc = (scale == 0.f);
    if (attn_waves() == 4) return presc ? launch_variant<4, 1>(p, stream) : launch_variant<4, 0>(p, stream);
    return presc ? launch_variant<8, 1>(p, stream) : launch_variant<8, 0>(p, stream);
}
