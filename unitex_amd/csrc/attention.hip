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
//   * K tile [64][128] / Vt tile [128][64] staged global -> regs -> LDS (issue early, write late), one
//     s_barrier per tile.  LDS rows are PADDED by 16 B (272 / 144-byte strides) instead of XOR-swizzled:
//     conflict-free for the ds_read_b128 lane groups AND every fragment address is base + immediate.
//   * the softmax VALU stream is minimised: score accumulators are INITIALISED to -m_run (the MFMA chain does
//     the max subtraction); with Q pre-scaled by scale*log2(e) upstream (qkv_post) a probability is ONE
//     v_exp_f32; the running max only moves when some row would exceed 2^8 ("defer-max", guide T13).
//   * FAST (default; UTX_ATTN_FAST=0 selects the per-tile-max form for A/B): tools/coissue_probe.hip shows that on
//     this chip VALU work is NOT hidden under another wave's MFMAs (cross-wave overlap ~0, same-wave ~0.5), so
//     every softmax instruction costs wall time.  The common path therefore has NO max reduction at all: the tile
//     is exponentiated against the running max as it stands (accumulator C-input = a register block holding
//     -m_run, so not even the accumulator initialisation is per-tile VALU work) and only the row sum is checked:
//     if any lane's partial sum exceeds 2^13 (some probability > 2^8 at the least), the tile is redone on the slow
//     path -- QK^T again with the true max, O/l rescaled -- before anything was accumulated.  Per tile and lane
//     that removes 16 v_max3 + 2 v_max + a ds_bpermute round trip + 18 v_mov; the row sum uses v_pk_add_f32.
//     Measured-negative schedules (kept out of the build, logs in profiles/r01_perf_ops_attn_variants.log): 4 waves x
//     2 workgroups / CU, QK(t+1) software-pipelined under softmax(t), late-PV "ping-pong" between wave halves.
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
#define ATT_LDS_BYTES(nvb) (2 * ATT_KTILE + (nvb) * ATT_VTILE)

template <int NW, int PRESC, int FAST>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(AttnParams p) {
    constexpr int ATT_QB = 32 * NW;          // queries per workgroup
    constexpr int NPASS = 1024 / (64 * NW);  // staging passes: 1024 16-byte chunks per K (and per V) tile
    constexpr int NVB = 2;                   // Vt ring depth
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kring = smem;
    char* const vring = smem + 2 * ATT_KTILE;
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
#ifdef UTX_ABLATION
    const int dbg = p.dbg;
#else
    constexpr int dbg = 0;   // timing ablations (wrong results) are compiled only into libunitex_hip_ablate.so
#endif

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
    const int v_lds##i_ = vd##i_ * ATT_VSTR + (v_slot << 4);                                     \
    const bf16_t* vsrc##i_ = vbase + (long)vd##i_ * p.vt_ds + v_slot * 8;                       \
    const bf16_t* ksrc##i_ = kbase + (long)krow##i_ * p.k_ss + k_slot * 8;
    ATT_SETUP(0) ATT_SETUP(1) ATT_SETUP(2) ATT_SETUP(3)
#define ATT_LOAD1(i_, kv0_)                                                                      \
    {   /* K rows / Vt columns are readable up to the next multiple of 64 past S (ABI contract) */  \
        kreg##i_ = *reinterpret_cast<const uint4*>(ksrc##i_ + (long)(kv0_) * p.k_ss);            \
        vreg##i_ = *reinterpret_cast<const uint4*>(vsrc##i_ + (kv0_));                           \
    }
#define ATT_LOAD_TILE(t_)                                                                        \
    do {                                                                                         \
        const int kv0__ = (t_) * ATT_KVB;                                                        \
        ATT_LOAD1(0, kv0__) ATT_LOAD1(1, kv0__)                                                  \
        if constexpr (NPASS > 2) { ATT_LOAD1(2, kv0__) ATT_LOAD1(3, kv0__) }                     \
    } while (0)
#define ATT_STORE1(i_)                                                                           \
    *reinterpret_cast<uint4*>(kst_ + k_lds##i_) = kreg##i_;                                      \
    *reinterpret_cast<uint4*>(vst_ + v_lds##i_) = vreg##i_;
#define ATT_STORE_TILE(kbuf_, vbuf_)                                                             \
    do {                                                                                         \
        char* kst_ = kring + (kbuf_) * ATT_KTILE;                                                \
        char* vst_ = vring + (vbuf_) * ATT_VTILE;                                                \
        ATT_STORE1(0) ATT_STORE1(1)                                                              \
        if constexpr (NPASS > 2) { ATT_STORE1(2) ATT_STORE1(3) }                                 \
    } while (0)

    // ---- per-lane LDS fragment bases (everything else is an immediate offset)
    // kappa: MFMA row i = 8a + 4h' + c  ->  key 16(a>>1) + 8h' + 4(a&1) + c   (within a 32-key block)
    const int ka = lq >> 3, khp = (lq >> 2) & 1, kc = lq & 3;
    const int krow = 16 * (ka >> 1) + 8 * khp + 4 * (ka & 1) + kc;
    const int k_off = krow * ATT_KSTR + lh * 16;   // + b*32*ATT_KSTR + kk*32
    const int v_off = lq * ATT_VSTR + lh * 16;     // + db*32*ATT_VSTR + s*32

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = 0.f;     // running (deferred) max, raw score units; set from tile 0
    float l_run = 0.f;     // this lane-half's partial row sum
    const float c2 = p.scale_log2;
    bf16x8 pb[4];          // P of the current tile as PV B-operands
    f32x16 negm;           // -m_run in every element: C-input of the first QK^T MFMA of each block
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    // O^T += Vt P^T for one tile: 4 blocks of 32 d x 4 k-steps of 16 keys; fragments 8 reads ahead
#define ATT_PV(vb_)                                                                              \
    {                                                                                            \
        const char* vt__ = (vb_) + v_off;                                                        \
        bf16x8 vf_[8], vg_[8];                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
            vf_[i] = *reinterpret_cast<const bf16x8*>(vt__ + (i & 3) * 32 * ATT_VSTR + (i >> 2) * 32); \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                          \
            vg_[i] = *reinterpret_cast<const bf16x8*>(vt__ + (i & 3) * 32 * ATT_VSTR + (2 + (i >> 2)) * 32); \
            oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_[i], pb[i >> 2], oacc[i & 3], 0, 0, 0); \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
            oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vg_[i], pb[2 + (i >> 2)], oacc[i & 3], 0, 0, 0); \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
        }                                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                           \
    }

    const int nt = (S + ATT_KVB - 1) / ATT_KVB;
    ATT_LOAD_TILE(0);
    ATT_STORE_TILE(0, 0);
    __syncthreads();
    // Retire the Q loads HERE, in the compiler's own scoreboard: otherwise hipcc guards every QK^T MFMA of the
    // loop with vmcnt(7)..vmcnt(0) for "possibly still pending" Q fragments, which drains the NEXT tile's
    // prefetch (issued at the top of the iteration) inside the QK^T phase and serialises HBM latency per tile.
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));

    int vcur = 0;   // ring slot of V(t); V(t-1) sits in the previous slot, V(t+1) goes to the next one
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt && !(dbg & 1)) ATT_LOAD_TILE(t + 1);
        const char* kb = kring + (t & 1) * ATT_KTILE;
        const int vnext = (vcur == NVB - 1) ? 0 : vcur + 1;


        if constexpr (FAST == 2) {
            // ---- PIPE: the 64-key tile is processed as two 32-key blocks, software-pipelined inside ONE wave so that the
            // exponentials of a block sit in the shadow of independent MFMAs (same-wave MFMA/VALU interleave is the only
            // overlap this chip grants, tools/coissue_probe.hip):
            //     S0  QK(0)                       S1  QK(1)  ||  exp(0)  -> check(0)
            //     S2  PV(0)  ||  exp(1) -> check(1)                       S3  PV(1)
            // check(b): lane row sums of block b must stay <= 2^13, else the block is redone against its true max (slow
            // path: QK(b) again, O / l rescaled; already-accumulated blocks are consistent with the old max by construction).
            f32x16 sa0, sa1;
            bf16x8 kfa[8], kfb[8], vfa[8], vfb[8];
            const char* vtb = vring + vcur * ATT_VTILE + v_off;
            // the last, partially filled tile takes the slow path for both blocks: keys >= S are masked there (cold code)
            const bool ragged = (t == nt - 1) && (S & (ATT_KVB - 1));
            const int lim = S - t * ATT_KVB - 8 * lh;
#define ATT_EXPB(sa_, p0_, p1_, ps_)                                                                 \
            {                                                                                        \
                f32x2 acc2_ = {0.f, 0.f};                                                            \
                _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                  \
                    f32x2 pv_;                                                                       \
                    pv_[0] = __builtin_amdgcn_exp2f(PRESC ? sa_[r] : sa_[r] * c2);                   \
                    pv_[1] = __builtin_amdgcn_exp2f(PRESC ? sa_[r + 1] : sa_[r + 1] * c2);           \
                    acc2_ += pv_;                                                                    \
                    if (r < 8) { p0_[r] = (__bf16)pv_[0]; p0_[r + 1] = (__bf16)pv_[1]; }             \
                    else { p1_[r - 8] = (__bf16)pv_[0]; p1_[r - 7] = (__bf16)pv_[1]; }               \
                }                                                                                    \
                ps_ = acc2_[0] + acc2_[1];                                                           \
            }
            // slow path for block (sa_, K rows at krow_off_): true max, move the running max, rescale what exists
#define ATT_SLOW(sa_, other_, fix_other_, krow_off_, boff_, first_)                                  \
            {                                                                                        \
                _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                   \
                    const bf16x8 kf_ = *reinterpret_cast<const bf16x8*>(kb + k_off + (krow_off_) + kk * 32); \
                    sa_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[kk], kk == 0 ? negm : sa_, 0, 0, 0); \
                }                                                                                    \
                if (ragged) {                                                                        \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                   \
                        if ((boff_) + 16 * (r >> 3) + (r & 7) >= lim) sa_[r] = -INFINITY;            \
                }                                                                                    \
                float mx_ = sa_[0];                                                                  \
                _Pragma("unroll") for (int r = 1; r < 16; ++r) mx_ = fmaxf(mx_, sa_[r]);             \
                mx_ = fmaxf(mx_, __shfl_xor(mx_, 32, 64));                                           \
                const float d_ = (first_) ? mx_ : fmaxf(mx_, 0.f);                                   \
                const float alpha_ = (first_) ? 1.0f : __builtin_amdgcn_exp2f(PRESC ? -d_ : -d_ * c2); \
                m_run += d_;                                                                         \
                l_run *= alpha_;                                                                     \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) { negm[r] = -m_run; sa_[r] -= d_; }   \
                if (fix_other_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) other_[r] -= d_; }  \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha_;             \
            }
            float ps0 = 0.f, ps1 = 0.f;
            // S0: QK(0); block-1 K fragments stream in behind the MFMAs
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) kfa[kk] = *reinterpret_cast<const bf16x8*>(kb + k_off + kk * 32);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                kfb[kk] = *reinterpret_cast<const bf16x8*>(kb + k_off + 32 * ATT_KSTR + kk * 32);
                sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[kk], qf[kk], kk == 0 ? negm : sa0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i_ = 0; i_ < 8; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // S1: QK(1) || exp(0); V fragments of block 0 stream in
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                vfa[kk] = *reinterpret_cast<const bf16x8*>(vtb + (kk & 3) * 32 * ATT_VSTR + (kk >> 2) * 32);
                sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[kk], qf[kk], kk == 0 ? negm : sa1, 0, 0, 0);
            }
            ATT_EXPB(sa0, pb[0], pb[1], ps0)
#pragma unroll
            for (int i_ = 0; i_ < 8; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x402, 4, 1);
            }
            if (t == 0 || ragged || !__all(ps0 <= 8192.0f)) {
                ATT_SLOW(sa0, sa1, true, 0, 0, t == 0)
                ATT_EXPB(sa0, pb[0], pb[1], ps0)
            }
            l_run += ps0;
            // S2: PV(0) || exp(1); V fragments of block 1 stream in
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vfb[i] = *reinterpret_cast<const bf16x8*>(vtb + (i & 3) * 32 * ATT_VSTR + (2 + (i >> 2)) * 32);
                oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[i], pb[i >> 2], oacc[i & 3], 0, 0, 0);
            }
            ATT_EXPB(sa1, pb[2], pb[3], ps1)
#pragma unroll
            for (int i_ = 0; i_ < 8; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x402, 4, 2);
            }
            if (ragged || !__all(ps1 <= 8192.0f)) {
                ATT_SLOW(sa1, sa0, false, 32 * ATT_KSTR, 32, false)
                ATT_EXPB(sa1, pb[2], pb[3], ps1)
            }
            l_run += ps1;
            // S3: PV(1)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[i], pb[2 + (i >> 2)], oacc[i & 3], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        } else {
        f32x16 sacc[2];
        float psum = 1.0f;
        // S'^T = K Q^T - m_run : the accumulator chain starts from the -m_run block, so the MFMAs do the subtraction.
        // lane (q, h): sacc[b][r] = score(key = kv0 + 32b + 16(r>>3) + 8h + (r&7), query q) - m_run
#define ATT_QK()                                                                                         \
        {                                                                                                \
            __builtin_amdgcn_s_setprio(1);                                                               \
            bf16x8 kf0[8], kf1[8];                                                                       \
            /* the two 32-key blocks alternate so consecutive MFMAs never share an accumulator; fragments  \
               run 4 k-steps (8 reads) ahead */                                                          \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                           \
                kf0[kk] = *reinterpret_cast<const bf16x8*>(kb + k_off + kk * 32);                        \
                kf1[kk] = *reinterpret_cast<const bf16x8*>(kb + k_off + 32 * ATT_KSTR + kk * 32);        \
            }                                                                                            \
            _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                           \
                if (kk + 4 < 8) {                                                                        \
                    kf0[kk + 4] = *reinterpret_cast<const bf16x8*>(kb + k_off + (kk + 4) * 32);          \
                    kf1[kk + 4] = *reinterpret_cast<const bf16x8*>(kb + k_off + 32 * ATT_KSTR + (kk + 4) * 32); \
                }                                                                                        \
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[kk], qf[kk], kk == 0 ? negm : sacc[0], 0, 0, 0); \
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[kk], qf[kk], kk == 0 ? negm : sacc[1], 0, 0, 0); \
            }                                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                           \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
            }                                                                                            \
            __builtin_amdgcn_s_setprio(0);                                                               \
            if (t == nt - 1 && (S & (ATT_KVB - 1))) {   /* ragged last tile: keys >= S -> -inf */        \
                const int lim_ = S - t * ATT_KVB - 8 * lh;                                               \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                            \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                       \
                        if (32 * b + 16 * (r >> 3) + (r & 7) >= lim_) sacc[b][r] = -INFINITY;            \
            }                                                                                            \
        }
        // probabilities (one v_exp_f32 each when Q is pre-scaled), bf16 PV operands, packed row sum
#define ATT_EXP()                                                                                        \
        {                                                                                                \
            f32x2 ps2_ = {0.f, 0.f};                                                                     \
            _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                \
                _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                      \
                    f32x2 pv_;                                                                           \
                    pv_[0] = __builtin_amdgcn_exp2f(PRESC ? sacc[b][r] : sacc[b][r] * c2);               \
                    pv_[1] = __builtin_amdgcn_exp2f(PRESC ? sacc[b][r + 1] : sacc[b][r + 1] * c2);       \
                    ps2_ += pv_;                                                                         \
                    pb[2 * b + (r >> 3)][r & 7] = (__bf16)pv_[0];                                        \
                    pb[2 * b + (r >> 3)][(r & 7) + 1] = (__bf16)pv_[1];                                  \
                }                                                                                        \
            psum = ps2_[0] + ps2_[1];                                                                    \
        }
        bool slow = (FAST == 0) || (t == 0);
        if (!slow) {
            // common path: no max reduction.  Any probability above 2^13 shows up in the lane's row sum; the
            // tile is then redone below against the true max (nothing has been accumulated yet).
            if (!(dbg & 16)) ATT_QK()
            if (!(dbg & 2)) ATT_EXP()
            slow = !__all(psum <= 8192.0f);
        }
        if (slow) {
            ATT_QK()
            float mx = sacc[0][0];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // tile max relative to m_run
            if (FAST != 0 || t == 0 || !__all((PRESC ? mx : mx * c2) <= 8.0f)) {
                // re-centre (wave-uniform): move the running max to the true max
                const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
                // tile 0: O = l = 0, nothing to rescale (and exp2(-d) may overflow for very negative first maxima)
                const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(PRESC ? -d : -d * c2);
                m_run += d;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_run;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[b][r] -= d;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            ATT_EXP()
        }
        l_run += psum;

        if (!(dbg & 8)) ATT_PV(vring + vcur * ATT_VTILE)
        }

        if (t + 1 < nt && !(dbg & 1)) ATT_STORE_TILE((t + 1) & 1, vnext);
        if (!(dbg & 4)) __syncthreads();
        vcur = vnext;
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

template <int NW, int PRESC, int FAST>
static int launch_variant(const AttnParams& p0, hipStream_t stream) {
    constexpr int lds = ATT_LDS_BYTES(2);
    UTX_ONCE_PER_DEVICE(attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<NW, PRESC, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    AttnParams p = p0;
    p.nqb = (p.S + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((attn_fwd_kernel<NW, PRESC, FAST>), dim3(p.nqb * p.H), dim3(64 * NW), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// A/B knob (utx_set_option UTX_ATTN_FAST): 0 selects the per-tile-max softmax of the register-staged kernel.
static int attn_fast() { return g_utx_opt.attn_fast; }

// softmax_scale > 0: scores are multiplied by softmax_scale (natural-exp softmax, the reference's SDPA).
// softmax_scale == 0: Q was pre-multiplied by scale*log2(e) upstream (utx_qkv_post q_scale) -> scores are
// already base-2 exponents and a probability is a single v_exp_f32.
extern "C" int utx_launch_attn_fwd(const void* q, const void* k, const void* vt, void* o,
                                   long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                                   long o_ss, int H, int S, int Sq, float scale, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes,
                                   hipStream_t stream) {
    return utx_launch_attn_fwd_blk(q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S, Sq, scale, key_bias_log2, key_bias_period, work, work_bytes,
                                   0, 0, 0, 0, stream);
}

// blk_rows > 0: Q / K / V^T in blocks of blk_rows tokens q_bs / k_bs / vt_bs elements apart (attention_glds.hip BLK; the default kernel only)
extern "C" int utx_launch_attn_fwd_blk(const void* q, const void* k, const void* vt, void* o,
                                       long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs, long vt_ds,
                                       long o_ss, int H, int S, int Sq, float scale, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes,
                                       int blk_rows, long q_bs, long k_bs, long vt_bs, hipStream_t stream) {
    // Sq > S is a plain case (round 6): queries and keys are separate arrays and nothing ties a query row to a key row -- the sequence-parallel launch whose keys carry the
    // ranks' identical text rows once (utx_sp_unpack_qkv_dedup) has P x 64 - 64 more queries than keys.  Block-strided operands share their blocks: Sq <= S there.
    if (S <= 0 || H <= 0 || scale < 0.f || key_bias_period < 0 || Sq < 0 || (Sq > S && blk_rows > 0) || blk_rows < 0) return -1;
    if (Sq == S) Sq = 0;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)vt)) & 15) return -2;                       // 16-byte aligned bases (LDS-DMA / b128 loads)
    if ((vt_ds & 7) || (q_ss & 7) || (k_ss & 7) || (o_ss & 3) || (q_hs & 7) || (k_hs & 7) || (vt_hs & 7)) return -2;   // 16-byte rows
    AttnParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
    p.q_hs = q_hs; p.q_ss = q_ss; p.k_hs = k_hs; p.k_ss = k_ss; p.vt_hs = vt_hs; p.vt_ds = vt_ds;
    p.o_ss = o_ss; p.H = H; p.S = S; p.nqb = 0; p.Sq = Sq;
    p.dbg = g_utx_opt.attn_debug_abl;   // always 0 in the product library (capi.cpp)
    p.scale_log2 = scale * 1.4426950408889634f;
    const bool presc = (scale == 0.f);
    p.flags = nullptr; p.flag_hs = 0;
    p.key_bias_log2 = key_bias_log2; p.key_bias_period = key_bias_period;
    p.work = work; p.work_bytes = work_bytes;
    p.blk_rows = blk_rows; p.q_bs = q_bs; p.k_bs = k_bs; p.vt_bs = vt_bs;
    if (blk_rows > 0 && (g_utx_opt.attn_glds == 0 || g_utx_opt.attn_tpb != 1)) return -2;
    // key multiplicity and a query count below the key count exist in the LDS-DMA staged kernels only
    if ((key_bias_log2 != 0.f || Sq != 0) && (g_utx_opt.attn_glds == 0 || g_utx_opt.attn_tpb != 1)) return -2;
    // UTX_ATTN_Q64=1: the 4 x 64 kernel (attention_q64.hip: one wave per SIMD, hand-placed stream) + its repair pass, for the launches it takes (pre-scaled Q, whole 64-key
    // tiles, contiguous operands, no periodic key multiplicity, caller scratch with room for its flags); everything else runs the 8 x 32 kernel below
    if (g_utx_opt.attn_q64 == 1 && g_utx_opt.attn_glds != 0 && g_utx_opt.attn_tpb == 1 && utx_attn_q64_takes(&p, presc ? 1 : 0)) return utx_launch_attn_fwd_q64(&p, 1, stream);
    // default: the LDS-DMA staged kernel (attention_glds.hip), +5 % over register staging (profiles/r01_perf_attn_ablation.log);
    // UTX_ATTN_GLDS=0 selects the register-staged variants below for A/B
    if (g_utx_opt.attn_glds != 0) return utx_launch_attn_fwd_glds(&p, presc ? 1 : 0, stream);
    const int fast = attn_fast();   // 2 = block-pipelined sum-checked softmax (default), 1 = sum-checked, 0 = per-tile max
    if (fast == 0) return presc ? launch_variant<8, 1, 0>(p, stream) : launch_variant<8, 0, 0>(p, stream);
    if (fast == 1) return presc ? launch_variant<8, 1, 1>(p, stream) : launch_variant<8, 0, 1>(p, stream);
    return presc ? launch_variant<8, 1, 2>(p, stream) : launch_variant<8, 0, 2>(p, stream);
}
