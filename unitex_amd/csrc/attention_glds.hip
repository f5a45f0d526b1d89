// bf16 MFMA flash-attention forward, LDS-DMA staged variant (gfx950).  Same contract, maths and block-pipelined
// sum-checked softmax as attn_fwd_kernel<8, PRESC, 2> in attention.hip (see the header there); what differs is how the
// K / Vt tiles reach LDS:
//   attention.hip      global_load -> 8 staging VGPRs x 4 -> ds_write_b128 (padded rows, immediate fragment offsets)
//   this file          global_load_lds (16 B / lane, no VGPRs, no ds_write): the LDS image of a wave-instruction is
//                      lane-linear (1 KB = 4 K rows or 8 Vt rows), so rows cannot be padded; bank conflicts are
//                      removed by an XOR swizzle applied on the SOURCE address (which 16-byte chunk a lane fetches)
//                      and on the fragment read address:
//                          K  tile [64 keys][256 B] : slot = chunk ^ (row & 15)
//                          Vt tile [128 d ][128 B]  : slot = chunk ^ ((row >> 1) & 7)
//                      both conflict-free for the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}.
// Why: round-1 ablation (profiles/r01_perf_attn_ablation.log) charges 2.9 of 27.8 ms per launch at S = 50 688 to the
// register staging (a ds_write_b128 costs ~13 LDS cycles per wave-instruction, 4 per lane and tile) -- the kernel is
// power/clock limited, so fewer instructions and less register traffic per tile is the lever that is left.
// Ring: K and Vt 2-deep (64 KB); tile t+1 is requested at the top of tile t into the slot last read in tile t-1 (free
// since the barrier that ended it) and retired by the explicit vmcnt(0) in front of that barrier (AG_BARRIER below).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

#define AG_KVB 64
#define AG_KTILE 16384
#define AG_VTILE 16384
#define AG_LDS(tpb) ((tpb) * 2 * (AG_KTILE + AG_VTILE))

// The barrier that publishes a DMA'd tile: this wave's LDS-DMAs retire through vmcnt, and the wait for them must be WRITTEN -- hipcc does not owe a global_load_lds a
// `vmcnt(0)` at __syncthreads() (cdna_hip_programming.md 5.7: "LDS-DMA data needs your own vmcnt(N), then a barrier, then the ds_read").  Until round 5 this file relied on
// the `s_waitcnt vmcnt(0) lgkmcnt(0)` hipcc happened to emit at the fence; built with -mllvm -amdgpu-sched-strategy=max-memory-clause the first barrier of the two-tile loop
// came out with `lgkmcnt(0)` only and the fast loop returned run-to-run different results at full occupancy (profiles/r05_attn_lib_compare.log).  With the default
// strategy the explicit wait is redundant (the listing shows both); tests/test_asm_hazards_cpu.py now requires a vmcnt(0) in front of every such barrier.
#define AG_BARRIER() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

__device__ __forceinline__ void ag_glds16(const bf16_t* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// l * alpha as a multiply of its own: under -ffp-contract=fast hipcc turned `l_run *= alpha; ...; l_run += ps` into v_fmac_f32 in the tail-duplicated copies of the re-centring
// path (the peeled loops) and left mul + add in the general loop -- one-ulp differences in l, 36 of 9.2 M outputs (profiles/r05_peel_diff_probe.log)
__device__ __forceinline__ float ag_mul_nofuse(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

#define AG_MM(acc_, a_, b_, c_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0);

// BLK: Q rows, K rows and V^T columns arrive in BLOCKS of p.blk_rows tokens (a multiple of the 64-key tile) that lie p.q_bs / p.k_bs / p.vt_bs elements apart --
// the receive buffer of the sequence-parallel Q / K / V exchange, [source rank][q | k | v][head][S_loc * 128] (flux/ulysses.py), read where the all-to-all put it
// instead of behind a relayout pass.  Token j = block j / blk_rows, row j % blk_rows; inside a block rows are q_ss / k_ss apart and V^T rows vt_ds (= blk_rows for
// the exchange buffer).  The staging cursor below walks tiles in order, so the block term is two scalar adds per tile; same tiles, same order, same arithmetic
// as the contiguous form: bit-identical results.
template <int PRESC, int TPB, bool BLK = false, bool FAST = false, bool KBP = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_glds_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kring = smem;
    char* const vring = smem + 2 * TPB * AG_KTILE;    // ring: 2 groups of TPB tiles, one barrier per group
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int nsp = p.nsplit;                        // > 1: this launch covers the tail items, each cut along the keys
    const int item = nsp > 1 ? wid / nsp : wid;
    const int split = wid - item * nsp;
    const int w = p.w_base + item;
    const int head = w / p.nqb;
    const int qb = w - head * p.nqb;
    const int S = p.S;                                // keys
    const int Sq = p.Sq > 0 ? p.Sq : p.S;             // query rows (Sq < S: only the first Sq rows of q are queries -- last-block pruning, flux/transformer.py)
    // keys of this workgroup: all of them, or tiles [tb, tb + tiles_per_split) of the sequence (base pointers are advanced, Sk counts from there)
    const int tb = nsp > 1 ? split * p.tiles_per_split : 0;
    const int Sk = nsp > 1 ? ((S - tb * AG_KVB < p.tiles_per_split * AG_KVB) ? S - tb * AG_KVB : p.tiles_per_split * AG_KVB) : S;
    if (p.flags) {
        // repair pass behind the 4 x 64 kernel: only query blocks in which one of its waves ran out of softmax headroom
        const unsigned char* f = p.flags + head * p.flag_hs + qb * 4;
        int any = 0;
        for (int g = 0; g < 4; ++g) any |= (qb * 4 + g < p.flag_hs) ? f[g] : 0;
        if (!any) return;
    }
    const int tpblk = BLK ? p.blk_rows / AG_KVB : 0;          // 64-key tiles per block
    const int blk0 = BLK ? tb / tpblk : 0, loc0 = BLK ? tb - blk0 * tpblk : 0;
    const bf16_t* kbase = BLK ? p.k + (long)head * p.k_hs + (long)blk0 * p.k_bs + (long)loc0 * AG_KVB * p.k_ss
                              : p.k + (long)head * p.k_hs + (long)tb * AG_KVB * p.k_ss;
    const bf16_t* vbase = BLK ? p.vt + (long)head * p.vt_hs + (long)blk0 * p.vt_bs + loc0 * AG_KVB
                              : p.vt + (long)head * p.vt_hs + tb * AG_KVB;

    const int q0 = qb * 256 + wave * 32;
    bf16x8 qf[8];
    {
        int qrow = q0 + lq;
        if (qrow > Sq - 1) qrow = Sq - 1;
        const int qblk = BLK ? qrow / p.blk_rows : 0;
        const bf16_t* qp = BLK ? p.q + (long)head * p.q_hs + (long)qblk * p.q_bs + (long)(qrow - qblk * p.blk_rows) * p.q_ss + lh * 8
                               : p.q + (long)head * p.q_hs + (long)qrow * p.q_ss + lh * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
    }

    // ---- DMA sources: wave-instruction (wave, j) fills LDS bytes [(2*wave + j) * 1024, +1024) of a tile
    //   K : slot s = (2*wave+j)*64 + lane -> row s>>4, LDS chunk s&15 <- global chunk (s&15) ^ (row&15)
    //   Vt: slot s                        -> row s>>3, LDS chunk s&7  <- global chunk (s&7) ^ ((row>>1)&7)
    const int ks0 = (2 * wave) * 64 + lane, ks1 = ks0 + 64;
    const int kr0 = ks0 >> 4, kr1 = ks1 >> 4;
    const bf16_t* ksrc0 = kbase + (long)kr0 * p.k_ss + (((ks0 & 15) ^ (kr0 & 15)) << 3);
    const bf16_t* ksrc1 = kbase + (long)kr1 * p.k_ss + (((ks1 & 15) ^ (kr1 & 15)) << 3);
    const int vr0 = ks0 >> 3, vr1 = ks1 >> 3;
    const bf16_t* vsrc0 = vbase + (long)vr0 * p.vt_ds + (((ks0 & 7) ^ ((vr0 >> 1) & 7)) << 3);
    const bf16_t* vsrc1 = vbase + (long)vr1 * p.vt_ds + (((ks1 & 7) ^ ((vr1 >> 1) & 7)) << 3);
    const int dma_off = (2 * wave) * 1024;
    // BLK: the staging cursor (tiles are requested strictly in order, once each): element offsets of the next tile from kbase / vbase, wave-uniform
    typedef typename std::conditional<BLK, long, int>::type ag_vadv_t;      // (the contiguous form keeps its 32-bit V^T column offset)
    long st_k = 0, st_v = 0;
    int st_loc = loc0;
#define AG_STAGE(t_, slot_)                                                                      \
    do {                                                                                         \
        const long kadv_ = BLK ? st_k : (long)(t_) * AG_KVB * p.k_ss;                            \
        const ag_vadv_t vadv_ = BLK ? (ag_vadv_t)st_v : (ag_vadv_t)((t_) * AG_KVB);              \
        ag_glds16(ksrc0 + kadv_, kring + (slot_) * AG_KTILE + dma_off);                          \
        ag_glds16(ksrc1 + kadv_, kring + (slot_) * AG_KTILE + dma_off + 1024);                   \
        ag_glds16(vsrc0 + vadv_, vring + (slot_) * AG_VTILE + dma_off);                          \
        ag_glds16(vsrc1 + vadv_, vring + (slot_) * AG_VTILE + dma_off + 1024);                   \
        if (BLK) {                                                                               \
            st_k += (long)AG_KVB * p.k_ss; st_v += AG_KVB;                                       \
            if (++st_loc == tpblk) { st_loc = 0; st_k += p.k_bs - (long)p.blk_rows * p.k_ss; st_v += p.vt_bs - p.blk_rows; } \
        }                                                                                        \
    } while (0)

    // ---- fragment read offsets.  kappa: MFMA row i = 8a + 4h' + c -> key 16(a>>1) + 8h' + 4(a&1) + c (attention.hip)
    const int ka = lq >> 3, khp = (lq >> 2) & 1, kc = lq & 3;
    const int krow = 16 * (ka >> 1) + 8 * khp + 4 * (ka & 1) + kc;
    const int kswz = krow & 15, vswz = (lq >> 1) & 7;
    int kx[8], vx[4];     // byte offset of (row, chunk 2kk+lh) / (row, chunk 2s+lh) inside a tile
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kx[kk] = krow * 256 + (((2 * kk + lh) ^ kswz) << 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) vx[s] = lq * 128 + (((2 * s + lh) ^ vswz) << 4);

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    const float c2 = p.scale_log2;
    bf16x8 pb[4];
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    const int nt = (Sk + AG_KVB - 1) / AG_KVB;
    const int ngrp = (nt + TPB - 1) / TPB;
#pragma unroll
    for (int i = 0; i < TPB; ++i)
        if (i < nt) AG_STAGE(i, i);
    AG_BARRIER();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));

    // AG_TILE_BODY: the GENERAL tile (it may be the first tile, a ragged last tile, a key-multiplicity tile); AG_FAST_A / AG_FAST_B below: a tile that is none of these.
#define AG_EXPB(sa_, p0_, p1_, ps_)                                                                  \
        {                                                                                            \
            float sc0_ = 0.f, sc1_ = 0.f;                                                            \
            _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                      \
                f32x2 pv_;                                                                           \
                pv_[0] = __builtin_amdgcn_exp2f(PRESC ? sa_[r] : sa_[r] * c2);                       \
                pv_[1] = __builtin_amdgcn_exp2f(PRESC ? sa_[r + 1] : sa_[r + 1] * c2);               \
                sc0_ += pv_[0]; sc1_ += pv_[1];                                                      \
                if (r < 8) { p0_[r] = (__bf16)pv_[0]; p0_[r + 1] = (__bf16)pv_[1]; }                 \
                else { p1_[r - 8] = (__bf16)pv_[0]; p1_[r - 7] = (__bf16)pv_[1]; }                   \
            }                                                                                        \
            ps_ = sc0_ + sc1_;                                                                       \
        }
#define AG_SLOW(sa_, other_, fix_other_, kblk_, boff_, first_)                                       \
        {                                                                                            \
            _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                       \
                const bf16x8 kf_ = *reinterpret_cast<const bf16x8*>(kb + (kblk_) + kx[kk]);          \
                sa_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[kk], kk == 0 ? negm : sa_, 0, 0, 0); \
            }                                                                                        \
            if (kbias) { _Pragma("unroll") for (int r = 0; r < 16; ++r) sa_[r] += kbv; }             \
            if (ragged) {                                                                            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                       \
                    if ((boff_) + 16 * (r >> 3) + (r & 7) >= lim) sa_[r] = -INFINITY;                \
            }                                                                                        \
            float mx_ = sa_[0];                                                                      \
            _Pragma("unroll") for (int r = 1; r < 16; ++r) mx_ = fmaxf(mx_, sa_[r]);                 \
            mx_ = fmaxf(mx_, __shfl_xor(mx_, 32, 64));                                               \
            const float d_ = (first_) ? mx_ : fmaxf(mx_, 0.f);                                       \
            const float alpha_ = (first_) ? 1.0f : __builtin_amdgcn_exp2f(PRESC ? -d_ : -d_ * c2);   \
            m_run += d_;                                                                             \
            l_run = ag_mul_nofuse(l_run, alpha_);   /* never contracted with the `l_run += ps` behind the branch: hipcc fuses them in some copies of this block and not in others */ \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { negm[r] = -m_run; sa_[r] -= d_; }       \
            if (fix_other_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) other_[r] -= d_; }      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha_;                 \
        }
#define AG_TILE_BODY                                                                            \
        {                                                                                            \
        const char* kb = kring + (gs + sub) * AG_KTILE;                                              \
        const char* vb = vring + (gs + sub) * AG_VTILE;                                              \
        const bool ragged = (t == nt - 1) && (Sk & (AG_KVB - 1));                                      \
        const int lim = Sk - t * AG_KVB - 8 * lh;                                                    \
        f32x16 sa0, sa1;                                                                             \
        bf16x8 kfa[8], kfb[8], vfa[8], vfb[8];                                                       \
        float ps0 = 0.f, ps1 = 0.f;                                                                  \
        /* key multiplicity: every key of this tile stands for 2^key_bias_log2 identical keys (text-token dedup) -> bias on its scores */ \
        const int tg = tb + t;   /* tile index in the whole sequence */                              \
        const bool kbias = (p.key_bias_log2 != 0.f) && (p.key_bias_period > 0 ? (tg % p.key_bias_period == 0) : (tg == 0)); \
        const float kbv = PRESC ? p.key_bias_log2 : p.key_bias_log2 / c2;                            \
        /* S0: QK(0); block-1 K fragments stream in behind the MFMAs */                              \
        _Pragma("unroll")                                                                            \
        for (int kk = 0; kk < 8; ++kk) kfa[kk] = *reinterpret_cast<const bf16x8*>(kb + kx[kk]);      \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll")                                                                            \
        for (int kk = 0; kk < 8; ++kk) {                                                             \
            kfb[kk] = *reinterpret_cast<const bf16x8*>(kb + 8192 + kx[kk]);                          \
            if (kk == 0) { AG_MM(sa0, kfa[kk], qf[kk], negm) } else { AG_MM(sa0, kfa[kk], qf[kk], sa0) } \
        }                                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                           \
        _Pragma("unroll")                                                                            \
        for (int i_ = 0; i_ < 8; ++i_) {                                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
        }                                                                                            \
        /* S1: QK(1) || exp(0); V fragments of block 0 (key chunks s = 0, 1) stream in */            \
        _Pragma("unroll")                                                                            \
        for (int kk = 0; kk < 8; ++kk) {                                                             \
            vfa[kk] = *reinterpret_cast<const bf16x8*>(vb + (kk & 3) * 4096 + vx[kk >> 2]);          \
            if (kk == 0) { AG_MM(sa1, kfb[kk], qf[kk], negm) } else { AG_MM(sa1, kfb[kk], qf[kk], sa1) } \
        }                                                                                            \
        if (kbias) {                                                                                 \
        _Pragma("unroll")                                                                            \
            for (int r = 0; r < 16; ++r) sa0[r] += kbv;                                              \
        }                                                                                            \
        AG_EXPB(sa0, pb[0], pb[1], ps0)                                                              \
        _Pragma("unroll")                                                                            \
        for (int i_ = 0; i_ < 8; ++i_) {                                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);                                       \
            __builtin_amdgcn_sched_group_barrier(0x402, 4, 1);                                       \
        }                                                                                            \
        if (t == 0 || ragged || !__all(ps0 <= 8192.0f)) {                                            \
            AG_SLOW(sa0, sa1, true, 0, 0, t == 0)                                           \
            AG_EXPB(sa0, pb[0], pb[1], ps0)                                                          \
        }                                                                                            \
        l_run += ps0;                                                                                \
        /* S2: PV(0) || exp(1); V fragments of block 1 (s = 2, 3) stream in */                       \
        _Pragma("unroll")                                                                            \
        for (int i = 0; i < 8; ++i) {                                                                \
            vfb[i] = *reinterpret_cast<const bf16x8*>(vb + (i & 3) * 4096 + vx[2 + (i >> 2)]);       \
            AG_MM(oacc[i & 3], vfa[i], pb[i >> 2], oacc[i & 3])                                      \
        }                                                                                            \
        if (kbias) {                                                                                 \
        _Pragma("unroll")                                                                            \
            for (int r = 0; r < 16; ++r) sa1[r] += kbv;                                              \
        }                                                                                            \
        AG_EXPB(sa1, pb[2], pb[3], ps1)                                                              \
        _Pragma("unroll")                                                                            \
        for (int i_ = 0; i_ < 8; ++i_) {                                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);                                       \
            __builtin_amdgcn_sched_group_barrier(0x402, 4, 2);                                       \
        }                                                                                            \
        if (ragged || !__all(ps1 <= 8192.0f)) {                                                      \
            AG_SLOW(sa1, sa0, false, 8192, 32, false)                                                \
            AG_EXPB(sa1, pb[2], pb[3], ps1)                                                          \
        }                                                                                            \
        l_run += ps1;                                                                                \
        /* S3: PV(1) */                                                                              \
        _Pragma("unroll")                                                                            \
        for (int i = 0; i < 8; ++i)                                                                  \
            AG_MM(oacc[i & 3], vfb[i], pb[2 + (i >> 2)], oacc[i & 3])                                \
        __builtin_amdgcn_s_setprio(0);                                                               \
        }
    // FAST (the default of the pre-scaled contiguous launch since round 5; UTX_ATTN_PEEL=0 selects the general loop for A/B): tile 0 and a ragged last tile run the general
    // body in front of / behind a loop whose tiles can be neither first, ragged nor key-multiplicity tiles.  In the general body `if (kbias)` and the first-tile test cut S1 / S2
    // into several basic blocks and hipcc puts the MFMAs into one block and the exponentials into the next: within a wave matrix and VALU work never overlap.  The fast tile is
    // branch-free up to the sum checks, and it is cut in two around the tile's ONE barrier, which moves from the end of the tile to between S2 and S3: S3 (PV of block 1) reads
    // nothing from LDS -- its V fragments came in during S2 -- so behind that barrier (a) the ring slot of tile t is free and takes the DMA of tile t + 2, and (b) tile t + 1,
    // requested a whole tile earlier, has landed and is visible: its first eight K fragments are read UNDER the S3 MFMAs into registers that live across the back edge, and the
    // next tile opens with an MFMA where the general loop has all eight waves issue sixteen ds_read_b128 behind the barrier with the matrix pipe idle.  Same arithmetic in the
    // same order per element as the general loop: bit-identical (tests/test_attention_peel_gpu.py).  Measured (profiles/r05_attn_peel_ab.log, same process, interleaved): 25.86 ->
    // 23.96 ms at S = 50 240 (1199 -> 1294 TF/s), 1.851 -> 1.756 ms at 13 376.  The S1 / S2 sched_group_barrier hints are NOT used here: with them 23.99 ms (no gain), and on the
    // peeled loop with the barrier at the end they LOSE (25.27 vs 24.41 ms): they place exponentials between the QK^T(1) MFMAs, which chain on one accumulator.  Launches whose
    // key-multiplicity tiles recur (key_bias_period > 0: sequence parallelism) run the KBP instance: runs of ordinary tiles in an inner loop, the key-multiplicity tile between two runs through a copy of its own.
#define AG_EXPF(sa_, p0_, p1_, ps_, qi_) AG_EXPB(sa_, p0_, p1_, ps_)
#define AG_LOAD_KFA(slot_)                                                                           \
        { _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) kfa_n[kk] = *reinterpret_cast<const bf16x8*>(kring + (slot_) * AG_KTILE + kx[kk]); }
#define AG_FAST_A(KB_)                                                                               \
        {                                                                                            \
        const char* kb = kring + gs * AG_KTILE;                                                      \
        const char* vb = vring + gs * AG_VTILE;                                                      \
        const bool ragged = false, kbias = (KB_);   /* KB_ literal: a key-multiplicity tile of a periodic launch (sequence parallelism, KBP instance) has a straight-line copy of its own; AG_SLOW's branches fold */ \
        const int lim = 0;                                                                           \
        const float kbv = PRESC ? p.key_bias_log2 : p.key_bias_log2 / c2;                            \
        f32x16 sa0, sa1;                                                                             \
        bf16x8 kfb[8], vfa[8];                                                                       \
        float ps0 = 0.f, ps1 = 0.f;                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                               \
        /* S0: QK(0) on the prefetched fragments; block-1 K fragments stream in behind the MFMAs */  \
        _Pragma("unroll")                                                                            \
        for (int kk = 0; kk < 8; ++kk) {                                                             \
            kfb[kk] = *reinterpret_cast<const bf16x8*>(kb + 8192 + kx[kk]);                          \
            if (kk == 0) { AG_MM(sa0, kfa_n[kk], qf[kk], negm) } else { AG_MM(sa0, kfa_n[kk], qf[kk], sa0) } \
            __builtin_amdgcn_sched_barrier(0);                                                       \
        }                                                                                            \
        /* S1: QK(1) || exp(0); V fragments of block 0 stream in */                                  \
        _Pragma("unroll")                                                                            \
        for (int kk = 0; kk < 8; ++kk) {                                                             \
            vfa[kk] = *reinterpret_cast<const bf16x8*>(vb + (kk & 3) * 4096 + vx[kk >> 2]);          \
            if (kk == 0) { AG_MM(sa1, kfb[kk], qf[kk], negm) } else { AG_MM(sa1, kfb[kk], qf[kk], sa1) } \
        }                                                                                            \
        if (KB_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) sa0[r] += kbv; }                    \
        AG_EXPF(sa0, pb[0], pb[1], ps0, 0)                                                              \
        if (!__all(ps0 <= 8192.0f)) {                                                                \
            AG_SLOW(sa0, sa1, true, 0, 0, false)                                                     \
            AG_EXPF(sa0, pb[0], pb[1], ps0, 0)                                                          \
        }                                                                                            \
        l_run += ps0;                                                                                \
        /* S2: PV(0) || exp(1); V fragments of block 1 stream in (they outlive this macro: S3 sits behind the barrier) */ \
        _Pragma("unroll")                                                                            \
        for (int i = 0; i < 8; ++i) {                                                                \
            vfb_n[i] = *reinterpret_cast<const bf16x8*>(vb + (i & 3) * 4096 + vx[2 + (i >> 2)]);     \
            AG_MM(oacc[i & 3], vfa[i], pb[i >> 2], oacc[i & 3])                                      \
        }                                                                                            \
        if (KB_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) sa1[r] += kbv; }                    \
        AG_EXPF(sa1, pb[2], pb[3], ps1, 2)                                                              \
        if (!__all(ps1 <= 8192.0f)) {                                                                \
            AG_SLOW(sa1, sa0, false, 8192, 32, false)                                                \
            AG_EXPF(sa1, pb[2], pb[3], ps1, 2)                                                          \
        }                                                                                            \
        l_run += ps1;                                                                                \
        }
    /* S3: PV(1); PF_ (literal): the next tile's first K fragments come in from ring slot gs ^ 1 behind the MFMAs */
#define AG_FAST_B(PF_)                                                                               \
        {                                                                                            \
        _Pragma("unroll")                                                                            \
        for (int i = 0; i < 8; ++i) {                                                                \
            if (PF_) kfa_n[i] = *reinterpret_cast<const bf16x8*>(kring + (gs ^ 1) * AG_KTILE + kx[i]); \
            AG_MM(oacc[i & 3], vfb_n[i], pb[2 + (i >> 2)], oacc[i & 3])                              \
            __builtin_amdgcn_sched_barrier(0);                                                       \
        }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                               \
        }
    if (FAST && TPB == 1) {      // (the launcher sends launches with periodic key multiplicity to the KBP instance)
        const bool rag_ = (Sk & (AG_KVB - 1)) != 0;
        const int fast_end_ = rag_ ? nt - 1 : nt;                 // tiles [1, fast_end_) take the fast form
        bf16x8 kfa_n[8], vfb_n[8];
        {
            const int gs = 0, sub = 0, t = 0;                     // tile 0: the general body, barrier at its end
            if (1 < nt) AG_STAGE(1, 1);
            AG_TILE_BODY
            AG_BARRIER();
        }
        if (2 < nt) AG_STAGE(2, 0);                               // slot 0 is free behind that barrier; from here on tile u + 2 is requested behind the barrier of tile u
        if (1 < fast_end_) AG_LOAD_KFA(1)
#define AG_FAST_TILE_(u_, gs_, KB_)                                                                  \
        {                                                                                            \
            const int u = (u_), gs = (gs_);                                                          \
            AG_FAST_A(KB_)                                                                           \
            AG_BARRIER();                                      /* tile u + 1 has landed (the vmcnt(0) this fence carries) and is visible; every wave is done with slot gs */ \
            if (u + 2 < nt) AG_STAGE(u + 2, gs);                                                     \
            AG_FAST_B(1)      /* always prefetches: behind the last fast tile the fragments are not used (slot gs ^ 1 then holds the ragged last tile or old data; nothing writes it) */ \
        }
        // two tiles per trip: the ring slots are literals, so every LDS address of the loop is a loop-invariant register + an immediate offset (the one-tile loop spent 20 v_add_u32 per
        // tile on them): -7 % SQ cycles, +2.7 ... 3.6 % TF/s at both operating points (profiles/r05_attn_fv_ab*.log, r05_attn_pmc_arms.log).  256 VGPRs; hipcc parks nine
        // loop-invariant dwords that only the epilogue needs in scratch AROUND the loop (40 B; nothing inside the loop touches scratch -- tests/test_asm_hazards_cpu.py checks the listing).
        if constexpr (KBP) {
            // periodic key multiplicity (sequence parallelism: the text tile of every rank's block, tile index % key_bias_period == 0): runs of ordinary tiles go through an inner loop of
            // the fast tile (one per trip, run-time slot), the key-multiplicity tile between two runs through a second copy with the general body's two `scores += bias` lines in it --
            // OUTSIDE the inner loop (both copies inside one loop made hipcc spill in it).  Same bits as the general loop.
            const int per_ = p.key_bias_period;
            int uu = 1;
            while (uu < fast_end_) {
                int nb = uu + (per_ - (tb + uu) % per_) % per_;      // the next key-multiplicity tile at or behind uu
                if (nb > fast_end_) nb = fast_end_;
                if (uu < nb && !(uu & 1)) { AG_FAST_TILE_(uu, 0, false) ++uu; }      // a run is entered on an odd tile: two tiles per trip with literal ring slots, as the plain instance
                for (; uu + 1 < nb; uu += 2) {
                    AG_FAST_TILE_(uu, 1, false)
                    AG_FAST_TILE_(uu + 1, 0, false)
                }
                if (uu < nb) { AG_FAST_TILE_(uu, 1, false) ++uu; }
                if (uu < fast_end_) { AG_FAST_TILE_(uu, uu & 1, true) ++uu; }
            }
        } else
        {
            int uu = 1;
            for (; uu + 1 < fast_end_; uu += 2) {
                AG_FAST_TILE_(uu, 1, false)
                AG_FAST_TILE_(uu + 1, 0, false)
            }
            if (uu < fast_end_) AG_FAST_TILE_(uu, 1, false)
        }
        if (rag_ && nt > 1) {
            const int gs = (nt - 1) & 1, sub = 0, t = nt - 1;     // its tile was requested two tiles ago and retired by the last barrier above (or by tile 0's)
            AG_TILE_BODY
            AG_BARRIER();
        }
    } else
    for (int u = 0; u < ngrp; ++u) {
      const int gs = (u & 1) * TPB;                      // first ring slot of this group
#pragma unroll
      for (int i = 0; i < TPB; ++i)
          if ((u + 1) * TPB + i < nt) AG_STAGE((u + 1) * TPB + i, (gs ^ TPB) + i);
      for (int sub = 0; sub < TPB; ++sub) {
        const int t = u * TPB + sub;
        if (t >= nt) break;
        AG_TILE_BODY
      }
      AG_BARRIER();     // this group fully read by every wave; the next group (DMA) retired by the vmcnt(0) of this fence
    }

    // ---- epilogue: lane (q, h) holds O[q][32db + 8a + 4h + c], r = 4a + c -- 8 bytes of a row per (db, a), the other half-wave the neighbouring 8.
    // (16-byte stores through v_permlane32_swap, nontemporal stores and a static priority for waves 4-7 were measured in round 3 and not adopted -- profiles/r03_attn_variants_v0.log:
    // with 785 key tiles per workgroup the epilogue is ~1 % of a workgroup's life; those arms, the per-workgroup timeline and the fast loop's cost-account arms lived in this file
    // until round 6 and are in the history, commit 299008c.)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + lq;
#define AG_STORE_ROW(dst_)                                                                                   \
    do {                                                                                                     \
        if (qrow < Sq) {                                                                                     \
            bf16_t* const o8_ = (dst_) + 4 * lh;                                                             \
            _Pragma("unroll") for (int db = 0; db < 4; ++db)                                                 \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                  \
                uint2 v;                                                                                     \
                v.x = pack2bf(oacc[db][4 * a + 0] * inv, oacc[db][4 * a + 1] * inv);                         \
                v.y = pack2bf(oacc[db][4 * a + 2] * inv, oacc[db][4 * a + 3] * inv);                         \
                *reinterpret_cast<uint2*>(o8_ + 32 * db + 8 * a) = v;                                        \
            }                                                                                                \
        }                                                                                                    \
    } while (0)
    if (nsp > 1) {
        // partial result of this key range: normalised rows (bf16) + log2-sum-exp; attn_merge_kernel combines the ranges
        const long prow = ((long)item * nsp + split) * 256 + wave * 32 + lq;
        AG_STORE_ROW(p.part_o + prow * 128);
        if (qrow < Sq && lh == 0) p.part_lse[prow] = (PRESC ? m_run : m_run * c2) + __builtin_amdgcn_logf(l_tot);
        return;
    }
    AG_STORE_ROW(p.o + (long)(qrow < Sq ? qrow : 0) * p.o_ss + head * 128);
}

// combine the key ranges of the tail items: out = sum_i 2^(lse_i - M) O_i / sum_i 2^(lse_i - M).  One thread per (query, 8 channels).
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnParams p, int n_items) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_items * 256 * 16;
    if (t >= total) return;
    const int c8 = (int)(t & 15);
    const int ql = (int)((t >> 4) & 255);
    const int item = (int)(t >> 12);
    const int w = p.w_base + item;
    const int head = w / p.nqb, qb = w - head * p.nqb;
    const int qrow = qb * 256 + ql;
    if (qrow >= (p.Sq > 0 ? p.Sq : p.S)) return;
    const int nsp = p.nsplit;
    float M = -INFINITY;
    for (int i = 0; i < nsp; ++i) M = fmaxf(M, p.part_lse[((long)item * nsp + i) * 256 + ql]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, W = 0.f;
    for (int i = 0; i < nsp; ++i) {
        const long prow = ((long)item * nsp + i) * 256 + ql;
        const float wgt = __builtin_amdgcn_exp2f(p.part_lse[prow] - M);
        W += wgt;
        const uint4 v = *reinterpret_cast<const uint4*>(p.part_o + prow * 128 + 8 * c8);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[2 * j] += wgt * __uint_as_float(u[j] << 16);
            acc[2 * j + 1] += wgt * __uint_as_float(u[j] & 0xffff0000u);
        }
    }
    const float inv = 1.0f / W;
    uint4 o;
    o.x = pack2bf(acc[0] * inv, acc[1] * inv); o.y = pack2bf(acc[2] * inv, acc[3] * inv);
    o.z = pack2bf(acc[4] * inv, acc[5] * inv); o.w = pack2bf(acc[6] * inv, acc[7] * inv);
    *reinterpret_cast<uint4*>(p.o + (long)qrow * p.o_ss + head * 128 + 8 * c8) = o;
}

// Tail split, the plan (pure host arithmetic; C ABI: utx_attn_plan): nwg = ceil(Sq / 256) * H workgroups; the r = nwg % ncu workgroups of the last, partly
// filled round are cut along the keys into ns ranges of tps 64-key tiles so that the round costs a fraction of a workgroup's duration instead of a
// whole one (S = 13 824: 1296 workgroups = 5 rounds of 256 CUs + 16 -> the 16 run as 256 sixteenths; S = 50 688: 18 rounds + 144 -> 1008 sevenths;
// sequence parallel over 8 ranks, 3 heads x 198 query blocks = 594 = 2 rounds + 82 -> 246 thirds).  UTX_ATTN_TAILSPLIT=0 disables it.
// out = {nwg, nfull, ns (1 = not split), tps}
extern "C" void utx_attn_split_plan_impl(int H, int Sq, int S, int ncu, int out[4]) {
    const int nqb = ((Sq > 0 ? Sq : S) + 255) / 256;
    const int nwg = nqb * H;
    const int nfull = (nwg / ncu) * ncu, r = nwg - nfull;
    const int nt_all = (S + AG_KVB - 1) / AG_KVB;
    int best_ns = 1;
    if (g_utx_opt.attn_tailsplit != 0 && nfull > 0 && r > 0) {
        double best = 0.92;                                    // worth it only below ~0.9 of a round
        for (int ns = 2; ns <= 16; ++ns) {
            if (r * ns > 2048 || (nt_all + ns - 1) / ns < 12) break;      // workspace bound; >= 12 tiles per split
            const double cost = (double)((r * ns + ncu - 1) / ncu) * (1.0 / ns + 0.02);   // rounds x (share + fixed cost of a workgroup)
            if (cost < best) { best = cost; best_ns = ns; }
        }
    }
    out[0] = nwg; out[1] = nfull; out[2] = 1; out[3] = nt_all;
    if (best_ns > 1) {
        const int tps = (nt_all + best_ns - 1) / best_ns;
        out[2] = (nt_all + tps - 1) / tps;                     // every split owns at least one tile
        out[3] = tps;
    }
}
// bytes of scratch the tail split of this shape needs (0: the launch is never split): normalised partial rows (bf16) + log2-sum-exp (f32); a multiple of 1024
extern "C" size_t utx_attn_split_bytes_impl(int H, int Sq, int S, int ncu) {
    int pl[4];
    utx_attn_split_plan_impl(H, Sq, S, ncu, pl);
    if (pl[2] <= 1) return 0;
    const size_t rows = (size_t)(pl[0] - pl[1]) * pl[2] * 256;
    return rows * 128 * sizeof(bf16_t) + rows * sizeof(float);
}
// bytes of caller-owned scratch of one attention call (C ABI: utx_attn_workspace_bytes): [tail-split scratch | headroom flags of the 4 x 64 kernel (attention_q64.hip), one byte per
// 64-query group].  The 8 x 32 kernel uses the first part only; a call without scratch runs it unsplit.
extern "C" size_t utx_attn_workspace_bytes_impl(int H, int Sq, int S, int ncu) {
    return utx_attn_split_bytes_impl(H, Sq, S, ncu) + utx_attn_q64_flag_bytes(H, Sq, S);
}
// the merge of a split tail round, for the 4 x 64 kernel's launcher (same work items, same partial-row layout)
extern "C" int utx_launch_attn_merge(const AttnParams* t, int n_items, hipStream_t stream) {
    const long mt = (long)n_items * 256 * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((mt + 255) / 256)), dim3(256), 0, stream, *t, n_items);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int PRESC, int TPB, bool BLK = false, bool FAST = false, bool KBP = false>
static int launch_glds(AttnParams p, hipStream_t stream) {
    UTX_ONCE_PER_DEVICE(attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_glds_kernel<PRESC, TPB, BLK, FAST, KBP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, AG_LDS(TPB)) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    p.nqb = ((p.Sq > 0 ? p.Sq : p.S) + 255) / 256;
    p.w_base = 0; p.nsplit = 1; p.tiles_per_split = 0; p.part_o = nullptr; p.part_lse = nullptr;
    int pl[4] = {p.nqb * p.H, 0, 1, 0};
    if (TPB == 1 && !p.flags) utx_attn_split_plan_impl(p.H, p.Sq, p.S, utx_ncu(), pl);
    const int nwg = pl[0], nfull = pl[1], ns = pl[2], tps = pl[3], r = nwg - nfull;
    // the scratch of the split is CALLER-OWNED (utx_attn_fwd_bf16_ws; the legacy entry points pass the context's own buffer, grown outside of any
    // capture): nothing is allocated here, a launch whose scratch is missing or too small runs unsplit -- same result up to one bf16 rounding of the
    // tail rows, a fraction of a round slower
    const size_t rows = (size_t)r * ns * 256;
    if (ns <= 1 || !p.work || p.work_bytes < rows * (128 * sizeof(bf16_t) + sizeof(float))) {
        hipLaunchKernelGGL((attn_fwd_glds_kernel<PRESC, TPB, BLK, FAST, KBP>), dim3(nwg), dim3(512), AG_LDS(TPB), stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    hipLaunchKernelGGL((attn_fwd_glds_kernel<PRESC, TPB, BLK, FAST, KBP>), dim3(nfull), dim3(512), AG_LDS(TPB), stream, p);
    AttnParams t = p;
    t.w_base = nfull; t.nsplit = ns; t.tiles_per_split = tps;
    t.part_o = (bf16_t*)p.work; t.part_lse = (float*)((char*)p.work + rows * 128 * sizeof(bf16_t));
    hipLaunchKernelGGL((attn_fwd_glds_kernel<PRESC, TPB, BLK, FAST, KBP>), dim3(r * ns), dim3(512), AG_LDS(TPB), stream, t);
    const long mt = (long)r * 256 * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((mt + 255) / 256)), dim3(256), 0, stream, t, r);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// UTX_ATTN_TPB: tiles per barrier (ring = 2 groups of TPB tiles): 1 -> 64 KB LDS, 2 -> 128 KB
extern "C" int utx_launch_attn_fwd_glds(const AttnParams* p, int presc, hipStream_t stream) {
    const int tpb = g_utx_opt.attn_tpb;
    if (p->blk_rows > 0) {      // block-strided operands (the sequence-parallel receive buffer): whole 64-key tiles per block, whole blocks per sequence
        if (tpb != 1 || p->flags || (p->blk_rows % AG_KVB) || (p->S % p->blk_rows) || ((p->q_bs | p->k_bs | p->vt_bs) & 7)) return -2;
        return presc ? launch_glds<1, 1, true>(*p, stream) : launch_glds<0, 1, true>(*p, stream);
    }
    if (tpb == 2) return presc ? launch_glds<1, 2>(*p, stream) : launch_glds<0, 2>(*p, stream);
    // the pre-scaled form the DiT uses: the fast loop (FAST above); UTX_ATTN_PEEL=0: the general loop, the default until round 5 (A/B and the reference bits of the stress tests)
    // launches whose key-multiplicity tiles recur (key_bias_period > 0: sequence parallelism): the KBP instance (a loop nest: a second copy of the fast tile for those tiles INSIDE the
    // loop made hipcc spill in it -- 45 scratch accesses per trip with one tile per trip, 293 with two)
    if (presc && g_utx_opt.attn_peel != 0 && (p->key_bias_period > 0 && p->key_bias_log2 != 0.f)) return launch_glds<1, 1, false, true, true>(*p, stream);
    if (presc && g_utx_opt.attn_peel != 0) return launch_glds<1, 1, false, true>(*p, stream);
    return presc ? launch_glds<1, 1>(*p, stream) : launch_glds<0, 1>(*p, stream);
}
