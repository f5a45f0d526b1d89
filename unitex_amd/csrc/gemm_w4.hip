// bf16 MFMA GEMM for the large-M FLUX linears, ONE wave per SIMD (gfx950) -- the default for >= 192 tiles of 256 x 256 (gemm.hip dispatch).
// 256x256 tile, four waves of 128x128, the accumulators of a wave (256 registers) in the AGPR half of its 512-entry register file, every
// K-step's LDS reads and LDS-DMA issued in the shadow of the previous K-step's MFMAs by the SAME wave; persistent workgroups (one per CU)
// walk tiles w, w + G, ... and the DMA cursor runs on across tile boundaries.
//
//   C[M,N] = epi( alpha * ( A[M,K] . B[N,K]^T  +  A2[M,K2] . B2[N,K2]^T ) + bias[N] )      (same contract as gemm.hip; bit-identical outputs)
//
// Why this structure (profiles/r02_gemm_clock_probe.log): on M = 50688, N = 21504, K = 3072 the vendor library's hand-scheduled 4-wave kernel
// keeps the matrix pipe 81 % busy against 70 % for the 8-wave kernel of gemm_pers.hip at identical MFMA work -- there a compute wave and a load
// wave alternate per SIMD behind two barriers per phase and the load phase (~350 cycles) outlasts the 256-cycle MFMA phase it should hide
// behind.  With one wave per SIMD there is no phase to balance: a wave owns 128x128 (a third fewer fragment bytes per FLOP: 32 ds_read_b128 per
// 64 MFMAs instead of 24 per 32), reads the fragments of K-step k+1 under the MFMAs of K-step k and meets the other three waves at ONE barrier
// per 64 MFMAs.  What the measurements say (profiles/r02_gemm_w4_*.log, r02_l2_fill_probe.log):
//   * the instruction stream has no bubble of its own: with the DMA cursor parked (every DMA re-reads one KB) a 64-k K-tile costs 0.97 us = the
//     matrix pipe's 2048 cycles at 2.1 GHz;
//   * with real operand traffic the kernel is bound by how the L2 -> LDS path is fed.  A first version staged 32-k sub-stages (64-byte row
//     segments, ring of four): tools/l2_fill_probe.hip shows that path delivers 34-44 GB/s per CU for 64-byte segments and 87-100 for 128-byte
//     ones with every CU streaming; that version ran at exactly 35 GB/s per CU (1.48 us per K-tile, 0.94-0.99 x the 8-wave kernel).  This
//     version stages 64-k K-tiles on 128-byte rows (two 64 KB stages) -- 1.33 us per K-tile;
//   * a burst of DMAs backs the vector-memory path up into the issuing wave, which has no partner wave to hide behind: eight in a row cost
//     ~30 cycles each beyond their MFMA shadow; one per three MFMAs (below) is worth +3-4 % over one per MFMA;
//   * the epilogue (8.5k cycles per tile: 256 v_accvgpr_read + fma + pack + LDS transpose) is exposed -- nothing overlaps it with one wave
//     per SIMD: bias through ONE scalar-load burst, accumulators re-zeroed by 16 MFMAs (0 x 0 + 0), C stores coalesced through LDS.
//   * the C stores carry the nontemporal hint: one round of tiles writes 4 MB of C per XCD, the size of its L2, and write-allocated C lines evicted the
//     operand panels the next K-tiles stream (-6 % per K = 3072 tile, -22 % at K = 1024: profiles/r02_gemm_w4_probe_nt.log; W4_STORE_U);
//   * whole rounds of 256 tiles are the unit of time of a persistent kernel: 636 tiles (the reference strip's N = 3072 linears) are 2.48 rounds and
//     cost 3.  The last, partly filled round is cut along K instead ("split tail" below, gemm_w4_fixup_kernel): -13 / -14 % on the K = 12288 /
//     15360 linears of the reference strip, -7 % on BASELINE's strip, -35 % on the pruned last block (profiles/r02_gemm_streamk_check_v4.log).
// Result: +8-11 % over gemm256_pers_kernel on the FLUX shapes, 2-13 % behind the vendor kernel (profiles/r02_gemm_w4_check_v8.log); bench A/B
// profiles/r02_bench_gemm_w4_ab.log.  Replacing every 32x32x16 MFMA by two 16x16x32 (the vendor kernel's shape; ablation) is worth 1-4 % more.
//
// Stream: K-tile = 64 k of the 256x256 tile = A[256][64] + B[256][64] bf16 = 64 KB, two stages in LDS (128 KB) + 8 KB of C staging per wave
// = all 160 KB.  Per K-tile and wave: 4 K-steps x 16 MFMA 32x32x16, 32 ds_read_b128, 16 LDS-DMA pieces of 1 KB (8 rows x 128 B).
//   K-step 0: MFMAs on F0 | reads -> F1 (kk 1) | pieces 6..10 of the cursor's K-tile (= compute K-tile + 1)
//   K-step 1: MFMAs on F1 | reads -> F0 (kk 2) | pieces 11..15, cursor advances
//   K-step 2: MFMAs on F0 | reads -> F1 (kk 3); s_waitcnt lgkmcnt(0) [every read of this stage retired]; vmcnt(0) [K-tile + 1 landed]; s_barrier
//   K-step 3: MFMAs on F1 | reads -> F0 (kk 0 of the NEXT stage) | pieces 0..5 of K-tile + 2 into THIS stage (free since the barrier)
// Hazards: RAW -- a stage is read only after every wave's vmcnt(0) for it and the barrier behind that; WAR -- the pieces of K-tile + 2 overwrite
// this K-tile's stage, whose last reads (K-step 3's fragments, read in K-step 2) every wave retired before that barrier.  Loads, DMAs and stores
// retire in order through one counter, so vmcnt(0) also waits for the C stores of an epilogue just behind; they have had a K-tile to retire.
// The MFMAs are inline asm (AGPR accumulators by constraint): hipcc's hazard recogniser does not see them -- the two places where a VALU result
// feeds an asm MFMA or an asm MFMA result feeds a VALU read carry their own s_nop (W4_ZERO_ACC, W4_MFMA_DRAIN).
//
// LDS stage layout: operand X at X * 32 KB, row r (128 B) at r * 128, its eight 16-byte chunks XOR-swizzled: slot = chunk ^ ((r >> 1) & 7)
// (conflict-free for the ds_read_b128 lane groups, as in gemm.hip); applied on the DMA source (which global chunk a lane fetches; the LDS image
// of a DMA is lane-linear) and on the fragment read address.
#include "common.h"
#include "kernels.h"
#include "gemm_w4_loop_asm.inc"      // GENERATED (tools/gen_gemm_w4_loop.py): the steady-state K loop as one hand-placed stream, two K-tiles per trip

#ifndef W4_ZERO_BY_MFMA
#define W4_ZERO_BY_MFMA 1
#endif
#define W4_STAGE 65536      // one K-tile of 64 k: A[256][64] at +0, B[256][64] at +32768, bf16, 128-byte rows

__device__ __forceinline__ float w4_gelu_tanh(float x) {   // same expression as gemm.hip
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

// one LDS-DMA: 64 lanes x 16 B from (sbase + voff) to the lane-linear 1 KB at LDS byte address lds_addr
__device__ __forceinline__ void w4_dma(unsigned lds_addr, unsigned voff, const char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
// the same with M0 already holding the LDS address (set in front of the preceding MFMA, which is the wait state M0 needs)
__device__ __forceinline__ void w4_dma_m0(unsigned voff, const char* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase) : "memory");
}

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)

// a wave-uniform pointer the compiler computed with vector instructions (it has no scalar 64-bit multiply) back into SGPRs: the DMA takes
// its base as an "s" operand
__device__ __forceinline__ const char* w4_uniform(const char* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
}

// (The timing arms of this kernel -- no DMA / no fragment reads / no barrier / no epilogue / parked cursor / the 16x16x32 MFMA-shape arm / the per-workgroup timeline / the
// start stagger / plain C stores / the MX DMA placements -- lived here until round 6 and are in the history, commit 299008c; their measurements are quoted where they decided something.)
// QKF: the fused q / k post-processing of utx_gemm_desc.qk_cols (plain kernel only)
// MX: OCP MX fp8 operands with tile-packed E8M0 scales (utx_gemm_desc.mx8 == 2; "MX fp8" below): same staging, ring and epilogues, a K-tile is
// 128 fp8 = the same 128-byte rows, 32 v_mfma_scale_f32_32x32x64_f8f6f4 per K-tile instead of 64 v_mfma_f32_32x32x16_bf16
// FK (round 6; bf16 only): inside a K segment the K-tiles run through the generated stream (W4F_ASM_TEXT) -- same schedule, same MFMAs in the same order per accumulator, 74 instead of
// 250 instructions beside a K-tile's 64 MFMAs; every boundary (tile, segment, LoRA switch, split-tail range, parking) stays with the loop below.  Bit-identical to FK = false.
template <bool GATED, bool QKF = false, bool MX = false, bool FK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256_w4_kernel(GemmParams p, int ntiles, int reserved /* (the ablation arms' timeline workgroup until round 6; kept so that the kernel-argument layout -- and with it the register allocation of the five
      instances -- is byte-for-byte what was measured: the listings before and after the arms left this file are instruction-identical) */, int sk_T, int sk_S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    const int wid = xcd_remap(blockIdx.x, G);
    const int ntn = p.ntn & 0xffff, group_m = (p.ntn >> 16) & 0xff;
    const int ntm_ = (p.M + 255) / 256;
    const int per_group = group_m * ntn;
    const int nss1 = p.K / 64, nss2 = p.K2 / 64;      // K-tiles of the base / LoRA segment
    const unsigned ldaB = (unsigned)p.lda * 2, ldbB = (unsigned)p.ldb * 2, lda2B = (unsigned)p.lda2 * 2, ldb2B = (unsigned)p.ldb2 * 2;
    // ---- split tail (sk_T > 0; launcher: utx_launch_gemm_w4).  The ntiles - sk_T tiles of the whole rounds are walked as before; each of the sk_T
    // tiles of the last, partly filled round is cut along K into sk_S equal ranges: range r = 0 .. sk_T sk_S - 1 is part r / sk_T of tail tile
    // r % sk_T, and workgroup w takes ranges w, w + G, ... after its whole tiles (one range each when sk_T sk_S <= G, the usual case).  Workgroups
    // with neighbouring indices (one XCD) then work on neighbouring tiles at the SAME K offsets -- the lockstep that lets them share operand
    // panels in L2 during a whole round (a first version cut the tail's K-tiles into G exactly equal ranges that straddled tiles: every
    // workgroup at its own K offset, and a third of the gain gone).  A range ends with its fp32 accumulators written as they lie in the
    // registers to slot r of p.sk_work (W4_DUMP); gemm_w4_fixup_kernel sums a tile's sk_S slots in K order and runs the epilogue.
    // nssu = K-tiles per tail tile, the same for every tail tile (the launcher checks).
    const int nfull_tiles = ntiles - sk_T;
    const int nssu = nss1 + ((p.K2 > 0 && p.lora_n_limit > 0) ? nss2 : 0);
    // the segment after the current one of a cursor (tj_, tile_): the next whole tile of this workgroup, then its tail ranges.  Scalar work only.
    // tj_: -1 whole tiles, else the number of tail ranges taken so far.  k1_ < 0 stands for "the whole tile" (its K extent depends on its columns)
#define W4_NEXT_SEG(tj_, tile_, ok_, k0_, k1_)                                                     \
    do {                                                                                           \
        ok_ = false; k0_ = 0; k1_ = -1;                                                            \
        if (tj_ < 0) {                                                                             \
            tile_ += G;                                                                            \
            if (tile_ < nfull_tiles) ok_ = true; else tj_ = 0;                                     \
        }                                                                                          \
        if (!ok_) {                                                                                \
            const int r_ = wid + tj_ * G;                                                          \
            if (r_ < sk_T * sk_S) {                                                                \
                const int j_ = r_ / sk_T;                                                          \
                ok_ = true; tile_ = nfull_tiles + r_ - j_ * sk_T; ++tj_;                           \
                k0_ = (j_ * nssu) / sk_S; k1_ = ((j_ + 1) * nssu) / sk_S;                          \
            }                                                                                      \
        }                                                                                          \
    } while (0)

#define W4_TILE_ORIGIN(w_, m0_, n0_)                                                       \
    do {                                                                                   \
        const int grp_ = (w_) / per_group, rem_ = (w_) - grp_ * per_group;                 \
        const int ftm_ = grp_ * group_m;                                                   \
        const int gs_ = (ntm_ - ftm_ < group_m) ? ntm_ - ftm_ : group_m;                   \
        const int tn_ = rem_ / gs_;                                                        \
        (m0_) = (ftm_ + rem_ - tn_ * gs_) * 256;                                           \
        (n0_) = tn_ * 256;                                                                 \
    } while (0)

    // ---- staging cursor.  A K-tile (64 k) of an operand = 256 rows x 128 B = 32 DMA pieces of 8 rows; wave w issues pieces w, w + 4, ...,
    // w + 28: lane -> row lane >> 3 of the piece, LDS chunk lane & 7 <- global chunk (lane & 7) ^ ((row >> 1) & 7) (the swizzle of gemm.hip's
    // 128-byte rows; (row >> 1) & 7 = (4 (piece & 1) + (lane >> 4)) & 7 and piece & 1 = wave & 1).  FULL 128-byte row segments per request:
    // tools/l2_fill_probe.hip measures 87-100 GB/s per CU for this pattern with every CU streaming from L2 and 34-44 for 64-byte segments
    // (the first version of this kernel staged 32-k sub-stages and was bound by exactly that: profiles/r02_l2_fill_probe.log).
    // One scalar base per operand (advanced 128 B per K-tile), eight per-lane byte offsets per operand (recomputed per tile / K-segment:
    // strides change with the LoRA segment, rows >= M re-read row M-1).
    const int drow = lane >> 3;
    const unsigned dchunk = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (drow >> 1)) & 7)) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    int s_tile = wid, s_ss = 0, s_seg_end = 0, s_seg = 1, s_m0 = 0, s_n0 = 0, s_slot = 0;
    int s_l2 = 0;       // K-tiles of the LoRA segment that follow the base segment of the cursor's tile (0: none)
    int s_tj = -1;      // W4_NEXT_SEG
    const char *s_pA = nullptr, *s_pB = nullptr;
    // MX fp8: the E8M0 scales of the cursor's K-tile, tile-packed [K/128][row blocks of 128][l31 = 32][im = 4] dwords (4 scale bytes = the four
    // 32-element blocks of the K-tile of row 128 rb + 32 im + l31): ONE 16-byte load per lane, operand and K-tile gives a lane the scale dwords of its
    // four fragment rows (512 contiguous bytes per wave; lanes 32-63 fetch the same bytes as lanes 0-31).  p.lds_a / p.lds_b = row blocks per K-tile slab.
    const char *s_pSA = nullptr, *s_pSB = nullptr;
    unsigned s_sstA = 0, s_sstB = 0;
    unsigned voA0 = 0, voA1 = 0, voA2 = 0, voA3 = 0, voA4 = 0, voA5 = 0, voA6 = 0, voA7 = 0;
    unsigned voB0 = 0, voB1 = 0, voB2 = 0, voB3 = 0, voB4 = 0, voB5 = 0, voB6 = 0, voB7 = 0;
#define W4_ROWOFF_A(d_, stride_) ((unsigned)(((s_m0 + 8 * (wave + 4 * (d_)) + drow > p.M - 1) ? p.M - 1 - s_m0 : 8 * (wave + 4 * (d_)) + drow)) * (stride_) + dchunk)
#define W4_ROWOFF_B(d_, stride_) ((unsigned)(8 * (wave + 4 * (d_)) + drow) * (stride_) + dchunk)
#define W4_SET_OFFS(sa_, sb_)                                                                              \
    do {                                                                                                   \
        voA0 = W4_ROWOFF_A(0, sa_); voA1 = W4_ROWOFF_A(1, sa_); voA2 = W4_ROWOFF_A(2, sa_); voA3 = W4_ROWOFF_A(3, sa_); \
        voA4 = W4_ROWOFF_A(4, sa_); voA5 = W4_ROWOFF_A(5, sa_); voA6 = W4_ROWOFF_A(6, sa_); voA7 = W4_ROWOFF_A(7, sa_); \
        voB0 = W4_ROWOFF_B(0, sb_); voB1 = W4_ROWOFF_B(1, sb_); voB2 = W4_ROWOFF_B(2, sb_); voB3 = W4_ROWOFF_B(3, sb_); \
        voB4 = W4_ROWOFF_B(4, sb_); voB5 = W4_ROWOFF_B(5, sb_); voB6 = W4_ROWOFF_B(6, sb_); voB7 = W4_ROWOFF_B(7, sb_); \
    } while (0)
    // ONE piece of code places the cursor: K-tile kk_ of [.., k1_) of tile s_tile (K-tiles 0 .. nss1 - 1 = base segment, then the LoRA segment) --
    // the start of a tile, the switch to its LoRA segment, the start of a tail segment and parking all come through here, so the sixteen
    // per-lane offsets have one writer inside the loop (several writers under runtime branches cost registers in the K loop).
    // park_: past its last segment the cursor keeps issuing (the loop has no conditional DMA): strides 0 -> every DMA re-reads the first 128 B
    // of row 0 of A and B into ring slots nobody reads any more.
#define W4_STAGE_AT(kk_, k1_, park_)                                                                       \
    do {                                                                                                   \
        W4_TILE_ORIGIN(s_tile, s_m0, s_n0);                                                                \
        const int k1r_ = (k1_) < 0 ? nss1 + (((p.K2 > 0) && (s_n0 < p.lora_n_limit)) ? nss2 : 0) : (k1_); \
        const bool in2_ = (kk_) >= nss1;                                                                   \
        const int kk2_ = in2_ ? (kk_) - nss1 : (kk_);                                                      \
        const unsigned sa_ = (park_) ? 0u : in2_ ? lda2B : ldaB, sb_ = (park_) ? 0u : in2_ ? ldb2B : ldbB; \
        const char* const a_ = in2_ ? (const char*)p.A2 + (long)((s_n0 / p.lora_seg_n) * p.K2) * 2 : (const char*)p.A; \
        const char* const b_ = in2_ ? (const char*)p.B2 : (const char*)p.B;                                \
        s_pA = w4_uniform(a_ + (long)s_m0 * sa_ + (long)kk2_ * 128);                                       \
        s_pB = w4_uniform(b_ + (long)s_n0 * sb_ + (long)kk2_ * 128);                                       \
        if constexpr (MX) {                                                                                \
            const int rba_ = ((s_m0 >> 7) + wm < (int)p.lds_a) ? (s_m0 >> 7) + wm : (int)p.lds_a - 1;     /* a wave half beyond M: any block (rows never stored) */ \
            const int rbb_ = (s_n0 >> 7) + wn;                                                             \
            s_sstA = (park_) ? 0u : (unsigned)p.lds_a * 512u; s_sstB = (park_) ? 0u : (unsigned)p.lds_b * 512u; \
            s_pSA = w4_uniform((const char*)p.a_scale + ((long)kk2_ * p.lds_a + rba_) * 512);              \
            s_pSB = w4_uniform((const char*)p.b_scale + ((long)kk2_ * p.lds_b + rbb_) * 512);              \
        }                                                                                                  \
        W4_SET_OFFS(sa_, sb_);                                                                             \
        s_ss = kk2_; s_seg = in2_ ? 2 : 1;                                                                 \
        s_seg_end = (park_) ? 0x7fffffff : in2_ ? k1r_ - nss1 : (k1r_ < nss1 ? k1r_ : nss1);               \
        s_l2 = in2_ ? 0 : (k1r_ > nss1 ? k1r_ - nss1 : 0);                                                 \
    } while (0)
#define W4_SVALID (s_tile < ntiles)
    // DMA d_ (0..7, literal) of operand A (isb_ = 0) / B (1) of the cursor's K-tile
#define W4_VO(isb_, d_) ((isb_) ? ((d_) == 0 ? voB0 : (d_) == 1 ? voB1 : (d_) == 2 ? voB2 : (d_) == 3 ? voB3 : (d_) == 4 ? voB4 : (d_) == 5 ? voB5 : (d_) == 6 ? voB6 : voB7) \
                                : ((d_) == 0 ? voA0 : (d_) == 1 ? voA1 : (d_) == 2 ? voA2 : (d_) == 3 ? voA3 : (d_) == 4 ? voA4 : (d_) == 5 ? voA5 : (d_) == 6 ? voA6 : voA7))
#define W4_DMA_LDS(isb_, d_) (lds0 + (unsigned)s_slot * W4_STAGE + (isb_) * 32768u + (unsigned)(wave + 4 * (d_)) * 1024u)
#define W4_DMA_M0(isb_, d_) w4_dma_m0(W4_VO(isb_, d_), (isb_) ? s_pB : s_pA)
#define W4_DMA(isb_, d_) w4_dma(W4_DMA_LDS(isb_, d_), W4_VO(isb_, d_), (isb_) ? s_pB : s_pA)
#define W4_STAGE_ADVANCE()                                                                 \
    do {                                                                                   \
        s_pA += 128; s_pB += 128;                                                          \
        if constexpr (MX) { s_pSA += s_sstA; s_pSB += s_sstB; }                            \
        s_slot ^= 1;                                                                       \
        ++s_ss;                                                                            \
        if (s_ss == s_seg_end) {                                                           \
            bool ok_; int kk_, k1_;                                                        \
            if (s_seg == 1 && s_l2 > 0) { ok_ = true; kk_ = nss1; k1_ = nss1 + s_l2; }     \
            else W4_NEXT_SEG(s_tj, s_tile, ok_, kk_, k1_);                                 \
            if (!ok_) { s_tile = 0; kk_ = 0; k1_ = 1; }                                    \
            W4_STAGE_AT(kk_, k1_, !ok_);                                                   \
        }                                                                                  \
    } while (0)
    // ---- compute cursor
    int c_tile = wid, c_ss = 0, c_nss = 0, c_m0 = 0, c_n0 = 0, c_slot = 0;
    int c_tj = -1;      // as s_tj; > 0 while the cursor is in a tail range (its c_tj-th)
#define W4_COMPUTE_AT(k0_, k1_)                                                            \
    do {                                                                                   \
        W4_TILE_ORIGIN(c_tile, c_m0, c_n0);                                                \
        c_nss = (k1_) < 0 ? nss1 + (((p.K2 > 0) && (c_n0 < p.lora_n_limit)) ? nss2 : 0) : (k1_) - (k0_); \
        c_ss = 0;                                                                          \
    } while (0)
#define W4_COMPUTE_NEXT(done_)                                                             \
    do {                                                                                   \
        bool ok_; int k0_, k1_;                                                            \
        W4_NEXT_SEG(c_tj, c_tile, ok_, k0_, k1_);                                          \
        (done_) = !ok_;                                                                    \
        if (ok_) W4_COMPUTE_AT(k0_, k1_);                                                  \
    } while (0)

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[4][4];   // [jn][im], swapped MFMA: rows = n, cols = m
    typedef __attribute__((ext_vector_type(4))) float w4_f32x4;
    typedef __attribute__((ext_vector_type(4))) unsigned int w4_u32x4s;
    // every use of the accumulators is an "a"-constrained asm operand (MFMA, zeroing, the epilogue's reads): the register class of the
    // tile is then AGPR by construction and the allocator has nothing to split
#define W4_ZERO_ACC()                                                       \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                        \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                     \
        if (W4_ZERO_BY_MFMA && r_ == 0) asm volatile("s_nop 2\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[a_][b_]) : "v"(zero8));   /* 0 x 0 + 0: one MFMA zeroes 16 registers (32 cycles; sixteen v_accvgpr_write take 80).  s_nop: hipcc may materialise the zero operand with a v_mov right in front of the asm and does not see the MFMA inside it (VALU write -> MFMA read needs wait states) */ \
        else if (!W4_ZERO_BY_MFMA) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc[a_][b_][r_]));    \
    }
#define W4_ACC(jn_, im_, r_) ({ float x_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x_) : "a"(acc[jn_][im_][r_])); x_; })
    W4_ZERO_ACC()

    // fragment read offsets inside a stage: row * 128 + ((2 kk + lh) ^ swz) * 16, swz = (row >> 1) & 7 = (l31 >> 1) & 7 (row blocks are multiples of 32)
    const int fsw = (l31 >> 1) & 7;
    const int xk0 = ((0 + lh) ^ fsw) << 4, xk1 = ((2 + lh) ^ fsw) << 4, xk2 = ((4 + lh) ^ fsw) << 4, xk3 = ((6 + lh) ^ fsw) << 4;
    const int arow = (wm * 128 + l31) * 128, brow = 32768 + (wn * 128 + l31) * 128;
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
#define W4_LD(off_) (*reinterpret_cast<const bf16x8*>(smem + (off_)))
    // read number r_ (0..7) of a K-step: B0 A0 A1 B1 A2 A3 B2 B3 -- the first MFMAs of the next K-step need B0, A0, A1 first
#define W4_READ(r_, FA_, FB_, base_, xk_)                                                                  \
    do {                                                                                                   \
        if ((r_) == 0) FB_[0] = W4_LD((base_) + brow + (xk_));                                             \
        if ((r_) == 1) FA_[0] = W4_LD((base_) + arow + (xk_));                                             \
        if ((r_) == 2) FA_[1] = W4_LD((base_) + arow + 4096 + (xk_));                                      \
        if ((r_) == 3) FB_[1] = W4_LD((base_) + brow + 4096 + (xk_));                                      \
        if ((r_) == 4) FA_[2] = W4_LD((base_) + arow + 8192 + (xk_));                                      \
        if ((r_) == 5) FA_[3] = W4_LD((base_) + arow + 12288 + (xk_));                                     \
        if ((r_) == 6) FB_[2] = W4_LD((base_) + brow + 8192 + (xk_));                                      \
        if ((r_) == 7) FB_[3] = W4_LD((base_) + brow + 12288 + (xk_));                                     \
    } while (0)
    // MFMA number i_ (0..15) of a K-step: (jn, im) in the order that touches the fragments as they arrive
#define W4_MF(i_, FA_, FB_)                                                                                \
    do {                                                                                                   \
        constexpr int jn_ = ((i_) >> 2), im_ = ((i_) & 3);                                                 \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_])); \
    } while (0)
#define W4_MF_M0(i_, FA_, FB_, m0_)                                                                        \
    do {                                                                                                   \
        constexpr int jn_ = ((i_) >> 2), im_ = ((i_) & 3);                                                 \
        asm volatile("s_mov_b32 m0, %3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_]), "s"(m0_)); \
    } while (0)
    // ---- MX fp8 (MX): fragments of 32 bytes per lane and MFMA -- lane (row, h) holds k = 16h .. 16h+15 of the MFMA's first 32-element scale block
    // in bytes 0-15 and k = 32 + 16h .. of its second block in bytes 16-31 (tools/mx_probe.hip), i.e. chunks 4 ks + h and 4 ks + 2 + h of the
    // 128-byte row for K-step ks = 0, 1: the same four swizzled chunk offsets xk0..xk3 as the bf16 form, two ds_read_b128 per fragment.  Block 0's scale
    // is taken from lane `row`, block 1's from lane `row + 32`, byte op_sel of the scale register: the lane's dword of four scale bytes is
    // pre-shifted by 8 h, K-step ks reads byte 2 ks (op_sel_hi, the instruction's second op_sel bit).
    typedef __attribute__((ext_vector_type(8))) int w4_i32x8;
    typedef __attribute__((ext_vector_type(4))) int w4_i32x4;
    w4_i32x8 xa0[4], xb0[4], xa1[4], xb1[4];
    w4_u32x4s sa_cur = {0u, 0u, 0u, 0u}, sb_cur = {0u, 0u, 0u, 0u}, sa_nxt = {0u, 0u, 0u, 0u}, sb_nxt = {0u, 0u, 0u, 0u};
    const unsigned svo = (unsigned)l31 * 16u;
    const unsigned lds_w = lds0 + (unsigned)wave * 1024u;
#define W4_LD4(off_) (*reinterpret_cast<const w4_i32x4*>(smem + (off_)))
    // read number r_ (0..15) of an MX K-step: fragment r_ / 2 in the order B0 A0 A1 B1 A2 A3 B2 B3, low half then high half
#define W4_XREAD(r_, FA_, FB_, base_, xlo_, xhi_)                                                          \
    do {                                                                                                   \
        constexpr int f_ = (r_) >> 1;                                                                      \
        const int xo_ = ((r_) & 1) ? (xhi_) : (xlo_);                                                      \
        constexpr int isb_ = (f_ == 0 || f_ == 3 || f_ == 6 || f_ == 7) ? 1 : 0;                           \
        constexpr int ix_ = f_ == 0 ? 0 : f_ == 1 ? 0 : f_ == 2 ? 1 : f_ == 3 ? 1 : f_ == 4 ? 2 : f_ == 5 ? 3 : f_ == 6 ? 2 : 3; \
        if constexpr (isb_) { if constexpr (((r_) & 1) == 0) FB_[ix_].lo = W4_LD4((base_) + brow + ix_ * 4096 + xo_); else FB_[ix_].hi = W4_LD4((base_) + brow + ix_ * 4096 + xo_); } \
        else                { if constexpr (((r_) & 1) == 0) FA_[ix_].lo = W4_LD4((base_) + arow + ix_ * 4096 + xo_); else FA_[ix_].hi = W4_LD4((base_) + arow + ix_ * 4096 + xo_); } \
    } while (0)
    // MFMA number i_ (0..15) of MX K-step ks_ (literal 0 / 1): scale byte 2 ks_
#define W4_XMF(i_, ks_, FA_, FB_)                                                                          \
    do {                                                                                                   \
        constexpr int jn_ = ((i_) >> 2), im_ = ((i_) & 3);                                                 \
        if constexpr ((ks_) == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"   \
                                               : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_]), "v"(sb_cur[jn_]), "v"(sa_cur[im_])); \
        else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]"      \
                          : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_]), "v"(sb_cur[jn_]), "v"(sa_cur[im_])); \
    } while (0)
#define W4_XMF_M0(i_, ks_, FA_, FB_, m0_)                                                                  \
    do {                                                                                                   \
        constexpr int jn_ = ((i_) >> 2), im_ = ((i_) & 3);                                                 \
        if constexpr ((ks_) == 0) asm volatile("s_mov_b32 m0, %5\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" \
                                               : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_]), "v"(sb_cur[jn_]), "v"(sa_cur[im_]), "s"(m0_)); \
        else asm volatile("s_mov_b32 m0, %5\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]" \
                          : "+a"(acc[jn_][im_]) : "v"(FB_[jn_]), "v"(FA_[im_]), "v"(sb_cur[jn_]), "v"(sa_cur[im_]), "s"(m0_)); \
    } while (0)
    // the scale dwords of the staging cursor's K-tile (a load hipcc does not see: its data is first touched behind the counted wait that names it)
#define W4_XSCALE_LOAD(dst_, base_) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(svo), "s"(base_) : "memory")
    // next -> current: the lane's byte lane: lanes 32-63 supply the scale of the second block of every MFMA.  s_nop: the MFMAs behind it are asm
    // (VALU write -> MFMA read needs wait states hipcc cannot insert)
#define W4_XSCALE_TAKE()                                                                                   \
    do {                                                                                                   \
        const unsigned sh_ = 8u * (unsigned)lh;                                                            \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { sa_cur[e_] = sa_nxt[e_] >> sh_; sb_cur[e_] = sb_nxt[e_] >> sh_; } \
        asm volatile("s_nop 4" : "+v"(sa_cur), "+v"(sb_cur));                                              \
    } while (0)
    // The MFMAs are inline asm so that the 256 accumulator registers are AGPRs by constraint (left to itself hipcc keeps part of the
    // accumulator tile in VGPRs and shuttles it through v_accvgpr_write around every MFMA, spilling the fragments); the price is that
    // its hazard recogniser does not see them: the only dependent non-MFMA reads are the epilogue's v_accvgpr_read, behind W4_MFMA_DRAIN.
#define W4_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

    // ---- epilogue of the compute cursor's tile (plain: bias / GELU / column split; gated: residual + gate * y)
    // lane (l31, lh): acc[jn][im][4a+c] -> n = n0 + wn*128 + jn*32 + 8a + 4lh + c ;  m = m0 + wm*128 + im*32 + l31.
    // Packed pairs of the groups a = 2q / 2q+1 are exchanged between the half-waves (v_permlane32_swap): lanes 0-31 then hold columns
    // 16q .. 16q+7, lanes 32-63 columns 16q+8 .. 16q+15 of their row -> one 16-byte store.  Bias / gate of the wave's 128 columns come
    // through the scalar cache (uniform address): no vmcnt traffic, nothing resident during the K loop.
    const bf16_t* const pres = (const bf16_t*)p.res;
    typedef __attribute__((ext_vector_type(16))) unsigned int w4_u32x16;
    typedef __attribute__((ext_vector_type(4))) unsigned int w4_u32x4;
    // 16 dwords (32 bf16 = the columns of one jn block) through the scalar cache; hipcc would use vector loads (it cannot prove
    // the array is not written by this kernel), whose vmcnt waits would drain the staging pipeline
#define W4_SLOAD16(dst_, ptr_) asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst_) : "s"(ptr_) : "memory")
    // the bias of the wave's 128 columns (4 x 16 dwords) in ONE request burst and one wait per tile: sixteen loads waited for one by one (one per
    // 32-row block and jn) cost ~3k of the epilogue's 10k cycles (tools/gemm_w4_trace.py)
#define W4_SLOAD64(d0_, d1_, d2_, d3_, ptr_)                                                                           \
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\t" \
                 "s_waitcnt lgkmcnt(0)" : "=&s"(d0_), "=&s"(d1_), "=&s"(d2_), "=&s"(d3_) : "s"(ptr_) : "memory")
    // C leaves through a per-wave 8 KB LDS buffer (the 32 KB of LDS above the ring), one 32-row block at a time: in accumulator layout
    // a lane owns 16 bytes of ONE row per store and a store instruction touches 32 rows x 32 B -- measured 10-12 us per tile against
    // 3 us when every store instruction writes 4 rows x 256 contiguous bytes (profiles/r02_gemm_w4_probe_v4.log).  Written [32 rows]
    // [16 chunks of 16 B] with chunk ^= row & 15 (ds_write_b128 lane groups are 8 consecutive lanes = 8 rows of one chunk column),
    // read back as lane -> (row 4t + lane/16, chunk lane%16): both conflict-free.  The residual of the gated epilogue is read in the
    // same lane -> (row, chunk) layout, so its loads are coalesced the same way.
    char* const stg = smem + 2 * W4_STAGE + wave * 8192;
    // residual / gate loads hipcc does not see (a visible load would be waited for with vmcnt(0): a drain of the staging pipeline)
#define W4_LOAD16_ASM(dst_, ptr_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
// Counted waits for registers an asm load fills.  THE WAIT NAMES NO REGISTER: with a tied operand ("+v"(r_) on the s_waitcnt itself) the compiler is free to
// give the asm's input and output different registers and to put the COPY between them IN FRONT of the asm -- a v_mov that reads the load's destination
// while the load is still in flight.  It did exactly that in the ragged-tile branch of the fused q / k epilogue (round 3's one-off one-ulp difference:
// v_mov_b64 of the cos / sin registers ahead of `s_waitcnt vmcnt(0)`, right whenever the tables happened to have landed -- tools/vmcnt_hazard_check.py finds
// it in the listing).  Here the wait is followed by a scheduling fence and only then by an empty asm that re-defines the registers (W4_LANDED): a copy the
// compiler makes for that asm sits behind the wait.  tests/test_asm_hazards_cpu.py runs the checker over every kernel of the build.
#define W4_WAIT_VM(n_) do { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n_) : "memory"); W4_FENCE(); } while (0)
#define W4_LANDED(r_) asm volatile("" : "+v"(r_))
#define W4_LANDED2(r0_, r1_) asm volatile("" : "+v"(r0_), "+v"(r1_))
#define W4_WAIT_RES(n_, r_) do { W4_WAIT_VM(n_); W4_LANDED(r_); W4_FENCE(); } while (0)
    // gate of this lane's 8 output columns (store layout); fused q / k tiles (QKF): the RMSNorm weight of this wave's head (q or k), eight
    // channels per lane in store layout.  Requested at the START OF THE EPILOGUE by a load hipcc does not see, and older than every residual /
    // cos-sin load behind it: the first counted wait of the epilogue covers it (in-order retirement).  It must not be requested earlier (say at
    // the tile's start, its latency hidden by the K loop): the compiler believes the register is defined when the asm statement ends and is
    // free to copy it -- across the loop's back edge it did, before the data had landed.
#define W4_GATE_FETCH()                                                                                                        \
    do {                                                                                                                       \
        if (GATED) W4_LOAD16_ASM(gq, (const bf16_t*)p.gate + c_n0 + wn * 128 + 8 * (lane & 15));                                \
        if (QKF && c_n0 < p.qk_cols)                                                                                           \
            W4_LOAD16_ASM(gq, (const bf16_t*)((c_n0 + wn * 128) * 2 >= p.qk_cols ? p.qk_wk : p.qk_wq) + 8 * (lane & 15));      \
    } while (0)

    // y of one 32-row block im_ -> staging buffer
#define W4_EPI_WRITE(im_, GELU_)                                                                                       \
        _Pragma("unroll") for (int jn_ = 0; jn_ < 4; ++jn_) {                                                          \
            w4_u32x16 bw_;                                                                                             \
            if constexpr (GATED) W4_SLOAD16(bw_, ebias + 32 * jn_);     /* the gated kernel has no SGPRs to spare for the burst */ \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                         \
                float b8_[8];                                                                                          \
                _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                       \
                    _Pragma("unroll") for (int d_ = 0; d_ < 2; ++d_) {                                                 \
                        /* plain kernel: the lane's words were selected once per tile (bsel); gated: per block from the scalar load */ \
                        uint32_t w_;                                                                                   \
                        if constexpr (GATED) w_ = (bw_[8 * q_ + 4 * h_ + 2 + d_] & hm) | (bw_[8 * q_ + 4 * h_ + d_] & ~hm); \
                        else w_ = bsel[jn_][q_][2 * h_ + d_];                                                          \
                        b8_[4 * h_ + 2 * d_] = p.bias ? bf2f((uint16_t)(w_ & 0xffff)) : 0.f;                           \
                        b8_[4 * h_ + 2 * d_ + 1] = p.bias ? bf2f((uint16_t)(w_ >> 16)) : 0.f;                          \
                    }                                                                                                  \
                float v_[8];                                                                                           \
                _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) {                                                     \
                    v_[c_] = W4_ACC(jn_, im_, 8 * q_ + c_) * p.alpha + b8_[c_];                                        \
                    if (GELU_) v_[c_] = w4_gelu_tanh(rbf(v_[c_]));                                                     \
                }                                                                                                      \
                const uint32_t w00_ = pack2bf(v_[0], v_[1]), w01_ = pack2bf(v_[2], v_[3]);                             \
                const uint32_t w10_ = pack2bf(v_[4], v_[5]), w11_ = pack2bf(v_[6], v_[7]);                             \
                auto s0_ = __builtin_amdgcn_permlane32_swap(w00_, w10_, false, false);                                 \
                auto s1_ = __builtin_amdgcn_permlane32_swap(w01_, w11_, false, false);                                 \
                *reinterpret_cast<uint4*>(stg + wofs + (((4 * jn_ + 2 * q_) << 4) ^ wxor)) = make_uint4(s0_[0], s1_[0], s0_[1], s1_[1]); \
            }                                                                                                          \
        }
    // piece t_ (rows 4t .. 4t+3) of block im_: staging buffer -> y (16 B of one row per lane).  All eight pieces of a block are read
    // before the first is used: one LDS latency per block, not per piece.
#define W4_EPI_READ(t_) (*reinterpret_cast<const uint4*>(stg + (t_) * 1024 + (rofs ^ (64 * ((t_) & 3)))))
#define W4_EPI_READ8() const uint4 y0_ = W4_EPI_READ(0), y1_ = W4_EPI_READ(1), y2_ = W4_EPI_READ(2), y3_ = W4_EPI_READ(3), \
                                   y4_ = W4_EPI_READ(4), y5_ = W4_EPI_READ(5), y6_ = W4_EPI_READ(6), y7_ = W4_EPI_READ(7);
    // addresses: uniform base of the wave's slab (+ piece rows, scalar) + one 32-bit per-lane offset -> SGPR-base global accesses
#define W4_CPTR(im_, t_) (cub + (long)(32 * (im_) + 4 * (t_)) * ldc * 2 + cvo)
#define W4_RPTR_U(im_, t_) (rub + (long)(32 * (im_) + 4 * (t_)) * p.ldres * 2)
#define W4_ROW(im_, t_) (rowg + 32 * (im_) + 4 * (t_))
    // The C stores carry the NONTEMPORAL hint.  One round of tiles writes 4 MB of C per XCD -- the size of its L2 -- and write-allocated C lines
    // evicted the operand panels the next K-tiles stream: per tile of M = 50 688, N = 3072 (us; profiles/r02_gemm_w4_probe_nt.log)
    //   K = 1024: 37.3 -> 29.2 (no C stores at all: 27.7) | K = 3072: 82.7 -> 77.7 (75.1) | K = 6144: 150.0 -> 146.8 (145.0)
    // i.e. the hint recovers 70-85 % of what the stores cost.
#define W4_STORE_U(im_, t_, O_) { const uint4 o4_ = O_; const w4_u32x4 ov_ = {o4_.x, o4_.y, o4_.z, o4_.w};                                                  \
                                  __builtin_nontemporal_store(ov_, reinterpret_cast<w4_u32x4*>(W4_CPTR(im_, t_))); }
#define W4_STORE_M(im_, t_, O_) if (W4_ROW(im_, t_) < p.M) { W4_STORE_U(im_, t_, O_) }
#define W4_PLAIN_BLOCK(im_, GELU_)                                                                                     \
        W4_EPI_WRITE(im_, GELU_)                                                                                       \
        {                                                                                                              \
            W4_EPI_READ8()                                                                                             \
            if (c_m0 + wm * 128 + 32 * (im_) + 32 <= p.M) {                                                            \
                W4_STORE_U(im_, 0, y0_) W4_STORE_U(im_, 1, y1_) W4_STORE_U(im_, 2, y2_) W4_STORE_U(im_, 3, y3_)        \
                W4_STORE_U(im_, 4, y4_) W4_STORE_U(im_, 5, y5_) W4_STORE_U(im_, 6, y6_) W4_STORE_U(im_, 7, y7_)        \
            } else {                                                                                                   \
                W4_STORE_M(im_, 0, y0_) W4_STORE_M(im_, 1, y1_) W4_STORE_M(im_, 2, y2_) W4_STORE_M(im_, 3, y3_)        \
                W4_STORE_M(im_, 4, y4_) W4_STORE_M(im_, 5, y5_) W4_STORE_M(im_, 6, y6_) W4_STORE_M(im_, 7, y7_)        \
            }                                                                                                          \
        }
    // ---- MX fp8 OUTPUT of the GELU tiles (MX kernel, utx_gemm_desc.q_out): after the LDS transpose a lane holds 8 consecutive columns of one row and 16
    // lanes share the row's 128 columns -- one K-tile of the NEXT GEMM, four blocks of 32 = four lanes each.  Block maximum by two shuffles, the four
    // scale bytes of the row's K-tile assembled into one dword by two more, stored by the first lane of the 16 at its tile-packed place
    // [K-tile][row block][row % 32][(row % 128) / 32]; the values are the bf16-rounded GELU outputs of the plain epilogue, so the bytes equal
    // epilogue -> utx_quant_mx8_packed.  8 bytes per lane and piece instead of 16: half the C traffic, and the quantiser's pass over the tensor is gone.
#define W4_MXQ_PIECE(im_, t_, Y_)                                                                                      \
        {                                                                                                              \
            const uint32_t yw_[4] = {Y_.x, Y_.y, Y_.z, Y_.w};                                                          \
            float f_[8], am_ = 0.f;                                                                                    \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                         \
                f_[2 * c_] = bf2f((uint16_t)(yw_[c_] & 0xffff)); f_[2 * c_ + 1] = bf2f((uint16_t)(yw_[c_] >> 16));     \
                am_ = fmaxf(am_, fmaxf(fabsf(f_[2 * c_]), fabsf(f_[2 * c_ + 1])));                                     \
            }                                                                                                          \
            am_ = fmaxf(am_, __shfl_xor(am_, 1, 64));                                                                  \
            am_ = fmaxf(am_, __shfl_xor(am_, 2, 64));                                                                  \
            int e_ = (int)((__float_as_uint(am_) >> 23) & 0xff) - 127 - 8;                                             \
            if (am_ == 0.f || e_ < -127) e_ = -127;                                                                    \
            if (e_ > 127) e_ = 127;                                                                                    \
            const float inv_ = __uint_as_float((uint32_t)(127 - e_) << 23);                                            \
            float a_[8];                                                                                               \
            _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) a_[c_] = fminf(fmaxf(f_[c_] * inv_, -448.f), 448.f);      \
            int p0_ = 0, p1_ = 0;                                                                                      \
            p0_ = __builtin_amdgcn_cvt_pk_fp8_f32(a_[0], a_[1], p0_, false); p0_ = __builtin_amdgcn_cvt_pk_fp8_f32(a_[2], a_[3], p0_, true);  \
            p1_ = __builtin_amdgcn_cvt_pk_fp8_f32(a_[4], a_[5], p1_, false); p1_ = __builtin_amdgcn_cvt_pk_fp8_f32(a_[6], a_[7], p1_, true);  \
            uint32_t w_ = (uint32_t)(e_ + 127) << (8 * ((elane >> 2) & 3));                                            \
            w_ |= (uint32_t)__shfl_xor((int)w_, 4, 64);                                                                \
            w_ |= (uint32_t)__shfl_xor((int)w_, 8, 64);                                                                \
            if (full || W4_ROW(im_, t_) < p.M) {                                                                       \
                *reinterpret_cast<uint2*>(mxq_b + (long)(32 * (im_) + 4 * (t_)) * p.ldq_out + mxq_vo) = make_uint2((uint32_t)p0_, (uint32_t)p1_); \
                if ((elane & 15) == 0) mxq_s[(4 * (t_) + (elane >> 4)) * 4 + (im_)] = w_;                              \
            }                                                                                                          \
        }
#define W4_MXQ_BLOCK(im_)                                                                                              \
        W4_EPI_WRITE(im_, true)                                                                                        \
        {                                                                                                              \
            W4_EPI_READ8()                                                                                             \
            W4_MXQ_PIECE(im_, 0, y0_) W4_MXQ_PIECE(im_, 1, y1_) W4_MXQ_PIECE(im_, 2, y2_) W4_MXQ_PIECE(im_, 3, y3_)    \
            W4_MXQ_PIECE(im_, 4, y4_) W4_MXQ_PIECE(im_, 5, y5_) W4_MXQ_PIECE(im_, 6, y6_) W4_MXQ_PIECE(im_, 7, y7_)    \
        }
    // ---- fused q / k post-processing (QKF; utx_gemm_desc.qk_cols): after the LDS transpose a lane holds 8 consecutive channels of one token's
    // head row and 16 lanes share the row -- exactly the decomposition of qkv_post_kernel (dit_elementwise.hip), so its arithmetic is repeated
    // verbatim: same sum order (in-lane pairs, then xor 8, 4, 2, 1 over the 16 lanes), same bf16 rounding points, IEEE div / sqrt, no contraction.
    // cos / sin (2 x float4 per piece and lane) are requested ahead in a rolling window of four pieces, the first window before the block's write
    // phase; the waits are literal counts of younger operations (full tiles; a ragged tile waits for everything).
    typedef __attribute__((ext_vector_type(4))) float w4_f32x4v;
    // all addresses of this path are (kernel-argument base in SGPRs) + (32-bit VGPR offset): a 64-bit per-piece address would be computed with
    // vector multiplies, hoisted out of the tile loop and spilled inside the K loop
#define W4_QK_ROWOFF(im_, t_) ((unsigned)(qsrow0 + 32 * (im_) + 4 * (t_) > qsrow_max ? qsrow_max : qsrow0 + 32 * (im_) + 4 * (t_)) * 256u)
#define W4_QK_CS_LOAD(dst_, tab_, im_, t_)                                                                             \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(qvo + W4_QK_ROWOFF(im_, t_)), "s"(tab_) : "memory")
#define W4_QK_PIECE(im_, t_, Y_, CS_, SN_, n_)                                                                         \
        {                                                                                                              \
            if (full) { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n_) : "memory"); }                                  \
            else { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }                                                \
            W4_FENCE();                                                                                                \
            W4_LANDED2(CS_, SN_);                                                                                      \
            if ((im_) == 0 && (t_) == 0) W4_LANDED(gq);     /* the norm weight requested by W4_GATE_FETCH is older than every cos / sin load */ \
            W4_FENCE();                                                                                                \
            {                                                                                                          \
            _Pragma("clang fp contract(off)")                                                                          \
            const uint32_t rw_[4] = {Y_.x, Y_.y, Y_.z, Y_.w};                                                          \
            float x_[8];                                                                                               \
            float ss_ = 0.f;                                                                                           \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                         \
                x_[2 * c_] = bf2f((uint16_t)(rw_[c_] & 0xffff)); x_[2 * c_ + 1] = bf2f((uint16_t)(rw_[c_] >> 16));     \
                ss_ += x_[2 * c_] * x_[2 * c_] + x_[2 * c_ + 1] * x_[2 * c_ + 1];                                      \
            }                                                                                                          \
            _Pragma("unroll") for (int o_ = 8; o_ > 0; o_ >>= 1) ss_ += __shfl_xor(ss_, o_, 64);                       \
            const float rstd_ = 1.0f / sqrtf(ss_ / 128.0f + p.qk_eps);                                                 \
            uint32_t ow_[4];                                                                                           \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                         \
                const float w0_ = bf2f((uint16_t)(gq[c_] & 0xffff)), w1_ = bf2f((uint16_t)(gq[c_] >> 16));             \
                const float a0_ = rbf(rbf(x_[2 * c_] * rstd_) * w0_);                                                  \
                const float a1_ = rbf(rbf(x_[2 * c_ + 1] * rstd_) * w1_);                                              \
                const float r0_ = (a0_ * CS_[c_] + (-a1_) * SN_[c_]) * qs;                                             \
                const float r1_ = (a1_ * CS_[c_] + a0_ * SN_[c_]) * qs;                                                \
                ow_[c_] = pack2bf(r0_, r1_);                                                                           \
            }                                                                                                          \
            if (full || W4_ROW(im_, t_) < p.M) {     /* s_nop 1 inside the store's string: hipcc does not see the store, and a VALU write of its data registers within two wait states of a > 64-bit store corrupts the data */ \
                w4_u32x4 od_ = {ow_[0], ow_[1], ow_[2], ow_[3]};                                                       \
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(qvo + qhead + (unsigned)(qsrow0 + 32 * (im_) + 4 * (t_)) * 256u), "v"(od_), "s"(qdst) : "memory"); \
            }                                                                                                          \
            }                                                                                                          \
            W4_FENCE();                                                                                                \
        }
#define W4_QK_BLOCK(im_)                                                                                               \
        {                                                                                                              \
            /* rolling window of four pieces of cos / sin (16 registers): piece t + 4 is requested into the registers of piece t as soon as that is done. \
               Issue order of a block: L0 L1 L2 L3 (two loads each) | [wait L(t), store S(t), refill R(t + 4) (two loads)] t = 0..3 | [wait R(4 + j), S(4 + j)] \
               j = 0..3.  Literal waits (full tiles) = the number of YOUNGER vector-memory operations, loads and stores alike -- one counter, in-order      \
               retirement (tools/vmcnt_order_probe.hip; DESIGN "waits and hazards"): behind L(t) come 2 (3 - t) window loads + t stores + 2 t refills      \
               = 6 + t; behind R(4 + j) come the refills and stores of pieces j + 1 .. 3 (3 (3 - j)) and the stores S(4) .. S(3 + j) (j) = 9 - 2 j.        \
               (Round 3 suspected these counts after a one-off one-ulp difference and dropped the stores from them.  The cause was elsewhere: the           \
               compiler's register copy IN FRONT of the ragged branch's tied-operand wait -- see W4_WAIT_VM.)  */ \
            w4_f32x4v c0_, s0_, c1_, s1_, c2_, s2_, c3_, s3_;                                                           \
            W4_QK_CS_LOAD(c0_, p.qk_cos, im_, 0); W4_QK_CS_LOAD(s0_, p.qk_sin, im_, 0); W4_QK_CS_LOAD(c1_, p.qk_cos, im_, 1); W4_QK_CS_LOAD(s1_, p.qk_sin, im_, 1); \
            W4_QK_CS_LOAD(c2_, p.qk_cos, im_, 2); W4_QK_CS_LOAD(s2_, p.qk_sin, im_, 2); W4_QK_CS_LOAD(c3_, p.qk_cos, im_, 3); W4_QK_CS_LOAD(s3_, p.qk_sin, im_, 3); \
            W4_FENCE();                                                                                                \
            W4_EPI_WRITE(im_, false)                                                                                   \
            W4_EPI_READ8()                                                                                             \
            W4_QK_PIECE(im_, 0, y0_, c0_, s0_, 6) W4_QK_CS_LOAD(c0_, p.qk_cos, im_, 4); W4_QK_CS_LOAD(s0_, p.qk_sin, im_, 4); W4_FENCE(); \
            W4_QK_PIECE(im_, 1, y1_, c1_, s1_, 7) W4_QK_CS_LOAD(c1_, p.qk_cos, im_, 5); W4_QK_CS_LOAD(s1_, p.qk_sin, im_, 5); W4_FENCE(); \
            W4_QK_PIECE(im_, 2, y2_, c2_, s2_, 8) W4_QK_CS_LOAD(c2_, p.qk_cos, im_, 6); W4_QK_CS_LOAD(s2_, p.qk_sin, im_, 6); W4_FENCE(); \
            W4_QK_PIECE(im_, 3, y3_, c3_, s3_, 9) W4_QK_CS_LOAD(c3_, p.qk_cos, im_, 7); W4_QK_CS_LOAD(s3_, p.qk_sin, im_, 7); W4_FENCE(); \
            W4_QK_PIECE(im_, 4, y4_, c0_, s0_, 9) W4_QK_PIECE(im_, 5, y5_, c1_, s1_, 7) W4_QK_PIECE(im_, 6, y6_, c2_, s2_, 5) W4_QK_PIECE(im_, 7, y7_, c3_, s3_, 3) \
        }
    // gated: out = res + bf16(gate * y), the rounding points of gemm.hip
#define W4_GATE_OUT(Y_, R_, O_)                                                                                        \
        {                                                                                                              \
            const uint32_t yy_[4] = {Y_.x, Y_.y, Y_.z, Y_.w};                                                          \
            uint32_t oo_[4];                                                                                           \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                         \
                const float y0f_ = bf2f((uint16_t)(yy_[c_] & 0xffff)), y1f_ = bf2f((uint16_t)(yy_[c_] >> 16));         \
                const float r0f_ = bf2f((uint16_t)(R_[c_] & 0xffff)), r1f_ = bf2f((uint16_t)(R_[c_] >> 16));           \
                const float g0f_ = bf2f((uint16_t)(gq[c_] & 0xffff)), g1f_ = bf2f((uint16_t)(gq[c_] >> 16));           \
                oo_[c_] = pack2bf(r0f_ + rbf(g0f_ * y0f_), r1f_ + rbf(g1f_ * y1f_));                                   \
            }                                                                                                          \
            O_ = make_uint4(oo_[0], oo_[1], oo_[2], oo_[3]);                                                           \
        }
    // residual loads: SGPR base + 32-bit lane offset, invisible to hipcc
#define W4_RES_LOAD(dst_, im_, t_) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(rvo), "s"(W4_RPTR_U(im_, t_)) : "memory")
    // fast path (full tile: every store instruction is issued, so the in-order op counts behind the waits are exact): the residual
    // pieces of two blocks are in flight at any time (16 x 4 registers); piece (im, t) is waited for with the literal count of
    // younger loads and stores: 15 + t, 23 + t, 30 - t, 22 - t for im = 0..3 (derivation in DESIGN.md, "GEMM, one wave per SIMD")
#define W4_GATE_PIECE(im_, t_, Y_, R_, n_, next_)                                                                      \
        {                                                                                                              \
            W4_WAIT_RES(n_, R_);                                                                                       \
            if ((im_) == 0 && (t_) == 0) { W4_LANDED(gq); W4_FENCE(); }     /* the gate vector (W4_GATE_FETCH) is older than every residual load */ \
            uint4 o_;                                                                                                  \
            W4_GATE_OUT(Y_, R_, o_)                                                                                    \
            W4_STORE_U(im_, t_, o_)                                                                                    \
            W4_FENCE();                                                                                                \
            if (next_) W4_RES_LOAD(R_, (im_) + 2, t_);                                                                 \
            W4_FENCE();                                                                                                \
        }
#define W4_GATE_BLOCK(im_, RA_, base_, sign_, next_)                                                                   \
        W4_EPI_WRITE(im_, false)                                                                                       \
        {                                                                                                              \
            W4_EPI_READ8()                                                                                             \
            W4_GATE_PIECE(im_, 0, y0_, RA_##0, (base_) + (sign_) * 0, next_) W4_GATE_PIECE(im_, 1, y1_, RA_##1, (base_) + (sign_) * 1, next_) \
            W4_GATE_PIECE(im_, 2, y2_, RA_##2, (base_) + (sign_) * 2, next_) W4_GATE_PIECE(im_, 3, y3_, RA_##3, (base_) + (sign_) * 3, next_) \
            W4_GATE_PIECE(im_, 4, y4_, RA_##4, (base_) + (sign_) * 4, next_) W4_GATE_PIECE(im_, 5, y5_, RA_##5, (base_) + (sign_) * 5, next_) \
            W4_GATE_PIECE(im_, 6, y6_, RA_##6, (base_) + (sign_) * 6, next_) W4_GATE_PIECE(im_, 7, y7_, RA_##7, (base_) + (sign_) * 7, next_) \
        }
    // ragged tile (rows >= M clamp their residual row, their stores are masked): one piece at a time, every load waited for on the spot
#define W4_GATE_SLOW_PIECE(im_, t_, Y_)                                                                                \
        {                                                                                                              \
            w4_u32x4 r_;                                                                                               \
            W4_LOAD16_ASM(r_, pres + (long)(W4_ROW(im_, t_) > p.M - 1 ? p.M - 1 : W4_ROW(im_, t_)) * p.ldres + gcol);  \
            W4_WAIT_RES(0, r_);                                                                                        \
            if ((im_) == 0 && (t_) == 0) { W4_LANDED(gq); W4_FENCE(); }                                                \
            uint4 o_;                                                                                                  \
            W4_GATE_OUT(Y_, r_, o_)                                                                                    \
            W4_STORE_M(im_, t_, o_)                                                                                    \
        }
#define W4_GATE_SLOW_BLOCK(im_)                                                                                        \
        W4_EPI_WRITE(im_, false)                                                                                       \
        {                                                                                                              \
            W4_EPI_READ8()                                                                                             \
            W4_GATE_SLOW_PIECE(im_, 0, y0_) W4_GATE_SLOW_PIECE(im_, 1, y1_) W4_GATE_SLOW_PIECE(im_, 2, y2_) W4_GATE_SLOW_PIECE(im_, 3, y3_) \
            W4_GATE_SLOW_PIECE(im_, 4, y4_) W4_GATE_SLOW_PIECE(im_, 5, y5_) W4_GATE_SLOW_PIECE(im_, 6, y6_) W4_GATE_SLOW_PIECE(im_, 7, y7_) \
        }
#define W4_EPILOGUE()                                                                                                  \
    do {                                                                                                               \
        int el31 = l31, elh = lh, elane = lane;                                                                        \
        asm volatile("" : "+v"(el31), "+v"(elh), "+v"(elane));                                                         \
        const uint32_t hm = 0u - (uint32_t)elh;      /* all-ones in lanes 32-63: bit-select instead of ?: (which hipcc turns into a 16-way dynamic SGPR index) */ \
        const bool do_gelu = !GATED && c_n0 >= p.gelu_from;                                                            \
        const bool to_c1 = !GATED && c_n0 >= p.n_split;                                                                \
        bf16_t* const cbase = to_c1 ? (bf16_t*)p.C1 : (bf16_t*)p.C;                                                    \
        const long ldc = to_c1 ? p.ldc1 : p.ldc;                                                                       \
        const int wofs = el31 * 256, wxor = ((el31 & 15) ^ elh) << 4;               /* staging write: row l31, chunk (4jn + 2q + lh) ^ (l31 & 15) */ \
        const int rofs = (elane >> 4) * 256 + (((elane & 15) ^ (elane >> 4)) << 4); /* staging read: row 4t + lane/16, chunk lane%16 ^ row%16 */ \
        const int rowg = c_m0 + wm * 128 + (elane >> 4);                                                               \
        const int gcol = c_n0 + wn * 128 + 8 * (elane & 15);                                                           \
        const int ccol = gcol - (to_c1 ? p.n_split : 0);                                                               \
        const bf16_t* const ebias = (const bf16_t*)(p.bias ? p.bias : p.B) + c_n0 + wn * 128;                          \
        char* const cub = (char*)cbase + ((long)(c_m0 + wm * 128) * ldc + (c_n0 + wn * 128 - (to_c1 ? p.n_split : 0))) * 2;   /* uniform */ \
        const unsigned cvo = (unsigned)((elane >> 4) * (unsigned)ldc * 2u + (unsigned)(elane & 15) * 16u);              \
        const char* const rub = (const char*)pres + ((long)(c_m0 + wm * 128) * p.ldres + c_n0 + wn * 128) * 2;         \
        const unsigned rvo = (unsigned)((elane >> 4) * (unsigned)p.ldres * 2u + (unsigned)(elane & 15) * 16u);          \
        w4_u32x4 gq = {0u, 0u, 0u, 0u};                                                                                \
        W4_GATE_FETCH();                                                                                               \
        w4_u32x16 bq0, bq1, bq2, bq3;                                                                                  \
        uint32_t bsel[4][2][4];          /* plain kernel: this lane's bias words (packed bf16 pairs) of every (jn, q) piece */ \
        if constexpr (!GATED) {                                                                                        \
            W4_SLOAD64(bq0, bq1, bq2, bq3, ebias);                                                                     \
            _Pragma("unroll") for (int jn_ = 0; jn_ < 4; ++jn_) {                                                      \
                const w4_u32x16 bw_ = jn_ == 0 ? bq0 : jn_ == 1 ? bq1 : jn_ == 2 ? bq2 : bq3;                          \
                _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                       \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                   \
                        bsel[jn_][q_][e_] = (bw_[8 * q_ + 4 * (e_ >> 1) + 2 + (e_ & 1)] & hm) | (bw_[8 * q_ + 4 * (e_ >> 1) + (e_ & 1)] & ~hm); \
            }                                                                                                          \
        }                                                                                                              \
        if constexpr (GATED) {                                                                                         \
            if (full) {                                                                                                \
                w4_u32x4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;               \
                W4_RES_LOAD(ra0, 0, 0); W4_RES_LOAD(ra1, 0, 1); W4_RES_LOAD(ra2, 0, 2); W4_RES_LOAD(ra3, 0, 3);         \
                W4_RES_LOAD(ra4, 0, 4); W4_RES_LOAD(ra5, 0, 5); W4_RES_LOAD(ra6, 0, 6); W4_RES_LOAD(ra7, 0, 7);         \
                W4_RES_LOAD(rb0, 1, 0); W4_RES_LOAD(rb1, 1, 1); W4_RES_LOAD(rb2, 1, 2); W4_RES_LOAD(rb3, 1, 3);         \
                W4_RES_LOAD(rb4, 1, 4); W4_RES_LOAD(rb5, 1, 5); W4_RES_LOAD(rb6, 1, 6); W4_RES_LOAD(rb7, 1, 7);         \
                W4_FENCE();                                                                                            \
                W4_GATE_BLOCK(0, ra, 15, 1, true)                                                                      \
                W4_GATE_BLOCK(1, rb, 23, 1, true)                                                                      \
                W4_GATE_BLOCK(2, ra, 30, -1, false)                                                                    \
                W4_GATE_BLOCK(3, rb, 22, -1, false)                                                                    \
            } else {                                                                                                   \
                W4_GATE_SLOW_BLOCK(0) W4_GATE_SLOW_BLOCK(1) W4_GATE_SLOW_BLOCK(2) W4_GATE_SLOW_BLOCK(3)                \
            }                                                                                                          \
        } else if (QKF && c_n0 < p.qk_cols) {                                                                         \
            /* this wave's 128 columns = one head of q (columns < qk_cols / 2) or k */                                 \
            const int col0 = c_n0 + wn * 128, dq = p.qk_cols >> 1;                                                      \
            const bool isk = col0 >= dq;                                                                               \
            const int head = (isk ? col0 - dq : col0) >> 7;                                                            \
            const float qs = isk ? 1.0f : p.qk_q_scale;                                                                \
            const int qsrow0 = p.qk_tok_off + c_m0 + wm * 128, qsrow_max = p.qk_tok_off + p.M - 1;                      \
            const unsigned qvo = (unsigned)((elane >> 4) * 256 + (elane & 15) * 16);                                    \
            const char* const qdst = (const char*)(isk ? p.qk_Kh : p.qk_Qh);                                          \
            const unsigned qhead = (unsigned)head * (unsigned)p.qk_hs * 2u;        /* byte offset of this head: < 4 GB for any S this library sizes for */ \
            W4_QK_BLOCK(0) W4_QK_BLOCK(1) W4_QK_BLOCK(2) W4_QK_BLOCK(3)                                                \
        } else if (MX && !GATED && do_gelu && p.q_out) {                                                              \
            /* the wave's 128 GELU columns = K-tile (n - gelu_from) / 128 of the next GEMM's activation operand */         \
            const int qc0 = c_n0 + wn * 128 - p.gelu_from;                                                             \
            uint8_t* const mxq_b = (uint8_t*)p.q_out + (long)(c_m0 + wm * 128) * p.ldq_out + qc0;                      \
            const unsigned mxq_vo = (unsigned)((elane >> 4) * (unsigned)p.ldq_out + (unsigned)(elane & 15) * 8u);      \
            uint32_t* const mxq_s = (uint32_t*)p.qs_out + ((long)(p.q_out_kt0 + (qc0 >> 7)) * p.qs_out_rb + ((c_m0 >> 7) + wm)) * 128; \
            W4_MXQ_BLOCK(0) W4_MXQ_BLOCK(1) W4_MXQ_BLOCK(2) W4_MXQ_BLOCK(3)                                            \
        } else {                                                                                                       \
            if (do_gelu) { W4_PLAIN_BLOCK(0, true) W4_PLAIN_BLOCK(1, true) W4_PLAIN_BLOCK(2, true) W4_PLAIN_BLOCK(3, true) } \
            else { W4_PLAIN_BLOCK(0, false) W4_PLAIN_BLOCK(1, false) W4_PLAIN_BLOCK(2, false) W4_PLAIN_BLOCK(3, false) } \
        }                                                                                                              \
    } while (0)

    // ---- end of a tail range: the accumulators as they lie in the registers -> slot r (= the range's number) of the workspace, [wave][jn][im][a][lane] x 16 B:
    // every store instruction writes one contiguous KB
#define W4_DUMP(slot_)                                                                                                 \
    do {                                                                                                               \
        int dlane = lane;                                                                                              \
        asm volatile("" : "+v"(dlane));                                                                                \
        float* const wsb = (float*)p.sk_work + ((long)(slot_) * 4 + wave) * 16384 + dlane * 4;                         \
        _Pragma("unroll") for (int jn_ = 0; jn_ < 4; ++jn_)                                                            \
        _Pragma("unroll") for (int im_ = 0; im_ < 4; ++im_)                                                            \
        _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_) {                                                             \
            const w4_f32x4 v_ = {W4_ACC(jn_, im_, 4 * a_), W4_ACC(jn_, im_, 4 * a_ + 1), W4_ACC(jn_, im_, 4 * a_ + 2), W4_ACC(jn_, im_, 4 * a_ + 3)}; \
            *reinterpret_cast<w4_f32x4*>(wsb + ((jn_ * 4 + im_) * 4 + a_) * 256) = v_;                                 \
        }                                                                                                              \
    } while (0)

    // ---- prologue: K-tile 0 complete and landed, operand A of K-tile 1 requested; F0 = fragments of K-step 0 of K-tile 0
#define W4_STAGE_HALF(isb_) do { W4_DMA(isb_, 0); W4_DMA(isb_, 1); W4_DMA(isb_, 2); W4_DMA(isb_, 3); W4_DMA(isb_, 4); W4_DMA(isb_, 5); W4_DMA(isb_, 6); W4_DMA(isb_, 7); } while (0)
    W4_STAGE_AT(0, -1, false);
    // the sixteen fragment reads of an MX K-step, literal read numbers (W4_XREAD)
#define W4_XREAD16(FA_, FB_, base_, xlo_, xhi_)                                                            \
    W4_XREAD(0, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(1, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(2, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(3, FA_, FB_, base_, xlo_, xhi_);     \
    W4_XREAD(4, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(5, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(6, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(7, FA_, FB_, base_, xlo_, xhi_);     \
    W4_XREAD(8, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(9, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(10, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(11, FA_, FB_, base_, xlo_, xhi_);   \
    W4_XREAD(12, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(13, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(14, FA_, FB_, base_, xlo_, xhi_); W4_XREAD(15, FA_, FB_, base_, xlo_, xhi_);
    if constexpr (MX) {
        // K-tile 0 complete with its scales, pieces 0..10 of K-tile 1 requested (what K-step 1 of a K-tile -1 would have requested)
        W4_STAGE_HALF(0); W4_STAGE_HALF(1);
        W4_XSCALE_LOAD(sa_nxt, s_pSA); W4_XSCALE_LOAD(sb_nxt, s_pSB);
        W4_STAGE_ADVANCE();
        W4_DMA(0, 0); W4_DMA(0, 1); W4_DMA(0, 2); W4_DMA(0, 3); W4_DMA(0, 4); W4_DMA(0, 5); W4_DMA(0, 6); W4_DMA(0, 7);
        W4_DMA(1, 0); W4_DMA(1, 1); W4_DMA(1, 2);
        W4_WAIT_VM(11);
        W4_LANDED2(sa_nxt, sb_nxt);
        W4_FENCE();
        W4_XSCALE_TAKE();
        W4_COMPUTE_AT(0, -1);
        __builtin_amdgcn_s_barrier();
        W4_FENCE();
        W4_XREAD16(xa0, xb0, 0, xk0, xk1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE();
    } else {
    W4_STAGE_HALF(0); W4_STAGE_HALF(1); W4_STAGE_ADVANCE();
    W4_DMA(0, 0); W4_DMA(0, 1); W4_DMA(0, 2); W4_DMA(0, 3); W4_DMA(0, 4); W4_DMA(0, 5);      // what K-step 3 of a K-tile -1 would have requested
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    W4_COMPUTE_AT(0, -1);
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
#pragma unroll
    for (int r = 0; r < 8; ++r) W4_READ(r, fa0, fb0, 0, xk0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
    }

    // One K-tile = four K-steps of 16 MFMAs (schedule in the file header).  Slot i (behind MFMA i) of a K-step carries one fragment read (i < 8: the
    // fragments of the K-step after this one) and / or one DMA; the LDS address of a DMA is put into M0 in front of the MFMA before it (the wait state
    // M0 needs).
    // fragment read addresses of the generated stream: absolute LDS byte address of (stage slot, k-chunk kk) for operand A / B; the stream adds 4096 x fragment index as an immediate
    unsigned w4f_ra[2][4], w4f_rb[2][4];
    if constexpr (FK && !MX) {
        const int xk_[4] = {xk0, xk1, xk2, xk3};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int k = 0; k < 4; ++k) { w4f_ra[sl][k] = lds0 + (unsigned)(sl * W4_STAGE + arow + xk_[k]); w4f_rb[sl][k] = lds0 + (unsigned)(sl * W4_STAGE + brow + xk_[k]); }
    }
    for (;;) {
        if constexpr (FK && !MX) {
            // W4_FAST_TRIPS: pairs of K-tiles during which neither cursor meets a boundary -- the compute cursor stays short of its segment's last K-tile (the tile boundary and its
            // epilogue belong to the loop below), the staging cursor short of the K-tile whose advance re-places it (W4_STAGE_ADVANCE: s_ss == s_seg_end).  Stage slots: the
            // stream's first K-tile computes on slot 0, so it is entered only with c_slot == 0 (s_slot == 1: the invariant s_slot == c_slot ^ 1 holds at every loop top).
            if (c_slot == 0) {
                int n_ = (s_seg_end - s_ss - 1) >> 1;
                const int nc_ = (c_nss - c_ss - 1) >> 1;
                if (nc_ < n_) n_ = nc_;
                if (n_ > 0) {
                    const unsigned long pa_ = (unsigned long)s_pA, pb_ = (unsigned long)s_pB;
                    const unsigned w4f_pa_lo = __builtin_amdgcn_readfirstlane((unsigned)pa_), w4f_pa_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa_ >> 32));
                    const unsigned w4f_pb_lo = __builtin_amdgcn_readfirstlane((unsigned)pb_), w4f_pb_hi = __builtin_amdgcn_readfirstlane((unsigned)(pb_ >> 32));
                    const unsigned w4f_trips = __builtin_amdgcn_readfirstlane((unsigned)n_);
                    const unsigned w4f_ldsdma = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
                    asm volatile(W4F_ASM_TEXT : W4F_ASM_OUTPUTS : W4F_ASM_INPUTS : W4F_ASM_CLOBBERS);
                    s_pA += 256L * n_; s_pB += 256L * n_;
                    s_ss += 2 * n_; c_ss += 2 * n_;
                }
            }
        }
        const int cb = c_slot * W4_STAGE, nb = (c_slot ^ 1) * W4_STAGE;
        // DMA schedule: piece p = 0..15 of the cursor's K-tile (p < 8: operand A piece p, else operand B piece p - 8), ONE DMA PER THREE MFMAs over
        // K-steps 3, 0, 1 (a burst of 1 KB requests backs the vector-memory path up into the issuing wave, which has no partner wave to hide behind;
        // measured: one per MFMA -> one per two +2 %, -> one per three +3-4 % more; the last piece then has 18 MFMAs ~ 0.37 us to land, against
        // 0.3-0.4 us of loaded L2 latency -- tools/l2_fill_probe.hip):
        //   K-step 3: p 0..5 behind MFMAs 0, 3, ..., 15 | K-step 0: p 6..10 behind MFMAs 2, 5, ..., 14 | K-step 1: p 11..15 behind MFMAs 1, 4, ..., 13, then the cursor advances
#define W4_PIECE_OF(ks_, i_) ((ks_) == 3 ? (((i_) % 3) == 0 ? (i_) / 3 : -1) : (ks_) == 0 ? (((i_) % 3) == 2 ? 6 + (i_) / 3 : -1) \
                              : (ks_) == 1 ? ((((i_) % 3) == 1 && (i_) < 15) ? 11 + (i_) / 3 : -1) : -1)
#define W4_KSTEP(i_, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)                                                \
        {                                                                                                  \
            constexpr int pc_ = W4_PIECE_OF(ks_, i_);                                                      \
            constexpr int isb_ = pc_ >= 8 ? 1 : 0, d_ = pc_ < 0 ? 0 : (pc_ & 7);                           \
            if constexpr (pc_ >= 0) W4_MF_M0(i_, FA_, FB_, W4_DMA_LDS(isb_, d_)); else W4_MF(i_, FA_, FB_); \
            W4_FENCE();                                                                                    \
            if ((i_) < 8) W4_READ(i_, FAn_, FBn_, base_, xk_);                                             \
            if constexpr (pc_ >= 0) W4_DMA_M0(isb_, d_);                                                   \
            W4_FENCE();                                                                                    \
        }
#define W4_KSTEP16(ks_, FA_, FB_, FAn_, FBn_, base_, xk_)                                                  \
        W4_KSTEP(0, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(1, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)   \
        W4_KSTEP(2, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(3, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)   \
        W4_KSTEP(4, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(5, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)   \
        W4_KSTEP(6, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(7, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)   \
        W4_KSTEP(8, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(9, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)   \
        W4_KSTEP(10, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(11, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) \
        W4_KSTEP(12, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(13, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) \
        W4_KSTEP(14, ks_, FA_, FB_, FAn_, FBn_, base_, xk_) W4_KSTEP(15, ks_, FA_, FB_, FAn_, FBn_, base_, xk_)
        // ---- MX fp8: a K-tile (128 k) is TWO K-steps of 16 scaled MFMAs (64 cycles each: the same 2048 matrix-pipe cycles as the 64 bf16 MFMAs
        // of a 64-k K-tile, on twice the k).  Slot i carries one 16-byte fragment read (the next K-step's) and at most one DMA:
        //   K-step 0: MFMAs on X0 | reads -> X1 (second half of this stage) | pieces 11..15 of the cursor's K-tile (= compute K-tile + 1) in slots
        //             0, 1, 3, 4, 6, its scale dwords in slots 0, 1, the cursor advances; lgkmcnt(0), vmcnt(0) [9 MFMAs = 576 cycles for the last piece
        //             to land: the bf16 schedule's 18 x 32], s_barrier
        //   K-step 1: MFMAs on X1 | reads -> X0 (first half of the NEXT stage) | pieces 0..10 of K-tile + 2 into THIS stage (two per three slots:
        //             one DMA per 96 matrix-pipe cycles, the spacing the bf16 schedule settled on); scales next -> current
        // (other placements of the same sixteen pieces were measured in round 3, profiles/r03_mx_dma_sweep_v0.log: 3 in K-step 0 + 13 in K-step 1 equal (+-0.3 %), 8 + 8
        // -2.3...-2.7 % on every shape (the last piece gets 8 instead of 9 MFMAs to land), the two scale loads in slots 0, 1 instead of 8, 9 +0.4...+1.8 % -> adopted)
#define W4_XPIECE_OF(ks_, i_) ((ks_) == 1 ? ((((i_) % 3) != 2) ? ((i_) / 3) * 2 + ((i_) % 3) : -1) : (((((i_) % 3) != 2) && (i_) < 7) ? 11 + ((i_) / 3) * 2 + ((i_) % 3) : -1))
#define W4_XKSTEP(i_, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)                                        \
        {                                                                                                  \
            constexpr int pc_ = W4_XPIECE_OF(ks_, i_);                                                     \
            constexpr int isb_ = pc_ >= 8 ? 1 : 0, d_ = pc_ < 0 ? 0 : (pc_ & 7);                           \
            if constexpr (pc_ >= 0) W4_XMF_M0(i_, ks_, FA_, FB_, xlds + (unsigned)(isb_ * 32768 + d_ * 4096)); else W4_XMF(i_, ks_, FA_, FB_); \
            W4_FENCE();                                                                                    \
            W4_XREAD(i_, FAn_, FBn_, base_, xlo_, xhi_);                                                   \
            if constexpr (pc_ >= 0) W4_DMA_M0(isb_, d_);                                                   \
            if constexpr ((ks_) == 0 && (i_) == 0) W4_XSCALE_LOAD(sa_nxt, s_pSA);                          \
            if constexpr ((ks_) == 0 && (i_) == 1) W4_XSCALE_LOAD(sb_nxt, s_pSB);                          \
            W4_FENCE();                                                                                    \
        }
#define W4_XKSTEP16(ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)                                          \
        W4_XKSTEP(0, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(1, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)   \
        W4_XKSTEP(2, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(3, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)   \
        W4_XKSTEP(4, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(5, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)   \
        W4_XKSTEP(6, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(7, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)   \
        W4_XKSTEP(8, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(9, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)   \
        W4_XKSTEP(10, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(11, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) \
        W4_XKSTEP(12, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(13, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) \
        W4_XKSTEP(14, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_) W4_XKSTEP(15, ks_, FA_, FB_, FAn_, FBn_, base_, xlo_, xhi_)
        if constexpr (MX) {
            unsigned xlds = lds_w + (unsigned)s_slot * W4_STAGE;     // one scalar base per K-step, a literal add per DMA (sixteen precomputed piece addresses do not fit the SGPR file beside the cursor)
            W4_XKSTEP16(0, xa0, xb0, xa1, xb1, cb, xk2, xk3)
            W4_STAGE_ADVANCE();
            xlds = lds_w + (unsigned)s_slot * W4_STAGE;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_WAIT_VM(0);      // K-tile + 1 and its scales landed
            W4_LANDED2(sa_nxt, sb_nxt);
            W4_FENCE();
            __builtin_amdgcn_s_barrier();
            W4_FENCE();
            W4_XKSTEP16(1, xa1, xb1, xa0, xb0, nb, xk0, xk1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_FENCE();
            W4_XSCALE_TAKE();
            W4_FENCE();
        } else {
        // ---- K-step 0
        W4_KSTEP16(0, fa0, fb0, fa1, fb1, cb, xk1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        // ---- K-step 1
        W4_KSTEP16(1, fa1, fb1, fa0, fb0, cb, xk2)
        W4_STAGE_ADVANCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        // ---- K-step 2
        W4_KSTEP16(2, fa0, fb0, fa1, fb1, cb, xk3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K-tile + 1 landed (its last piece was requested 1.5 K-steps ago)
        W4_FENCE();
        __builtin_amdgcn_s_barrier();
        W4_FENCE();
        // ---- K-step 3
        W4_KSTEP16(3, fa1, fb1, fa0, fb0, nb, xk0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        }
        c_slot ^= 1;
        ++c_ss;
        if (c_ss != c_nss) continue;
        // ---- tile boundary
        const bool full = (c_m0 + 256 <= p.M);
        W4_MFMA_DRAIN();
        if (c_tj > 0) { W4_DUMP(wid + (c_tj - 1) * G); }
        else W4_EPILOGUE();
        W4_ZERO_ACC()      // one zeroing behind both paths: the accumulator tile has a single definition at the join
        if constexpr (MX) {    // as QKF below: X0 (64 registers) is not kept alive across the epilogue, the stage is intact
            const int rb = c_slot * W4_STAGE;
            W4_XREAD16(xa0, xb0, rb, xk0, xk1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_FENCE();
        }
        if constexpr (QKF) {   // F0 of the next tile's first K-step was prefetched by K-step 3 -- the fused q / k variant re-reads it here instead of
            // keeping 32 registers alive across its larger epilogue (the stage is intact; one exposed LDS latency per tile)
            const int rb = c_slot * W4_STAGE;
#pragma unroll
            for (int r = 0; r < 8; ++r) W4_READ(r, fa0, fb0, rb, xk0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_FENCE();
        }
        bool done;
        W4_COMPUTE_NEXT(done);
        if (done) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the parked cursor's DMAs target this workgroup's LDS: retire them before it is released
}

// Second kernel of the split tail: tail tile t = ntiles - sk_T + blockIdx.x / 16; workgroup (t, wave quadrant wv, 32-row block im) sums the tile's sk_S
// slots (j sk_T + t, ascending j = ascending K: a fixed order) and runs the epilogue of the main kernel (same expressions, same rounding points).
// A thread owns what a lane of the main kernel owned: (jn = tid / 64, lane) x a = 0..3 x 4 columns.
template <bool GATED>
__global__ __launch_bounds__(256) void gemm_w4_fixup_kernel(GemmParams p, int ntiles, int sk_T, int sk_S) {
    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    const int t = blockIdx.x >> 4, wv = (blockIdx.x >> 2) & 3, im = blockIdx.x & 3;
    const int tid = threadIdx.x, lane = tid & 63, jn = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int ntn = p.ntn & 0xffff, group_m = (p.ntn >> 16) & 0xff, ntm = (p.M + 255) / 256, per_group = group_m * ntn;
    const int w = ntiles - sk_T + t;
    const int grp = w / per_group, rem = w - grp * per_group, ftm = grp * group_m;
    const int gs = (ntm - ftm < group_m) ? ntm - ftm : group_m;
    const int tn = rem / gs;
    const int m0 = (ftm + rem - tn * gs) * 256, n0 = tn * 256;
    f32x4v acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < sk_S; ++j) {
        const float* src = (const float*)p.sk_work + ((long)(j * sk_T + t) * 4 + wv) * 16384 + (jn * 4 + im) * 1024 + lane * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] += *reinterpret_cast<const f32x4v*>(src + a * 256);
    }
    const int m = m0 + wm * 128 + im * 32 + l31;
    if (m >= p.M) return;
    const bool do_gelu = !GATED && n0 >= p.gelu_from, to_c1 = !GATED && n0 >= p.n_split;
    bf16_t* const crow = to_c1 ? (bf16_t*)p.C1 + (long)m * p.ldc1 - p.n_split : (bf16_t*)p.C + (long)m * p.ldc;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n0 + wn * 128 + jn * 32 + 8 * a + 4 * lh;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float b = p.bias ? bf2f(((const bf16_t*)p.bias)[n + c]) : 0.f;
            v[c] = acc[a][c] * p.alpha + b;
            if (do_gelu) v[c] = w4_gelu_tanh(rbf(v[c]));
        }
        uint32_t o0 = pack2bf(v[0], v[1]), o1 = pack2bf(v[2], v[3]);
        if constexpr (GATED) {
            const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)p.res + (long)m * p.ldres + n);
            const uint2 gt = *reinterpret_cast<const uint2*>((const bf16_t*)p.gate + n);
            const uint32_t yy[2] = {o0, o1}, rr[2] = {r.x, r.y}, gg[2] = {gt.x, gt.y};
            uint32_t oo[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float y0f = bf2f((uint16_t)(yy[c] & 0xffff)), y1f = bf2f((uint16_t)(yy[c] >> 16));
                const float r0f = bf2f((uint16_t)(rr[c] & 0xffff)), r1f = bf2f((uint16_t)(rr[c] >> 16));
                const float g0f = bf2f((uint16_t)(gg[c] & 0xffff)), g1f = bf2f((uint16_t)(gg[c] >> 16));
                oo[c] = pack2bf(r0f + rbf(g0f * y0f), r1f + rbf(g1f * y1f));
            }
            o0 = oo[0]; o1 = oo[1];
        }
        *reinterpret_cast<uint2*>(crow + n) = make_uint2(o0, o1);
    }
}

extern "C" size_t utx_gemm_streamk_workspace_bytes_impl(void) {
    int dev = 0; hipDeviceProp_t pr;
    const int ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    return (size_t)ncu * 2 * 262144;      // up to two fp32 256 x 256 partial tiles per workgroup
}

// Split tail (kernel: "split tail"), the plan: the T = tiles % grid tiles of the last round are cut along K into S ranges each, T S ranges over the grid
// in ceil(T S / grid) passes, instead of T workgroups running a whole tile while the others idle.  S by a cost model in K-tiles (1.3 us), fitted to
// profiles/r02_gemm_streamk_check_v*.log: a range costs its K-tiles + 4 (dump), the fix-up kernel 6 + 8 per 256 partial tiles; unsplit, the round
// costs a tile's K-tiles.  UTX_GEMM_STREAMK: 0 never, 1 the model with a margin of 4 K-tiles, 2..999 that margin, 1000 + S force S ranges wherever
// the structure allows it (tests: ranges of one K-tile, ranges that start inside the LoRA segment, odd range lengths -- shapes the model never
// splits).  Pure host arithmetic; *T = *S = 0 when the last round stays whole.  At most 2 grid partial tiles (the workspace of
// utx_gemm_streamk_workspace_bytes).
extern "C" void utx_gemm_w4_split_plan(const GemmParams* pp, int tiles, int grid, int has_work, int* T_out, int* S_out) {
    const GemmParams& p = *pp;
    *T_out = 0; *S_out = 0;
    const int opt = g_utx_opt.gemm_streamk;
    if (opt <= 0 || !has_work || grid <= 0 || tiles <= grid || p.qk_cols != 0 || p.q_out != nullptr) return;      // (the fix-up kernel has neither the fused q / k nor the fp8-output epilogue)
    const int T = tiles % grid;
    if (T == 0) return;
    if (!(p.K2 == 0 || p.lora_n_limit <= 0 || p.lora_n_limit >= p.N)) return;       // every tail tile must have the same K extent
    const int nssu = p.K / 64 + ((p.K2 > 0 && p.lora_n_limit > 0) ? p.K2 / 64 : 0);
    if (opt >= 1000) {
        const int S = opt - 1000;
        if (S >= 2 && S <= 8 && nssu >= S && T * S <= 2 * grid) { *T_out = T; *S_out = S; }
        return;
    }
    const int margin = opt > 1 ? opt : 4;
    int best = nssu - margin, bestS = 0;
    for (int S = 2; S <= 8; ++S) {
        if (nssu / S < 8 || T * S > 2 * grid) break;
        const int passes = (T * S + grid - 1) / grid;
        const int est = passes * ((nssu + S - 1) / S + 4) + 6 + (8 * T * S) / 256;
        if (est < best) { best = est; bestS = S; }
    }
    if (bestS) { *T_out = T; *S_out = bestS; }
}

extern "C" int utx_launch_gemm_w4(GemmParams p, hipStream_t stream) {
    constexpr int LDS = 2 * W4_STAGE + 4 * 8192;     // the ring + the four waves' C staging buffers = all 160 KB
    UTX_ONCE_PER_DEVICE(attr_set) {
        const void* ks[2] = {reinterpret_cast<const void*>(gemm256_w4_kernel<false>), reinterpret_cast<const void*>(gemm256_w4_kernel<true>)};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    const int ncu = utx_ncu();
    const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
    // tile rows per L2 block of the tile order (an XCD's 32 consecutive tiles = group_m rows x 32 / group_m columns).  Re-swept in round 6 on the kernel as it is now
    // (nontemporal C stores, generated K loop: profiles/r06_gemm_group_m_sweep.log, _fine.log): outputs of >= 32 tile columns want the 8 x 4 block (N = 21 504: 4.654 ms against
    // 4.811 for round 2's choice of 2 -- with write-allocated C lines in the L2 the narrow block had won, r02_gemm_group_m_sweep.log), N = 3072 the narrow one (K = 15 360: 3.52 vs 3.56)
    int group_m = g_utx_opt.gemm_group_m > 0 ? g_utx_opt.gemm_group_m : (ntn >= 32 ? 8 : 2);
    if (group_m > ntm) group_m = ntm;
    p.ntn = ntn | (group_m << 16);
    const int tiles = ntm * ntn;
    int grid = tiles < ncu ? tiles : ncu;
    if (g_utx_opt.gemm_pers_grid > 0 && g_utx_opt.gemm_pers_grid < grid) grid = g_utx_opt.gemm_pers_grid;
    int sk_T = 0, sk_S = 0;
    utx_gemm_w4_split_plan(&p, tiles, grid, (p.sk_work && p.sk_work_bytes >= (size_t)2 * grid * 262144) ? 1 : 0, &sk_T, &sk_S);     // a plan holds at most 2 grid partial tiles
    if (p.mx8 == 2) {
        // MX fp8 operands, tile-packed scales (K, lda, ldb arrive in bf16 units = halved, as for the 128^2 form: the staging code is byte-identical)
        if (p.K2 > 0 || p.qk_cols > 0 || (p.K % 64) || !p.a_scale || !p.b_scale || p.lds_a < (p.M + 127) / 128 || p.lds_b < p.N / 128 ||
            ((uintptr_t)p.a_scale & 15) || ((uintptr_t)p.b_scale & 15)) return -2;
        if (p.gate && (p.gelu_from < p.N || p.n_split < p.N)) return -2;
        if (p.q_out && (p.gate || !p.qs_out || p.gelu_from >= p.N || (p.gelu_from % 128) || (p.ldq_out & 7) || p.ldq_out < p.N - p.gelu_from ||
                        p.qs_out_rb < (p.M + 127) / 128 || p.q_out_kt0 < 0 || ((uintptr_t)p.qs_out & 15) || ((uintptr_t)p.q_out & 7))) return -2;
        UTX_ONCE_PER_DEVICE(ax) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_w4_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_w4_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
            UTX_ONCE_DONE(ax);
        }
        if (p.gate) hipLaunchKernelGGL((gemm256_w4_kernel<true, false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
        else hipLaunchKernelGGL((gemm256_w4_kernel<false, false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
    } else if (g_utx_opt.gemm_fastk != 0) {      // UTX_GEMM_FASTK (round 6): the instances whose steady-state K loop is the generated stream; same bits
        UTX_ONCE_PER_DEVICE(af) {
            const void* kf[3] = {reinterpret_cast<const void*>(gemm256_w4_kernel<false, false, false, true>), reinterpret_cast<const void*>(gemm256_w4_kernel<true, false, false, true>),
                                 reinterpret_cast<const void*>(gemm256_w4_kernel<false, true, false, true>)};
            for (const void* k : kf)
                if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
            UTX_ONCE_DONE(af);
        }
        if (p.gate) {
            if (p.gelu_from < p.N || p.n_split < p.N) return -2;
            hipLaunchKernelGGL((gemm256_w4_kernel<true, false, false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
        } else if (p.qk_cols > 0) hipLaunchKernelGGL((gemm256_w4_kernel<false, true, false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
        else hipLaunchKernelGGL((gemm256_w4_kernel<false, false, false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
    } else if (p.gate) {
        if (p.gelu_from < p.N || p.n_split < p.N) return -2;
        hipLaunchKernelGGL((gemm256_w4_kernel<true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
    } else {
        if (p.qk_cols > 0) {
            UTX_ONCE_PER_DEVICE(aq) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_w4_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3; UTX_ONCE_DONE(aq); }
            hipLaunchKernelGGL((gemm256_w4_kernel<false, true>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
        } else
        hipLaunchKernelGGL((gemm256_w4_kernel<false>), dim3(grid), dim3(256), LDS, stream, p, tiles, 0, sk_T, sk_S);
    }
    if (sk_T > 0) {
        if (p.gate) hipLaunchKernelGGL((gemm_w4_fixup_kernel<true>), dim3(sk_T * 16), dim3(256), 0, stream, p, tiles, sk_T, sk_S);
        else hipLaunchKernelGGL((gemm_w4_fixup_kernel<false>), dim3(sk_T * 16), dim3(256), 0, stream, p, tiles, sk_T, sk_S);
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
