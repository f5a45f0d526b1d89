// Persistent bf16 MFMA GEMM for the large-M FLUX linears (gfx950): the 256x256x64 "8-phase" schedule of gemm.hip run as ONE
// continuous K-tile stream per workgroup.
//
//   C[M,N] = epi( alpha * ( A[M,K] . B[N,K]^T  +  A2[M,K2] . B2[N,K2]^T ) + bias[N] )      (same contract as gemm.hip)
//
// Why: the non-persistent kernel pays 7-12 us per 256x256 tile outside its K loop -- pipeline fill (prologue DMA), pipeline drain,
// and an epilogue that stages C through the operand ring behind two workgroup barriers per 128-row chunk -- against 1.4 us per
// K-tile, i.e. 10-18 % of a K = 3072 GEMM (profiles/r01_perf_gemm_8phase.log, tools/gemm_fixed_cost.py), which is exactly the
// margin hipBLASLt had on those shapes (profiles/r01_perf_gemm_vs_hipblaslt_v3.log).  Here:
//   * one workgroup per CU walks its tiles (w, w + G, w + 2G, ...; G = grid size; same XCD-aware L2-blocked order as gemm.hip);
//   * the LDS-DMA staging cursor runs 1-2 K-tiles ahead of the MFMA cursor and simply CONTINUES into the next tile: three
//     half-tiles stay in flight across the tile boundary, no fill, no drain;
//   * the epilogue never touches the operand ring and has NO barrier: each wave converts its own accumulators, exchanges half
//     rows between its two half-waves (v_permlane32_swap: every lane then owns 8 consecutive columns of one row = one 16-byte
//     store), and stores / reads the residual directly.  Bias and gate of the tile sit in a private 1 KB LDS slot per wave,
//     fetched by ONE LDS-DMA at the tile's first K-tile (they never enter the VGPR file before use, and do not disturb the
//     counted vmcnt waits).  Residual rows are read by inline-asm loads with counted waits, so hipcc never emits the vmcnt(0)
//     that a plain load next to in-flight LDS-DMA draws (cdna_hip_programming.md 5.7 / "Three .s-level traps" (b)).
// Arithmetic is identical to gemm.hip: same K order, same MFMA, y = bf16(alpha*acc + bias) -> GELU -> gated residual with the same
// rounding points, so the outputs are BIT-IDENTICAL to the three kernels there (tests/test_fullsize_gpu.py, tools/gemm_race_screen.py).
//
// LDS map (bytes): operand ring as in gemm.hip -- region r in {A0,A1,B0,B1} at r*32768, K-tile parity at +16384 (128 KB);
// epilogue slots at 131072 + wave*1024: [0,512) bias of the tile's 256 columns, [512,1024) gate.
// Hazards inside a K-tile are those of gemm256_8ph_kernel (DESIGN.md "GEMM schedule"); the tile boundary adds none: the stream of
// K-tiles is the same stream, the epilogue sits between the last barrier of a K-tile and the first of the next and contains no
// barrier, so both wave groups keep their barrier counts.
#include "common.h"
#include "kernels.h"

#define GP_BK 64

__device__ __forceinline__ float gp_gelu_tanh(float x) {   // same expression as gemm.hip (sigmoid form of the tanh approximation)
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

// LDS destination given as a byte address in the LDS aperture (M0 value): wave-uniform
__device__ __forceinline__ void gp_glds16(const void* g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) unsigned int gp_u32x4;

// residual rows: loads hipcc does not see (no vmcnt(0) beside the in-flight LDS-DMA); waited for by gp_wait_res<N>()
__device__ __forceinline__ gp_u32x4 gp_load16_asm(const void* p) {
    gp_u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
// counted wait for one residual load.  The wait names no register (a tied "+v" operand on the s_waitcnt itself lets the compiler place a copy of the
// register IN FRONT of the wait, i.e. read the load's destination while it is in flight -- gemm_w4.hip, W4_WAIT_VM); the destination is re-defined by an
// empty asm behind the wait and a scheduling fence, which pins every consumer below it.  tools/vmcnt_hazard_check.py audits the listing.
template <int N>
__device__ __forceinline__ void gp_wait_res(gp_u32x4& a) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0);
}

#define GP_BAR()                                      \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("" ::: "memory");                \
        __builtin_amdgcn_s_barrier();                 \
        asm volatile("" ::: "memory");                \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)

// GATED: the gated-residual epilogue (out = res + gate * y; no GELU, no column split in that mode) -- two instantiations keep each
// epilogue's register footprint inside the 256-register budget of a 512-thread workgroup.
// SCHED: where the eight DMAs of a K-tile are issued.  0 = two per phase (the schedule of gemm256_8ph_kernel); 1 = none in q0 (whose
// twelve fragment reads make it the longest load phase), three in q1, one in q2, four in q3: a K-tile lasts 2 * sum_p max(load_p, 256
// MFMA cycles), so the load phases should be EQUAL, not the DMA counts (profiles/r02_gemm_ab_v4.log).
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool GATED, int SCHED, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm256_pers_kernel(GemmParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    const int wid = xcd_remap(blockIdx.x, G);
#ifdef UTX_ABLATION
    const int dbg = p.ntn >> 24;   // timing ablations (libunitex_hip_ablate.so only; wrong results): 4 = no C stores, 8 = no epilogue at all
#else
    constexpr int dbg = 0;
#endif

    const int ntn = p.ntn & 0xffff, group_m = (p.ntn >> 16) & 0xff;
    const int ntm_ = (p.M + 255) / 256;
    const int per_group = group_m * ntn;
    const int nk1 = p.K / GP_BK, nk2 = p.K2 / GP_BK;
    const long ldaB = (long)p.lda * 2, ldbB = (long)p.ldb * 2, lda2B = (long)p.lda2 * 2, ldb2B = (long)p.ldb2 * 2;

    // tile id -> origin (XCD-aware, L2-blocked order: groups of group_m tile rows, tn-major inside a group)
#define GP_TILE_ORIGIN(w_, m0_, n0_)                                                       \
    do {                                                                                   \
        const int grp_ = (w_) / per_group, rem_ = (w_) - grp_ * per_group;                 \
        const int ftm_ = grp_ * group_m;                                                   \
        const int gs_ = (ntm_ - ftm_ < group_m) ? ntm_ - ftm_ : group_m;                   \
        const int tn_ = rem_ / gs_;                                                        \
        (m0_) = (ftm_ + rem_ - tn_ * gs_) * 256;                                           \
        (n0_) = tn_ * 256;                                                                 \
    } while (0)

    // ---- staging.  The load phases of the schedule (ds_reads + address arithmetic + DMA issue) are its critical path -- an MFMA
    // phase is 8 x 32 cycles, a load phase measured ~350 -- so the in-loop address work is cut to nothing: four running 64-bit
    // SGPR pointers (first DMA of each half-tile A0 / A1 / B0 / B1 at the cursor's K-tile) advanced by 128 B per K-tile, the second
    // DMA of a half-tile 64 rows further (one scalar add), ONE per-lane byte offset per operand (VGPR).  Everything else -- tile
    // origin, LoRA K-segment switch, ragged-M clamping -- happens at segment / tile events, once per 48+ K-tiles.
    const int srow_in = lane >> 3, sslot = lane & 7;
    const unsigned chunkb = (unsigned)((sslot ^ (((8 * wave + srow_in) >> 1) & 7)) << 4);
    // this wave's 1 KB slice of every 8 KB DMA block, as an LDS byte address.  Re-read through an opaque asm in every K-tile so the
    // sixteen destination addresses of a K-tile pair are formed by one scalar add each instead of living in (spilled) SGPRs.
    const unsigned ldst0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)wave * 1024u;
    int s_tile = wid, s_kt = 0, s_seg_end = 0, s_seg = 1, s_m0 = 0, s_n0 = 0;
    bool s_lora = false, s_ragged = false;
    const char *s_pA0 = nullptr, *s_pA1 = nullptr, *s_pB0 = nullptr, *s_pB1 = nullptr;
    unsigned s_rA = 0, s_rB = 0;            // row strides in bytes of the current K-segment
    unsigned voA = 0, voB = 0;              // per-lane byte offset inside an 8-row DMA block: srow_in * stride + swizzled chunk
#define GP_SEG1_SETUP()                                                                                    \
    do {                                                                                                   \
        s_rA = (unsigned)ldaB; s_rB = (unsigned)ldbB;                                                      \
        s_pA0 = (const char*)p.A + (long)(s_m0 + 8 * wave) * ldaB;  s_pA1 = s_pA0 + 128 * ldaB;            \
        s_pB0 = (const char*)p.B + (long)(s_n0 + 8 * wave) * ldbB;  s_pB1 = s_pB0 + 128 * ldbB;            \
        voA = (unsigned)srow_in * s_rA + chunkb; voB = (unsigned)srow_in * s_rB + chunkb;                  \
        s_kt = 0; s_seg_end = nk1; s_seg = 1;                                                              \
    } while (0)
#define GP_SEG2_SETUP()                                                                                    \
    do {                                                                                                   \
        s_rA = (unsigned)lda2B; s_rB = (unsigned)ldb2B;                                                    \
        s_pA0 = (const char*)p.A2 + (long)((s_n0 / p.lora_seg_n) * p.K2) * 2 + (long)(s_m0 + 8 * wave) * lda2B; s_pA1 = s_pA0 + 128 * lda2B; \
        s_pB0 = (const char*)p.B2 + (long)(s_n0 + 8 * wave) * ldb2B; s_pB1 = s_pB0 + 128 * ldb2B;          \
        voA = (unsigned)srow_in * s_rA + chunkb; voB = (unsigned)srow_in * s_rB + chunkb;                  \
        s_kt = 0; s_seg_end = nk2; s_seg = 2;                                                              \
    } while (0)
#define GP_STAGE_SETUP()                                                                                   \
    do {                                                                                                   \
        GP_TILE_ORIGIN(s_tile, s_m0, s_n0);                                                                \
        s_lora = (p.K2 > 0) && (s_n0 < p.lora_n_limit);                                                    \
        s_ragged = (s_m0 + 256 > p.M);                                                                     \
        GP_SEG1_SETUP();                                                                                   \
    } while (0)
    // stage half-tile h of operand A (isb = 0) / B (isb = 1) of the cursor's K-tile into LDS parity par_
#define GP_STAGE(isb_, h_, par_)                                                                            \
    do {                                                                                                    \
        const char* p0_ = (isb_) ? ((h_) ? s_pB1 : s_pB0) : ((h_) ? s_pA1 : s_pA0);                         \
        const unsigned rs_ = (isb_) ? s_rB : s_rA;                                                          \
        const unsigned vo_ = (isb_) ? voB : voA;                                                            \
        const unsigned l_ = ldst + (2 * (isb_) + (h_)) * 32768 + (par_) * 16384;                            \
        if (!(isb_) && s_ragged) {   /* last tile row of a ragged M: rows >= M re-read row M-1 (never stored) */ \
            const int r_ = s_m0 + 8 * wave + 128 * (h_) + srow_in;                                          \
            gp_glds16(p0_ + vo_ - (long)((r_ > p.M - 1) ? r_ - (p.M - 1) : 0) * rs_, l_);                   \
            gp_glds16(p0_ + 64 * (long)rs_ + vo_ - (long)((r_ + 64 > p.M - 1) ? r_ + 64 - (p.M - 1) : 0) * rs_, l_ + 8192); \
        } else {                                                                                            \
            gp_glds16(p0_ + vo_, l_);                                                                       \
            gp_glds16(p0_ + 64 * (unsigned long)rs_ + vo_, l_ + 8192);                                      \
        }                                                                                                   \
    } while (0)
    // one of the two DMAs (d_ = 0: rows 0-63 of the half-tile's slice, 1: rows 64-127) -- for schedules that spread them over phases
#define GP_STAGE1(isb_, h_, d_, par_)                                                                       \
    do {                                                                                                    \
        const char* p0_ = (isb_) ? ((h_) ? s_pB1 : s_pB0) : ((h_) ? s_pA1 : s_pA0);                         \
        const unsigned rs_ = (isb_) ? s_rB : s_rA;                                                          \
        const unsigned vo_ = (isb_) ? voB : voA;                                                            \
        const unsigned l_ = ldst + (2 * (isb_) + (h_)) * 32768 + (par_) * 16384 + 8192 * (d_);              \
        if (!(isb_) && s_ragged) {                                                                          \
            const int r_ = s_m0 + 8 * wave + 128 * (h_) + 64 * (d_) + srow_in;                              \
            gp_glds16(p0_ + (d_) * 64 * (long)rs_ + vo_ - (long)((r_ > p.M - 1) ? r_ - (p.M - 1) : 0) * rs_, l_); \
        } else {                                                                                            \
            gp_glds16(p0_ + (d_) * 64 * (unsigned long)rs_ + vo_, l_);                                      \
        }                                                                                                   \
    } while (0)
#define GP_STAGE_ADVANCE()                                                                 \
    do {                                                                                   \
        s_pA0 += GP_BK * 2; s_pA1 += GP_BK * 2; s_pB0 += GP_BK * 2; s_pB1 += GP_BK * 2;    \
        ++s_kt;                                                                            \
        if (s_kt == s_seg_end) {                                                           \
            if (s_seg == 1 && s_lora) {                                                    \
                GP_SEG2_SETUP();                                                           \
            } else {                                                                       \
                s_tile += G;                                                               \
                if (s_tile < ntiles) GP_STAGE_SETUP();                                     \
            }                                                                              \
        }                                                                                  \
    } while (0)
#define GP_SVALID (s_tile < ntiles)

    // compute cursor
    int c_tile = wid, c_kt = 0, c_nk = 0, c_m0 = 0, c_n0 = 0;
#define GP_COMPUTE_SETUP()                                                                 \
    do {                                                                                   \
        GP_TILE_ORIGIN(c_tile, c_m0, c_n0);                                                \
        c_nk = nk1 + (((p.K2 > 0) && (c_n0 < p.lora_n_limit)) ? nk2 : 0);                  \
        c_kt = 0;                                                                          \
    } while (0)

    f32x16 acc[2][4];   // [j][2i+f], swapped MFMA: rows = n, cols = m
#define GP_ZERO_ACC()                                                       \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                        \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;
    GP_ZERO_ACC()

    // fragment read bases: row*128 + ((2kk + lh) ^ swz) * 16, swz = (row>>1)&7 = (l31>>1)&7 for every fragment
    const int swz = (l31 >> 1) & 7;
    const int arow = (wr * 64 + l31) * 128, brow = 65536 + (wc * 32 + l31) * 128;
    const int x0 = ((0 + lh) ^ swz) << 4, x1 = ((2 + lh) ^ swz) << 4, x2 = ((4 + lh) ^ swz) << 4, x3 = ((6 + lh) ^ swz) << 4;
    const char* const fa0 = smem + arow + x0; const char* const fa1 = smem + arow + x1;
    const char* const fa2 = smem + arow + x2; const char* const fa3 = smem + arow + x3;
    const char* const fb0 = smem + brow + x0; const char* const fb1 = smem + brow + x1;
    const char* const fb2 = smem + brow + x2; const char* const fb3 = smem + brow + x3;
#define GP_LD(ptr_, off_) (*reinterpret_cast<const bf16x8*>((ptr_) + (off_)))
#define GP_LOAD_B(dst_, j_, par_)                                                      \
    dst_##0 = GP_LD(fb0, (j_) * 32768 + (par_) * 16384); dst_##1 = GP_LD(fb1, (j_) * 32768 + (par_) * 16384); \
    dst_##2 = GP_LD(fb2, (j_) * 32768 + (par_) * 16384); dst_##3 = GP_LD(fb3, (j_) * 32768 + (par_) * 16384);
#define GP_LOAD_A(i_, par_)                                                            \
    a00 = GP_LD(fa0, (i_) * 32768 + (par_) * 16384);        a01 = GP_LD(fa1, (i_) * 32768 + (par_) * 16384);        \
    a02 = GP_LD(fa2, (i_) * 32768 + (par_) * 16384);        a03 = GP_LD(fa3, (i_) * 32768 + (par_) * 16384);        \
    a10 = GP_LD(fa0, (i_) * 32768 + (par_) * 16384 + 4096); a11 = GP_LD(fa1, (i_) * 32768 + (par_) * 16384 + 4096); \
    a12 = GP_LD(fa2, (i_) * 32768 + (par_) * 16384 + 4096); a13 = GP_LD(fa3, (i_) * 32768 + (par_) * 16384 + 4096);
    // M16 -- timing experiment (ablation build only, wrong results): every 32x32x16 MFMA replaced by two 16x16x32 MFMAs on the same
    // operand registers: same FLOPs, same register / LDS traffic, the instruction shape hipBLASLt uses (tools/mfma_power_probe.py)
#define GP_MM(acc_, b_, a_)                                                                                     \
    if constexpr (M16) {                                                                                        \
        f32x4_t lo_ = {acc_[0], acc_[1], acc_[2], acc_[3]}, hi_ = {acc_[4], acc_[5], acc_[6], acc_[7]};           \
        lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_, a_, lo_, 0, 0, 0);                                    \
        hi_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, hi_, 0, 0, 0);                                    \
        acc_[0] = lo_[0]; acc_[1] = lo_[1]; acc_[2] = lo_[2]; acc_[3] = lo_[3];                                 \
        acc_[4] = hi_[0]; acc_[5] = hi_[1]; acc_[6] = hi_[2]; acc_[7] = hi_[3];                                 \
    } else acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_, a_, acc_, 0, 0, 0);
#define GP_MFMA(b_, j_, i_)                                                                                     \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        GP_MM(acc[j_][2 * (i_)], b_##0, a00)                              \
        GP_MM(acc[j_][2 * (i_) + 1], b_##0, a10)                      \
        GP_MM(acc[j_][2 * (i_)], b_##1, a01)                              \
        GP_MM(acc[j_][2 * (i_) + 1], b_##1, a11)                      \
        GP_MM(acc[j_][2 * (i_)], b_##2, a02)                              \
        GP_MM(acc[j_][2 * (i_) + 1], b_##2, a12)                      \
        GP_MM(acc[j_][2 * (i_)], b_##3, a03)                              \
        GP_MM(acc[j_][2 * (i_) + 1], b_##3, a13)                      \
        __builtin_amdgcn_s_setprio(0);                                                                          \
    } while (0)
    bf16x8 a00, a01, a02, a03, a10, a11, a12, a13;       // a{f}{kk}: current A piece (64 rows x 64 k)
    bf16x8 bz0, bz1, bz2, bz3, bo0, bo1, bo2, bo3;       // B pieces j = 0 (bz) and j = 1 (bo)

    // ---- epilogue operands of a tile: bias [256] and gate [256] of the tile's columns -> this wave's private LDS slot, by ONE
    // LDS-DMA (lanes 0-31 fetch the bias, lanes 32-63 the gate; 16 B each).  Issued at the first K-tile of the tile BEFORE that
    // K-tile's own staging, so the K-tile's counted wait retires it; read only by this wave, after that wait.
    char* const eslot = smem + 131072 + wave * 1024;   // (only its LDS address is used: the slot is read and written by inline asm)
    const bf16_t* const ebias = p.bias ? (const bf16_t*)p.bias : (const bf16_t*)p.B;    // always a readable address
    const bf16_t* const egate = GATED ? (const bf16_t*)p.gate : ebias;
    // The DMA is issued from inline asm (M0 = LDS destination, saved / restored in the same statement: cdna_hip_programming.md 5.7):
    // hipcc models an LDS-DMA it can see as a pending LDS write and would put `s_waitcnt vmcnt(0)` -- a drain of the whole staging
    // pipeline -- in front of every ds_read of the slot in the epilogue.  Completion is guaranteed by this K-tile's own counted
    // wait (the DMA is older than the K-tile's staging); two exec-masked statements, lanes 0-31 bias, lanes 32-63 gate.
    const unsigned eslot_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)eslot);
#define GP_EPI_DMA(src_)                                                                                          \
    do {                                                                                                          \
        unsigned keep_;                                                                                           \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(eslot_lds) : "memory");                                      \
    } while (0)
#define GP_EPI_FETCH()                                                                     \
    do {                                                                                   \
        const unsigned lo_ = (unsigned)(l31 << 4);                                         \
        if (lh == 0) GP_EPI_DMA((const char*)(ebias + c_n0) + lo_);                        \
        else GP_EPI_DMA((const char*)(egate + c_n0) + lo_);                                \
    } while (0)

    // one K-tile (4 phases); par_ is a literal so every LDS offset is an immediate
#define GP_KTILE(par_)                                                                             \
    do {                                                                                           \
        unsigned ldst = ldst0;                                                                     \
        asm volatile("" : "+s"(ldst));                                                             \
        /* q0 */                                                                                   \
        GP_LOAD_B(bz, 0, par_)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        GP_LOAD_A(0, par_)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (c_kt == 0) GP_EPI_FETCH();                                                             \
        if (SCHED == 0) { if (GP_SVALID) { GP_STAGE(0, 1, (par_) ^ 1); GP_STAGE_ADVANCE(); } }     \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                         \
        GP_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        GP_MFMA(bz, 0, 0);                                                                         \
        GP_BAR();                                                                                  \
        /* q1 */                                                                                   \
        GP_LOAD_B(bo, 1, par_)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (SCHED == 0) { if (GP_SVALID) GP_STAGE(1, 0, par_); }                                   \
        else {                                                                                     \
            if (GP_SVALID) { GP_STAGE(0, 1, (par_) ^ 1); GP_STAGE_ADVANCE(); }                     \
            if (GP_SVALID) GP_STAGE1(1, 0, 0, par_);                                               \
        }                                                                                          \
        GP_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        GP_MFMA(bo, 1, 0);                                                                         \
        GP_BAR();                                                                                  \
        /* q2 */                                                                                   \
        GP_LOAD_A(1, par_)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (SCHED == 0) { if (GP_SVALID) GP_STAGE(0, 0, par_); }                                   \
        else { if (GP_SVALID) GP_STAGE1(1, 0, 1, par_); }                                          \
        GP_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        GP_MFMA(bo, 1, 1);                                                                         \
        GP_BAR();                                                                                  \
        /* q3 */                                                                                   \
        if (GP_SVALID) {                                                                           \
            if (SCHED != 0) GP_STAGE(0, 0, par_);                                                  \
            GP_STAGE(1, 1, par_);                                                                  \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                       \
        } else {                                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        }                                                                                          \
        GP_BAR();                                                                                  \
        GP_MFMA(bz, 0, 1);                                                                         \
        GP_BAR();                                                                                  \
    } while (0)

    // ---- direct epilogue of the compute cursor's tile.
    // lane (l31, lh): acc[j][2i+f][4a+c] -> n = n0 + j*128 + wc*32 + 8a + 4lh + c ;  m = m0 + i*128 + wr*64 + f*32 + l31.
    // After packing, the groups a = 2q and a = 2q+1 are exchanged between the half-waves (v_permlane32_swap, T21): lanes 0-31 then
    // hold columns 16q .. 16q+7, lanes 32-63 columns 16q+8 .. 16q+15 of their row -> one 16-byte store each.
    const bf16_t* const pres = (const bf16_t*)p.res;
    // Epilogue operand reads are inline asm too (one batched statement, its own lgkmcnt wait): a ds_read hipcc can see draws a
    // `s_waitcnt vmcnt(0)` whenever LDS-DMA is in flight -- it cannot tell the operand ring from the epilogue slot.
    typedef __attribute__((ext_vector_type(2))) unsigned int gp_u32x2;
#define GP_READ_BIAS()                                                                                                     \
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:32\n\tds_read_b64 %3, %8 offset:48\n\t" \
                 "ds_read_b64 %4, %8 offset:256\n\tds_read_b64 %5, %8 offset:272\n\tds_read_b64 %6, %8 offset:288\n\t"    \
                 "ds_read_b64 %7, %8 offset:304\n\ts_waitcnt lgkmcnt(0)"                                                    \
                 : "=&v"(bb00), "=&v"(bb01), "=&v"(bb02), "=&v"(bb03), "=&v"(bb10), "=&v"(bb11), "=&v"(bb12), "=&v"(bb13)   \
                 : "v"(eslot_lds + (unsigned)((wc * 32 + 4 * elh) * 2)) : "memory")
#define GP_READ_GATE()                                                                                                     \
    asm volatile("ds_read_b128 %0, %4 offset:512\n\tds_read_b128 %1, %4 offset:544\n\tds_read_b128 %2, %4 offset:768\n\t"  \
                 "ds_read_b128 %3, %4 offset:800\n\ts_waitcnt lgkmcnt(0)"                                                   \
                 : "=&v"(gg0), "=&v"(gg1), "=&v"(gg2), "=&v"(gg3)                                                           \
                 : "v"(eslot_lds + (unsigned)((wc * 32 + 8 * elh) * 2)) : "memory")

    // ---- step A of a piece: y = bf16(alpha * acc + bias) (+ GELU), packed, half rows exchanged between the half-waves.
    // piece = row combination k (i = k >> 1, f = k & 1; rows rowb + 128 i + 32 f) x column piece pi = 2j + q.
    // Instruction diet (the epilogue's VALU work, not its stores, was the per-tile fixed cost: tools/gemm_pers_probe.py): the bias
    // of a column piece is unpacked once and reused by the four row combinations; without GELU the value goes from the fma
    // straight into v_cvt_pk_bf16_f32 (rounding once to bf16 is what rounding to bf16 and packing does); GELU and plain tiles
    // take separate straight-line paths (the branch is per tile, never per value).
#define GP_BIAS8(pi_)                                                                                                  \
    float b8_[8];                                                                                                      \
    {                                                                                                                  \
        const gp_u32x2 lo_ = (pi_) == 0 ? bb00 : (pi_) == 1 ? bb02 : (pi_) == 2 ? bb10 : bb12;   /* a = 2q   */        \
        const gp_u32x2 hi_ = (pi_) == 0 ? bb01 : (pi_) == 1 ? bb03 : (pi_) == 2 ? bb11 : bb13;   /* a = 2q+1 */        \
        b8_[0] = bf2f((uint16_t)(lo_[0] & 0xffff)); b8_[1] = bf2f((uint16_t)(lo_[0] >> 16));                           \
        b8_[2] = bf2f((uint16_t)(lo_[1] & 0xffff)); b8_[3] = bf2f((uint16_t)(lo_[1] >> 16));                           \
        b8_[4] = bf2f((uint16_t)(hi_[0] & 0xffff)); b8_[5] = bf2f((uint16_t)(hi_[0] >> 16));                           \
        b8_[6] = bf2f((uint16_t)(hi_[1] & 0xffff)); b8_[7] = bf2f((uint16_t)(hi_[1] >> 16));                           \
        if (!p.bias) { _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) b8_[c_] = 0.f; }                               \
    }
#define GP_PACK_CORE(k_, pi_, O_, GELU_)                                                                               \
    {                                                                                                                  \
        constexpr int j_ = (pi_) >> 1, q_ = (pi_) & 1;                                                                 \
        float v_[8];                                                                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) {                                                             \
            v_[c_] = acc[j_][k_][8 * q_ + c_] * p.alpha + b8_[c_];                                                     \
            if (GELU_) v_[c_] = gp_gelu_tanh(rbf(v_[c_]));                                                             \
        }                                                                                                              \
        const uint32_t w00_ = pack2bf(v_[0], v_[1]), w01_ = pack2bf(v_[2], v_[3]);                                     \
        const uint32_t w10_ = pack2bf(v_[4], v_[5]), w11_ = pack2bf(v_[6], v_[7]);                                     \
        auto s0_ = __builtin_amdgcn_permlane32_swap(w00_, w10_, false, false);                                         \
        auto s1_ = __builtin_amdgcn_permlane32_swap(w01_, w11_, false, false);                                         \
        O_[0] = s0_[0]; O_[1] = s1_[0]; O_[2] = s0_[1]; O_[3] = s1_[1];                                                \
    }
#define GP_STORE_PIECE(k_, pi_, O_)                                                                                    \
    if (rowb + 32 * ((k_) & 1) + 128 * ((k_) >> 1) < p.M && !(dbg & 4))                                                \
        *reinterpret_cast<uint4*>(cbase + (long)(rowb + 32 * ((k_) & 1) + 128 * ((k_) >> 1)) * ldc + ccol + 128 * ((pi_) >> 1) + 16 * ((pi_) & 1)) = \
            make_uint4(O_[0], O_[1], O_[2], O_[3]);
    // plain epilogue: column piece outer (bias unpacked once), row combination inner; pack + store, one piece at a time
#define GP_PLAIN_PIECE(k_, pi_, GELU_)                                                                                 \
    if ((k_) < nv) {                                                                                                   \
        gp_u32x4 o_;                                                                                                   \
        GP_PACK_CORE(k_, pi_, o_, GELU_)                                                                               \
        GP_STORE_PIECE(k_, pi_, o_)                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
#define GP_PLAIN_COLS(pi_, GELU_)                                                                                      \
    {                                                                                                                  \
        GP_BIAS8(pi_)                                                                                                  \
        GP_PLAIN_PIECE(0, pi_, GELU_) GP_PLAIN_PIECE(1, pi_, GELU_) GP_PLAIN_PIECE(2, pi_, GELU_) GP_PLAIN_PIECE(3, pi_, GELU_) \
    }
#define GP_PACK_PIECE(k_, pi_, O_) { GP_BIAS8(pi_) GP_PACK_CORE(k_, pi_, O_, false) }

    // gated epilogue.  The 16 residual pieces of the wave (64 VGPRs -- the K loop's operand fragments are dead here) are requested
    // up front (12 at once, the last 4 as soon as packing the first row combination has freed their registers), so the whole 128 KB
    // residual tile of the workgroup is in flight under ONE memory latency; meanwhile every piece is packed in place (128
    // accumulator registers shrink to 64), then pieces are finished in request order behind a counted wait: piece n has 15 - n
    // younger loads and n younger stores, always 15.  (Named registers, not arrays: hipcc demotes an array
    // written under a runtime branch to scratch.)
#define GP_RES_LOAD(k_, pi_, R_)                                                                           \
    {                                                                                                      \
        const int rr_ = rowb + 32 * ((k_) & 1) + 128 * ((k_) >> 1);                                        \
        R_ = gp_load16_asm(pres + (long)(rr_ > p.M - 1 ? p.M - 1 : rr_) * p.ldres + gcol + 128 * ((pi_) >> 1) + 16 * ((pi_) & 1)); \
    }
#define GP_GATE_PIECE(k_, pi_, R_, Y_, full_)                                                                          \
    {                                                                                                                  \
        if (full_) gp_wait_res<15>(R_); else gp_wait_res<0>(R_);                                                       \
        const gp_u32x4 graw_ = (pi_) == 0 ? gg0 : (pi_) == 1 ? gg1 : (pi_) == 2 ? gg2 : gg3;                           \
        gp_u32x4 o_;                                                                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                             \
            const float y0_ = bf2f((uint16_t)(Y_[c_] & 0xffff)), y1_ = bf2f((uint16_t)(Y_[c_] >> 16));                 \
            const float r0_ = bf2f((uint16_t)(R_[c_] & 0xffff)), r1_ = bf2f((uint16_t)(R_[c_] >> 16));                 \
            const float g0_ = bf2f((uint16_t)(graw_[c_] & 0xffff)), g1_ = bf2f((uint16_t)(graw_[c_] >> 16));           \
            o_[c_] = pack2bf(r0_ + rbf(g0_ * y0_), r1_ + rbf(g1_ * y1_));                                              \
        }                                                                                                              \
        GP_STORE_PIECE(k_, pi_, o_)                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
#define GP_FOR_COMBO0(M_) M_(0, 0, rq00, yy00) M_(0, 1, rq01, yy01) M_(0, 2, rq02, yy02) M_(0, 3, rq03, yy03)
#define GP_FOR_COMBO1(M_) M_(1, 0, rq10, yy10) M_(1, 1, rq11, yy11) M_(1, 2, rq12, yy12) M_(1, 3, rq13, yy13)
#define GP_FOR_COMBO2(M_) M_(2, 0, rq20, yy20) M_(2, 1, rq21, yy21) M_(2, 2, rq22, yy22) M_(2, 3, rq23, yy23)
#define GP_FOR_COMBO3(M_) M_(3, 0, rq30, yy30) M_(3, 1, rq31, yy31) M_(3, 2, rq32, yy32) M_(3, 3, rq33, yy33)
#define GP_FOR_PIECES(M_) GP_FOR_COMBO0(M_) GP_FOR_COMBO1(M_) GP_FOR_COMBO2(M_) GP_FOR_COMBO3(M_)
#define GP_M_LOAD(k_, pi_, R_, Y_) GP_RES_LOAD(k_, pi_, R_)
#define GP_M_PACK(k_, pi_, R_, Y_) { GP_PACK_PIECE(k_, pi_, Y_) __builtin_amdgcn_sched_barrier(0); }
#define GP_M_GATE(k_, pi_, R_, Y_) GP_GATE_PIECE(k_, pi_, R_, Y_, true)
    // ragged wave (some of its rows are >= M): one piece at a time, every load waited for on the spot -- rare, not pipelined
#define GP_M_SLOW(k_, pi_, R_, Y_)                                                                                     \
    if ((k_) < nv) {                                                                                                   \
        gp_u32x4 y_, r_;                                                                                               \
        GP_PACK_PIECE(k_, pi_, y_)                                                                                     \
        GP_RES_LOAD(k_, pi_, r_)                                                                                       \
        GP_GATE_PIECE(k_, pi_, r_, y_, false)                                                                          \
    }

    // Row combinations k = 2i + f are increasing in their first row; nv = how many of them have a row < M for this wave (ragged
    // last tile row): the others issue NOTHING and a valid combination always stores (its first row is in range), so the op
    // counts behind the waits are exact for nv == 4; a ragged wave (nv < 4) simply waits for everything (vmcnt(0)).
    // Loads and stores of gfx9-family parts retire in order through the one vmcnt counter, older LDS-DMA included.
#define GP_EPILOGUE()                                                                                      \
    do {                                                                                                   \
        /* per-lane epilogue addressing starts from values the compiler cannot see through: nothing of it is hoisted out of the \
           tile loop into long-lived registers (the K loop has none to spare) */                            \
        int el31 = l31, elh = lh;                                                                          \
        asm volatile("" : "+v"(el31), "+v"(elh));                                                          \
        const bool do_gelu = !GATED && c_n0 >= p.gelu_from;                                                \
        const bool to_c1 = !GATED && c_n0 >= p.n_split;                                                    \
        bf16_t* const cbase = to_c1 ? (bf16_t*)p.C1 : (bf16_t*)p.C;                                        \
        const long ldc = to_c1 ? p.ldc1 : p.ldc;                                                           \
        const int rowb = c_m0 + wr * 64 + el31;                                                            \
        const int gcol = c_n0 + wc * 32 + 8 * elh;                                                         \
        const int ccol = gcol - (to_c1 ? p.n_split : 0);                                                   \
        const int wrow0 = c_m0 + wr * 64;                       /* wave-uniform first row of combination 0 */ \
        const int nv = (wrow0 + 160 < p.M) ? 4 : (wrow0 + 128 < p.M) ? 3 : (wrow0 + 32 < p.M) ? 2 : (wrow0 < p.M) ? 1 : 0; \
        gp_u32x2 bb00, bb01, bb02, bb03, bb10, bb11, bb12, bb13;                                           \
        if constexpr (GATED) {                                                                             \
            gp_u32x4 gg0, gg1, gg2, gg3;                                                                   \
            if (nv == 4) {                                                                                 \
                gp_u32x4 rq00, rq01, rq02, rq03, rq10, rq11, rq12, rq13, rq20, rq21, rq22, rq23, rq30, rq31, rq32, rq33; \
                gp_u32x4 yy00, yy01, yy02, yy03, yy10, yy11, yy12, yy13, yy20, yy21, yy22, yy23, yy30, yy31, yy32, yy33; \
                GP_FOR_COMBO0(GP_M_LOAD) GP_FOR_COMBO1(GP_M_LOAD) GP_FOR_COMBO2(GP_M_LOAD)                 \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                GP_READ_BIAS();                                                                            \
                GP_FOR_COMBO0(GP_M_PACK)                                                                   \
                GP_FOR_COMBO3(GP_M_LOAD)       /* the last 16 registers become free once combination 0 is packed */ \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                GP_FOR_COMBO1(GP_M_PACK) GP_FOR_COMBO2(GP_M_PACK) GP_FOR_COMBO3(GP_M_PACK)                 \
                GP_READ_GATE();                                                                            \
                GP_FOR_PIECES(GP_M_GATE)                                                                   \
            } else {                                                                                       \
                GP_READ_BIAS();                                                                            \
                GP_READ_GATE();                                                                            \
                GP_FOR_PIECES(GP_M_SLOW)                                                                   \
            }                                                                                              \
        } else {                                                                                           \
            GP_READ_BIAS();                                                                                \
            if (do_gelu) { GP_PLAIN_COLS(0, true) GP_PLAIN_COLS(1, true) GP_PLAIN_COLS(2, true) GP_PLAIN_COLS(3, true) } \
            else { GP_PLAIN_COLS(0, false) GP_PLAIN_COLS(1, false) GP_PLAIN_COLS(2, false) GP_PLAIN_COLS(3, false) } \
        }                                                                                                  \
        GP_ZERO_ACC()                                                                                      \
    } while (0)

    // ---- prologue: K-tile 0 of the first tile complete, B0 / A0 / B1 of the next K-tile in flight
    const unsigned ldst = ldst0;
    if (GP_SVALID) {
        GP_STAGE_SETUP();
        GP_STAGE(1, 0, 0); GP_STAGE(0, 0, 0); GP_STAGE(1, 1, 0); GP_STAGE(0, 1, 0);
        GP_STAGE_ADVANCE();
    }
    if (GP_SVALID) {
        GP_STAGE(1, 0, 1); GP_STAGE(0, 0, 1); GP_STAGE(1, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GP_COMPUTE_SETUP();
    GP_BAR();
    if (wr == 1) GP_BAR();          // stagger: the wr = 1 waves run one barrier behind
    // Two unrolled K-tile bodies (LDS parity is an immediate in every ds_read / DMA address), ONE epilogue body: a tile with an
    // odd number of K-tiles (LoRA segment: 48 + 1) leaves the stream at odd parity, `odd` re-enters the loop at the second body.
    bool odd = false;
    for (;;) {
        bool tile_done = false;
        if (!odd) {
            GP_KTILE(0);
            ++c_kt;
            if (c_kt == c_nk) { tile_done = true; odd = true; }
        }
        if (!tile_done) {
            GP_KTILE(1);
            ++c_kt;
            odd = false;
            if (c_kt != c_nk) continue;
        }
        // Tile boundary.  The wr = 1 waves run one barrier behind the wr = 0 waves; left alone, the two groups' epilogues would
        // serialise (each waits at its next barrier for the other to finish).  One extra barrier per group re-aligns them for the
        // epilogue -- group 0 waits for group 1's last MFMA cluster (~0.2 us), both convert and store at the same time, group 1
        // then lets group 0 go one barrier ahead again.  Both groups execute one extra s_barrier per tile: the counts stay equal.
        if (wr == 0) GP_BAR();
        if (!(dbg & 8)) GP_EPILOGUE();
        if (wr == 1) GP_BAR();
        c_tile += G;
        if (c_tile >= ntiles) break;
        GP_COMPUTE_SETUP();
    }
    if (wr == 0) GP_BAR();          // re-align the barrier count
}

extern "C" int utx_launch_gemm_pers(GemmParams p, hipStream_t stream) {
    constexpr int LDS = 131072 + 8 * 1024;
    UTX_ONCE_PER_DEVICE(attr_set) {
        const void* ks[4] = {reinterpret_cast<const void*>(gemm256_pers_kernel<false, 0>), reinterpret_cast<const void*>(gemm256_pers_kernel<true, 0>),
                             reinterpret_cast<const void*>(gemm256_pers_kernel<false, 1>), reinterpret_cast<const void*>(gemm256_pers_kernel<true, 1>)};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    const int ncu = utx_ncu();
    const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
    int group_m = g_utx_opt.gemm_group_m > 0 ? g_utx_opt.gemm_group_m : 4;
    if (group_m > ntm) group_m = ntm;
    p.ntn = ntn | (group_m << 16) | (g_utx_opt.gemm_debug_abl << 24);
    const int tiles = ntm * ntn;
    int grid = tiles < ncu ? tiles : ncu;
    if (g_utx_opt.gemm_pers_grid > 0 && g_utx_opt.gemm_pers_grid < grid) grid = g_utx_opt.gemm_pers_grid;   // A/B: fewer workgroups than CUs
    // DMA placement: 0 = two per phase, 1 = none beside the twelve fragment reads of q0 (3 / 1 / 4 in q1 / q2 / q3).  Interleaved A/B
    // (profiles/r02_gemm_ab_v4.log): 1 wins +3.5...+6 % on the long-K shapes (K = 12288 / 15360) and +0.4...+0.8 % on N >= 12288,
    // loses 1.5-2.7 % on the short narrow ones; UTX_GEMM_PERS_SCHED = 1 / 2 forces schedule 0 / 1.
    int sched = (p.K + p.K2 >= 6144 || p.N >= 12288) ? 1 : 0;
    if (g_utx_opt.gemm_pers_sched == 1) sched = 0;
    if (g_utx_opt.gemm_pers_sched == 2) sched = 1;
#ifdef UTX_ABLATION
    if ((g_utx_opt.gemm_debug_abl & 16) && !p.gate) {
        UTX_ONCE_PER_DEVICE(a2) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_pers_kernel<false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3; UTX_ONCE_DONE(a2); }
        hipLaunchKernelGGL((gemm256_pers_kernel<false, 1, true>), dim3(grid), dim3(512), LDS, stream, p, tiles);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
#endif
    if (p.gate) {
        if (p.gelu_from < p.N || p.n_split < p.N) return -2;     // the gated epilogue carries neither GELU nor the column split
        if (sched == 0) hipLaunchKernelGGL((gemm256_pers_kernel<true, 0>), dim3(grid), dim3(512), LDS, stream, p, tiles);
        else hipLaunchKernelGGL((gemm256_pers_kernel<true, 1>), dim3(grid), dim3(512), LDS, stream, p, tiles);
    } else {
        if (sched == 0) hipLaunchKernelGGL((gemm256_pers_kernel<false, 0>), dim3(grid), dim3(512), LDS, stream, p, tiles);
        else hipLaunchKernelGGL((gemm256_pers_kernel<false, 1>), dim3(grid), dim3(512), LDS, stream, p, tiles);
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
