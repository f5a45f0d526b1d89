// bf16 MFMA GEMM with fused epilogues for the FLUX DiT linears (gfx950).
//
//   C[M,N] = epi( alpha * ( A[M,K] . B[N,K]^T  +  A2[M,K2] . B2[N,K2]^T ) + bias[N] )
//
// A = activations (token-major), B = torch Linear weight layout [out, in] -> both operands are
// K-contiguous.  The optional second K-segment (A2/B2) is the LoRA path: A2 = s * (x A_lora^T)
// (rank padded to 64), B2 = B_lora; it is consumed by the SAME K-loop / accumulators, i.e. the
// LoRA up-projection is fused into the base GEMM instead of being a second GEMM + add
// (peft semantics: /root/reference/pipeline.py:108-112,245,263; targets
// /root/reference/flux_piplines/texturing/trainer.py:282-295).
//
// Epilogues (all applied on y = bf16(alpha*acc + bias), mirroring the bf16 tensor boundary of a
// bf16 torch Linear):
//   - GELU(tanh) on columns n >= gelu_from              (FeedForward 'gelu-approximate', proj_mlp)
//   - gated residual  out = res + gate[n] * y            (AdaLN-Zero gates of the FLUX blocks)
//   - column split: n >= n_split is written to a second buffer C1[m, n - n_split]
//                                                        (single-stream block: [qkv | mlp] in one GEMM)
//
// Structure (template): BM x BN x 64 tile, NWM x NWN waves, each wave (BM/NWM) x (BN/NWN) in 32x32x16 MFMAs,
// operands staged HBM/L2 -> LDS with global_load_lds (16 B / lane, LDS image linear, XOR swizzle applied on
// the SOURCE address and on the ds_read address), 2-deep LDS ring, one barrier per K-step, row pointers hoisted
// out of the K loop.  MFMA is issued "swapped" (A-operand = weight rows) so each lane owns 4 consecutive n for
// one m; C goes through LDS so global stores / residual loads are full rows.
//
// Two instantiations:
//   256x256 (8 waves, 128x64 per wave, 128 KB LDS, 1 workgroup / CU)  -- the large-M FLUX linears
//   128x128 (4 waves,  64x64 per wave,  64 KB LDS, 2 workgroups / CU) -- small M (text stream), ragged shapes
// Why 256: round-1 ablation (profiles/r01_perf_gemm_ablation.log): with the MFMAs removed the 128^2 kernel
// takes as long as with them -- it is bound by the L2 -> LDS fill rate (~10 TB/s chip-wide for LDS-DMA), i.e.
// by bytes staged per flop: 1/64 B/flop at 128^2, 1/128 B/flop at 256^2.
// Tile order: XCD-aware and L2-blocked (groups of GM_GROUP_M tile rows, tn-major) so the tiles resident on an
// XCD share A/B panels in its 4 MB L2 (row-major order measured 52 % L2 miss, 8.4 GB fetched for 0.5 GB).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define GM_BK 64
#define GM_GROUP_M 4

// GELU (tanh approximation, diffusers FeedForward 'gelu-approximate' [3p]):  0.5 x (1 + tanh u) == x * sigmoid(2u),
// u = sqrt(2/pi) (x + 0.044715 x^3).  Evaluated in the sigmoid form with the hardware exp / rcp (~8 instructions) instead
// of libm tanhf (~40): the result is rounded to bf16 right after, and the GELU columns cost +14 us per 256x256 tile with
// tanhf (profiles/r01_perf_gemm_8phase.log).  Limits are exact: exp -> inf gives x * 0, exp -> 0 gives x.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}


// Epilogue row pass shared by the GEMM kernels: full-row 16-byte stores from the C staging image, gated residual fused.
// The 8 residual rows of a chunk live in NAMED registers (rq0..rq7) requested together before the chunk is processed:
// a load placed next to its store is serialised behind vmcnt(0) (res may alias C, the compiler cannot hoist it -- each
// thread only re-reads what it alone writes, so hoisting by hand is safe), which made the gated epilogue 8 serial HBM
// round trips per chunk (profiles/r01_perf_gemm_8phase.log: 19.6 us fixed cost per tile vs 9.5 us plain).  Named scalars
// because hipcc demotes arrays written under a runtime branch to scratch.
#define GM_RES_LOAD1(it_, rowexpr_) rq##it_ = *reinterpret_cast<const uint4*>(pres + (long)(rowexpr_) * p.ldres + gn0);
#define GM_ROW_PASS1(it_, guard_)                                                                                \
    {                                                                                                            \
        const int ml = rrow0 + RPP * (it_);                                                                      \
        const int gm = m0 + 128 * chunk + ml;                                                                    \
        if (guard_) {                                                                                            \
            uint4 yv = *reinterpret_cast<const uint4*>(smem + ml * CROW + rslot * 16);                           \
            if (pgate) {                                                                                         \
                uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};                                                       \
                const uint32_t rw[4] = {rq##it_.x, rq##it_.y, rq##it_.z, rq##it_.w};                             \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                  \
                    const float y0 = bf2f((uint16_t)(yw[c] & 0xffff)), y1 = bf2f((uint16_t)(yw[c] >> 16));       \
                    const float r0 = bf2f((uint16_t)(rw[c] & 0xffff)), r1 = bf2f((uint16_t)(rw[c] >> 16));       \
                    const float o0 = r0 + rbf(gv[2 * c] * y0);                                                   \
                    const float o1 = r1 + rbf(gv[2 * c + 1] * y1);                                               \
                    yw[c] = pack2bf(o0, o1);                                                                     \
                }                                                                                                \
                yv = make_uint4(yw[0], yw[1], yw[2], yw[3]);                                                     \
            }                                                                                                    \
            *reinterpret_cast<uint4*>(cbase + (long)gm * ldc + cn) = yv;                                         \
        }                                                                                                        \
    }

// MX8: the base K-segment is OCP MX fp8 (e4m3 elements, one E8M0 scale per 32 elements along K) on BOTH operands, consumed by
// v_mfma_scale_f32_32x32x64_f8f6f4 at twice the bf16 MFMA rate (BASELINE configs[4] "fp8 MFMA weights").  A 128-byte LDS row is
// then 128 fp8 (one K-tile of 128) instead of 64 bf16: staging, swizzle and ring are byte-identical (the launcher passes K and the
// leading dimensions in bf16 units, i.e. halved); only the fragment reads (32 bytes per lane and MFMA) and the per-lane scale
// bytes differ.  The LoRA K-segment stays bf16 and feeds the same fp32 accumulators.
typedef __attribute__((ext_vector_type(8))) int gm_i32x8;
template <int BM, int BN, int NWM, int NWN, bool CONV, bool MX8 = false>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void gemm_bf16_kernel(GemmParams p) {
    constexpr int NW = NWM * NWN, NT = 64 * NW;
    constexpr int WMR = BM / NWM, WNR = BN / NWN;   // rows of C per wave along m / n
    constexpr int MI = WMR / 32, NI = WNR / 32;
    constexpr int A_BYTES = BM * GM_BK * 2, B_BYTES = BN * GM_BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int CROW = BN * 2 + 16;              // epilogue C row stride in bytes (pad: <= 2-way write conflicts)
    static_assert(BM / 8 / NW == 4 && BN / 8 / NW == 4, "4 A + 4 B global_load_lds per thread per K-step");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* pA = (const bf16_t*)p.A; const bf16_t* pB = (const bf16_t*)p.B;
    const bf16_t* pA2 = (const bf16_t*)p.A2; const bf16_t* pB2 = (const bf16_t*)p.B2;
    const bf16_t* pbias = (const bf16_t*)p.bias; const bf16_t* pgate = (const bf16_t*)p.gate;
    const bf16_t* pres = (const bf16_t*)p.res;
    bf16_t* pC = (bf16_t*)p.C; bf16_t* pC1 = (bf16_t*)p.C1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- XCD-aware, L2-blocked tile order (see header)
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int ntn = p.ntn & 0xffff, group_m = (p.ntn >> 16) & 0xff;
    const int ntm_ = (p.M + BM - 1) / BM;
    const int per_group = group_m * ntn;
    const int grp = w / per_group, rem = w - grp * per_group;
    const int first_tm = grp * group_m;
    const int gsize = (ntm_ - first_tm < group_m) ? ntm_ - first_tm : group_m;
    const int tm = first_tm + rem % gsize, tn = rem / gsize;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk1 = p.K / GM_BK;
    const bool lora = (p.K2 > 0) && (n0 < p.lora_n_limit);
    const int nk = nk1 + (lora ? p.K2 / GM_BK : 0);
    const long a2_off = lora ? (long)(n0 / p.lora_seg_n) * p.K2 : 0;

    // ---- glds source addressing: wave-instruction i covers tile rows 8i..8i+7 (1 KB of LDS).
    // Row pointers live in named VGPR pairs and advance by BK elements per K-step; re-based once when the loop
    // enters the LoRA K-segment.  (No lambda / arrays: hipcc spills by-reference pointer sets to scratch.)
    const int srow_in = lane >> 3, sslot = lane & 7;
#define GM_SRC(j_)                                                                                       \
    const int srow##j_ = 8 * (wave + NW * (j_)) + srow_in;                                              \
    const int schunk##j_ = (sslot ^ ((srow##j_ >> 1) & 7)) * 8;                                          \
    const int sgm##j_ = (m0 + srow##j_ > p.M - 1) ? p.M - 1 : m0 + srow##j_;                             \
    const int sgn##j_ = (n0 + srow##j_ > p.N - 1) ? p.N - 1 : n0 + srow##j_;                             \
    const bf16_t* pa##j_ = pA + (long)sgm##j_ * p.lda + schunk##j_;                                      \
    const bf16_t* pb##j_ = pB + (long)sgn##j_ * p.ldb + schunk##j_;
    GM_SRC(0) GM_SRC(1) GM_SRC(2) GM_SRC(3)
    // CONV (implicit 3x3 convolution, see unitex_hip.h): A row m is output pixel (oy, ox); the source row of K-step kt
    // is input pixel (oy*stride + ky - pad, ox*stride + kx - pad) of tap = kt*64 / Cin, or the zero page.
    const bf16_t* pzero = (const bf16_t*)p.zero_page;
    const int chlim = CONV ? (p.conv_Hi << p.conv_up) : 0, cwlim = CONV ? (p.conv_Wi << p.conv_up) : 0;
#define GM_CONV_PIX(j_)                                                                                  \
    const int coy##j_ = CONV ? (sgm##j_ / p.conv_Wo) * p.conv_stride - p.conv_pad : 0;                   \
    const int cox##j_ = CONV ? (sgm##j_ % p.conv_Wo) * p.conv_stride - p.conv_pad : 0;
    GM_CONV_PIX(0) GM_CONV_PIX(1) GM_CONV_PIX(2) GM_CONV_PIX(3)
#define GM_CONV_PTR(j_)                                                                                  \
    {                                                                                                    \
        const int iy_ = coy##j_ + ky_, ix_ = cox##j_ + kx_;                                              \
        const bool ok_ = (iy_ >= 0) & (iy_ < chlim) & (ix_ >= 0) & (ix_ < cwlim);                        \
        const long src_ = (((long)(iy_ >> p.conv_up) * p.conv_Wi + (ix_ >> p.conv_up)) << p.conv_cin_log2) + c0_; \
        pa##j_ = (ok_ ? pA + src_ : pzero) + schunk##j_;                                                 \
    }
#define GM_REBASE(j_)                                                                                    \
    pa##j_ = pA2 + a2_off + (long)sgm##j_ * p.lda2 + schunk##j_;                                         \
    pb##j_ = pB2 + (long)sgn##j_ * p.ldb2 + schunk##j_;
#define GM_STAGE1(j_)                                                                                    \
    glds16(pa##j_, sa_ + (wave + NW * (j_)) * 1024); pa##j_ += GM_BK;                                    \
    glds16(pb##j_, sb_ + (wave + NW * (j_)) * 1024); pb##j_ += GM_BK;
#define GM_STAGE(kt_, buf_)                                                                              \
    do {                                                                                                 \
        if constexpr (CONV) {                                                                            \
            const int kk0_ = (kt_) * GM_BK;                                                              \
            const int tap_ = kk0_ >> p.conv_cin_log2, c0_ = kk0_ & ((1 << p.conv_cin_log2) - 1);         \
            const int ky_ = tap_ / 3, kx_ = tap_ - 3 * ky_;                                              \
            GM_CONV_PTR(0) GM_CONV_PTR(1) GM_CONV_PTR(2) GM_CONV_PTR(3)                                  \
        }                                                                                                \
        if ((kt_) == nk1) { GM_REBASE(0) GM_REBASE(1) GM_REBASE(2) GM_REBASE(3) }                        \
        char* sa_ = smem + (buf_) * STAGE;                                                               \
        char* sb_ = sa_ + A_BYTES;                                                                       \
        GM_STAGE1(0) GM_STAGE1(1) GM_STAGE1(2) GM_STAGE1(3)                                              \
    } while (0)

    f32x16 acc[NI][MI];  // swapped MFMA: rows = n, cols = m
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane fragment read offsets: row r, chunk c -> r*128 + ((c ^ ((r>>1)&7)) << 4).  Rows of one wave differ
    // by multiples of 32, which leaves (r>>1)&7 unchanged: ONE swizzle term per operand, the other fragments are
    // immediate offsets (i * 32 rows * 128 B).
    const int ra0 = wm * WMR + l31, rb0 = wn * WNR + l31;
    const int aoff0 = ra0 * 128, aswz0 = (ra0 >> 1) & 7;
    const int boff0 = rb0 * 128 + A_BYTES, bswz0 = (rb0 >> 1) & 7;
    // MX8: E8M0 scale bytes of this lane's fragment rows, one dword (4 blocks of 32 = one K-tile of 128) per row and K-tile,
    // fetched one K-tile ahead; MFMA step kk spans blocks 2 kk (scale supplied by lanes 0-31) and 2 kk + 1 (lanes 32-63): byte 2 kk + lh
    // of the dword -> pre-shifted by 8 lh, op_sel 2 kk
    uint32_t sa_cur[MI], sb_cur[NI], sa_nxt[MI], sb_nxt[NI];
    const uint8_t* sa_row[MI]; const uint8_t* sb_row[NI];
    if constexpr (MX8) {
#pragma unroll
        for (int i = 0; i < MI; ++i) { int r = m0 + ra0 + 32 * i; if (r > p.M - 1) r = p.M - 1; sa_row[i] = (const uint8_t*)p.a_scale + (long)r * p.lds_a; }
#pragma unroll
        for (int i = 0; i < NI; ++i) { int r = n0 + rb0 + 32 * i; if (r > p.N - 1) r = p.N - 1; sb_row[i] = (const uint8_t*)p.b_scale + (long)r * p.lds_b; }
#pragma unroll
        for (int i = 0; i < MI; ++i) sa_nxt[i] = *reinterpret_cast<const uint32_t*>(sa_row[i]);
#pragma unroll
        for (int i = 0; i < NI; ++i) sb_nxt[i] = *reinterpret_cast<const uint32_t*>(sb_row[i]);
    }
    GM_STAGE(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) GM_STAGE(kt + 1, buf ^ 1);
        const char* st = smem + buf * STAGE;
        if constexpr (MX8) {
            if (kt < nk1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) sa_cur[i] = sa_nxt[i] >> (8 * lh);
#pragma unroll
                for (int i = 0; i < NI; ++i) sb_cur[i] = sb_nxt[i] >> (8 * lh);
                if (kt + 1 < nk1) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) sa_nxt[i] = *reinterpret_cast<const uint32_t*>(sa_row[i] + 4 * (kt + 1));
#pragma unroll
                    for (int i = 0; i < NI; ++i) sb_nxt[i] = *reinterpret_cast<const uint32_t*>(sb_row[i] + 4 * (kt + 1));
                }
#define GM_MX_STEP(kk_)                                                                                              \
                {                                                                                                    \
                    gm_i32x8 af8[MI], bf8[NI];                                                                       \
                    /* operand layout of the instruction (tools/mx_probe.py): lane (row, h) holds k = 16h .. 16h+15 of scale   \
                       block 0 in bytes 0-15 and k = 32 + 16h .. of scale block 1 in bytes 16-31; block 0's scale comes from   \
                       lane row, block 1's from lane row + 32 */                                                     \
                    const int c0_ = 4 * (kk_) + lh, c1_ = c0_ + 2;                                                   \
                    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                 \
                        const uint4 lo_ = *reinterpret_cast<const uint4*>(st + boff0 + i * 4096 + ((c0_ ^ bswz0) << 4));       \
                        const uint4 hi_ = *reinterpret_cast<const uint4*>(st + boff0 + i * 4096 + ((c1_ ^ bswz0) << 4)); \
                        bf8[i] = gm_i32x8{(int)lo_.x, (int)lo_.y, (int)lo_.z, (int)lo_.w, (int)hi_.x, (int)hi_.y, (int)hi_.z, (int)hi_.w}; \
                    }                                                                                                \
                    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                 \
                        const uint4 lo_ = *reinterpret_cast<const uint4*>(st + aoff0 + i * 4096 + ((c0_ ^ aswz0) << 4));       \
                        const uint4 hi_ = *reinterpret_cast<const uint4*>(st + aoff0 + i * 4096 + ((c1_ ^ aswz0) << 4)); \
                        af8[i] = gm_i32x8{(int)lo_.x, (int)lo_.y, (int)lo_.z, (int)lo_.w, (int)hi_.x, (int)hi_.y, (int)hi_.z, (int)hi_.w}; \
                    }                                                                                                \
                    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                \
                        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                            \
                            acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf8[ni], af8[mi], acc[ni][mi], 0, 0, 2 * (kk_), \
                                                                                          (int)sb_cur[ni], 2 * (kk_), (int)sa_cur[mi]);     \
                }
                GM_MX_STEP(0)
                GM_MX_STEP(1)
                continue;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const char* pbk = st + boff0 + (((2 * kk + lh) ^ bswz0) << 4);
            const char* pak = st + aoff0 + (((2 * kk + lh) ^ aswz0) << 4);
#pragma unroll
            for (int i = 0; i < NI; ++i) bfr[i] = *reinterpret_cast<const bf16x8*>(pbk + i * 4096);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(pak + i * 4096);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }

    // ---- epilogue, in chunks of 128 tile rows through LDS
    const int rslot = tid % (BN / 8);           // 16-byte chunk within the tile row
    const int rrow0 = tid / (BN / 8);
    constexpr int RPP = NT / (BN / 8);          // rows stored per pass
    const int gn0 = n0 + rslot * 8;
    float gv[8];
    if (pgate && gn0 < p.N) {
        const uint4 graw = *reinterpret_cast<const uint4*>(pgate + gn0);
        const uint32_t gw[4] = {graw.x, graw.y, graw.z, graw.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { gv[2 * c] = bf2f((uint16_t)(gw[c] & 0xffff)); gv[2 * c + 1] = bf2f((uint16_t)(gw[c] >> 16)); }
    }
    bf16_t* cbase; long ldc; int cn;
    if (gn0 >= p.n_split) { cbase = pC1; ldc = p.ldc1; cn = gn0 - p.n_split; }
    else { cbase = pC; ldc = p.ldc; cn = gn0; }

    // every bias word of this lane is requested up front (a load inside the loops below is followed by vmcnt(0))
    uint2 bq[NI][4];
    {
        const bf16_t* bsrc = pbias ? pbias : pB;   // always a readable address: no branch around the array writes
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                int gnb = n0 + wn * WNR + ni * 32 + 8 * a + 4 * lh; if (gnb > p.N - 4) gnb = (p.N >= 4) ? p.N - 4 : 0;
                bq[ni][a] = *reinterpret_cast<const uint2*>(bsrc + gnb);
            }
    }
    static_assert(128 / RPP == 8, "8 row passes per 128-row chunk");
#define GM_RROW(it_) ((m0 + 128 * chunk + rrow0 + RPP * (it_) > p.M - 1) ? p.M - 1 : m0 + 128 * chunk + rrow0 + RPP * (it_))
#pragma unroll
    for (int chunk = 0; chunk < BM / 128; ++chunk) {
        uint4 rq0, rq1, rq2, rq3, rq4, rq5, rq6, rq7;
        if (pgate && gn0 < p.N) {
            GM_RES_LOAD1(0, GM_RROW(0)) GM_RES_LOAD1(1, GM_RROW(1)) GM_RES_LOAD1(2, GM_RROW(2)) GM_RES_LOAD1(3, GM_RROW(3))
            GM_RES_LOAD1(4, GM_RROW(4)) GM_RES_LOAD1(5, GM_RROW(5)) GM_RES_LOAD1(6, GM_RROW(6)) GM_RES_LOAD1(7, GM_RROW(7))
        }
        __syncthreads();  // operand ring (or the previous chunk's C image) no longer needed
        // phase 1: y = bf16(alpha*acc + bias) (+GELU) -> LDS [128][CROW]
        // lane (m = l31, h): acc[ni][mi][r] -> n = wn*WNR + ni*32 + (r&3) + 8(r>>2) + 4h ; m = wm*WMR + mi*32 + l31
        if ((wm * WMR) / 128 == chunk) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int nl = wn * WNR + ni * 32 + 8 * a + 4 * lh;  // local n of 4 consecutive columns
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (pbias) {
                        const uint2 braw = bq[ni][a];
                        bv[0] = bf2f((uint16_t)(braw.x & 0xffff)); bv[1] = bf2f((uint16_t)(braw.x >> 16));
                        bv[2] = bf2f((uint16_t)(braw.y & 0xffff)); bv[3] = bf2f((uint16_t)(braw.y >> 16));
                    }
                    const bool do_gelu = (n0 + nl) >= p.gelu_from;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        float y[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = rbf(acc[ni][mi][4 * a + c] * p.alpha + bv[c]);
                            if (do_gelu) v = gelu_tanh(v);
                            y[c] = v;
                        }
                        uint2 o; o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]);
                        const int ml = wm * WMR + mi * 32 + l31 - 128 * chunk;
                        *reinterpret_cast<uint2*>(smem + ml * CROW + nl * 2) = o;
                    }
                }
            }
        }
        __syncthreads();
        // phase 2: full-row stores; gated residual fused here
        if (gn0 < p.N) {
            GM_ROW_PASS1(0, gm < p.M) GM_ROW_PASS1(1, gm < p.M) GM_ROW_PASS1(2, gm < p.M) GM_ROW_PASS1(3, gm < p.M)
            GM_ROW_PASS1(4, gm < p.M) GM_ROW_PASS1(5, gm < p.M) GM_ROW_PASS1(6, gm < p.M) GM_ROW_PASS1(7, gm < p.M)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves, "8-phase" schedule (4 phases per K-tile): the large-M FLUX linears.
//
// Each wave (wr = wave>>2, wc = wave&3) owns output rows {i*128 + wr*64 + [0,64)} and columns
// {j*128 + wc*32 + [0,32)}, i, j in {0,1}: one 64-row piece of EACH A half-tile and one 32-column piece of
// EACH B half-tile.  A K-tile is staged as four 16 KB half-tiles (A0, A1, B0, B1; 2 global_load_lds per
// thread each) and consumed as four quadrant phases:
//     q0: read B0 (4 ds_read_b128), A0 (8)   MFMA (a0,b0)      stage A1 of K-tile t+1
//     q1: read B1 (4)                        MFMA (a0,b1)      stage B0 of K-tile t+2   (same LDS buffer as t)
//     q2: read A1 (8)                        MFMA (a1,b1)      stage A0 of K-tile t+2
//     q3: --                                 MFMA (a1,b0)      stage B1 of K-tile t+2 ; s_waitcnt vmcnt(6)
// Every phase is [ds_reads, 2 glds] barrier [8 MFMA] barrier, and the wr=1 waves run one barrier behind the
// wr=0 waves, so on each SIMD one wave is in its MFMA cluster while the other issues LDS reads / DMA.
// Three half-tiles (6 glds per thread) stay in flight across every barrier; the only vmcnt wait of the loop
// is the counted one in q3, which retires K-tile t+1 one full phase before its first read.
// LDS hazards (derivation in DESIGN.md "GEMM schedule"):
//   RAW  a half-tile is read >= 1 phase after the q3 wait that retired it and after a barrier both wave
//        groups passed;
//   WAR  a region is re-staged >= 2 phases after its last ds_read, except B0 (1 phase), whose reads are
//        retired by the s_waitcnt lgkmcnt(8) BEFORE the first barrier of q0 (B reads are issued first).
// LDS map (bytes): region r in {A0,A1,B0,B1} at r*32768, K-tile parity at +16384: every fragment read is
// base VGPR + immediate.

#define G8_BAR()                                      \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("" ::: "memory");                \
        __builtin_amdgcn_s_barrier();                 \
        asm volatile("" ::: "memory");                \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)

// tail split (launcher: launch_gemm8, opt-in UTX_GEMM_TAILSPLIT=1): the tiles of the last, partly filled round are cut along
// K into `ks` workgroups each.  Every split writes its fp32 accumulators (lane-linear) to `part` with agent-scope (write-through)
// stores, takes a ticket, and the LAST one to arrive sums all ks partials in split order (deterministic) and runs the normal
// epilogue -- no second kernel, no duplicated epilogue, and no agent-scope fence (an L2 write-back + invalidate on this part,
// measured -20 %, profiles/r01_perf_gemm_tailsplit_negative.log).
struct G8Split { int ks; int tile_base; float* part; int* tick; };

__global__ __launch_bounds__(512, 2) void gemm256_8ph_kernel(GemmParams p, G8Split sp) {
    constexpr int BM = 256, BN = 256, NT = 512;
    constexpr int CROW = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* pbias = (const bf16_t*)p.bias; const bf16_t* pgate = (const bf16_t*)p.gate;
    const bf16_t* pres = (const bf16_t*)p.res;
    bf16_t* pC = (bf16_t*)p.C; bf16_t* pC1 = (bf16_t*)p.C1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;

    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = sp.ks;                                  // 1: one workgroup per tile
    const int item = ks > 1 ? wid / ks : wid;
    const int split = wid - item * ks;
    const int w = sp.tile_base + item;
    const int ntn = p.ntn & 0xffff, group_m = (p.ntn >> 16) & 0xff;
#ifdef UTX_ABLATION
    const int dbg = p.ntn >> 24;   // timing ablation (libunitex_hip_ablate.so only; wrong results): 1 = no operand staging, 2 = always stage K-tile 0
#else
    constexpr int dbg = 0;
#endif
    const int ntm_ = (p.M + BM - 1) / BM;
    const int per_group = group_m * ntn;
    const int grp = w / per_group, rem = w - grp * per_group;
    const int first_tm = grp * group_m;
    const int gsize = (ntm_ - first_tm < group_m) ? ntm_ - first_tm : group_m;
    const int tm = first_tm + rem % gsize, tn = rem / gsize;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk1 = p.K / GM_BK;
    const bool lora = (p.K2 > 0) && (n0 < p.lora_n_limit);
    const int nk_all = nk1 + (lora ? p.K2 / GM_BK : 0);
    // K-tiles [kb, kb + nk) of the concatenated (K ++ K2) range belong to this workgroup (all of them without a split)
    int kb = 0, nk = nk_all;
    if (ks > 1) {
        const int q = nk_all / ks, rm = nk_all - q * ks;
        kb = split * q + (split < rm ? split : rm);
        nk = q + (split < rm ? 1 : 0);
    }
    const long a2_off = lora ? (long)(n0 / p.lora_seg_n) * p.K2 : 0;

    // ---- staging sources: wave-uniform base (SGPR) + one per-lane byte offset per operand and K-segment.
    // glds instruction (wave, j) of a half-tile covers its rows 8*(wave + 8j) .. +7  (1 KB of LDS, lane-linear).
    const int srow_in = lane >> 3, sslot = lane & 7;
    const unsigned chunkb = (unsigned)((sslot ^ (((8 * wave + srow_in) >> 1) & 7)) << 4);
    const long ldaB = (long)p.lda * 2, ldbB = (long)p.ldb * 2, lda2B = (long)p.lda2 * 2, ldb2B = (long)p.ldb2 * 2;
    const unsigned voA1 = (unsigned)(srow_in * ldaB) + chunkb, voB1 = (unsigned)(srow_in * ldbB) + chunkb;
    const unsigned voA2 = (unsigned)(srow_in * lda2B) + chunkb, voB2 = (unsigned)(srow_in * ldb2B) + chunkb;
    const char* ubA1 = (const char*)p.A + (long)(m0 + 8 * wave) * ldaB;
    const char* ubB1 = (const char*)p.B + (long)(n0 + 8 * wave) * ldbB;
    const char* ubA2 = (const char*)p.A2 + a2_off * 2 + (long)(m0 + 8 * wave) * lda2B;
    const char* ubB2 = (const char*)p.B2 + (long)(n0 + 8 * wave) * ldb2B;
    char* const ldst = smem + wave * 1024;
    const bool ragged_m = (m0 + BM > p.M);                 // wave-uniform
    const int arow_lane = m0 + 8 * wave + srow_in;         // first A row this lane stages (half-tile 0, first DMA)
    // stage half-tile h of operand A (isb = 0) / B (isb = 1) of K-tile t_
#define G8_STAGE(t_, isb_, h_)                                                                              \
    do {                                                                                                    \
        const int tt_ = (t_);                                                                               \
        if (dbg == 1) break;                                                                                \
        const int ts_ = (dbg == 2) ? 0 : tt_ + kb;                                                          \
        const bool s2_ = ts_ >= nk1;                                                                        \
        const long rs_ = (isb_) ? (s2_ ? ldb2B : ldbB) : (s2_ ? lda2B : ldaB);                              \
        const char* ub_ = ((isb_) ? (s2_ ? ubB2 : ubB1) : (s2_ ? ubA2 : ubA1)) +                            \
                          (long)(s2_ ? ts_ - nk1 : ts_) * (GM_BK * 2) + (long)(128 * (h_)) * rs_;           \
        const unsigned vo_ = (isb_) ? (s2_ ? voB2 : voB1) : (s2_ ? voA2 : voA1);                            \
        char* l_ = ldst + (2 * (isb_) + (h_)) * 32768 + (tt_ & 1) * 16384;                                  \
        const char* g0_ = ub_ + vo_;                                                                        \
        const char* g1_ = ub_ + 64 * rs_ + vo_;                                                             \
        if (!(isb_) && ragged_m) {   /* last tile row of a ragged M: rows >= M re-read row M-1 (never stored) */ \
            const int r_ = arow_lane + 128 * (h_);                                                          \
            g0_ -= (long)((r_ > p.M - 1) ? r_ - (p.M - 1) : 0) * rs_;                                       \
            g1_ -= (long)((r_ + 64 > p.M - 1) ? r_ + 64 - (p.M - 1) : 0) * rs_;                             \
        }                                                                                                   \
        glds16((const bf16_t*)g0_, l_);                                                                     \
        glds16((const bf16_t*)g1_, l_ + 8192);                                                              \
    } while (0)

    f32x16 acc[2][4];   // [j][2i+f], swapped MFMA: rows = n, cols = m
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment read bases: row*128 + ((2kk + lh) ^ swz) * 16, swz = (row>>1)&7 = (l31>>1)&7 for every fragment
    const int swz = (l31 >> 1) & 7;
    const int arow = (wr * 64 + l31) * 128, brow = 65536 + (wc * 32 + l31) * 128;
    const int x0 = ((0 + lh) ^ swz) << 4, x1 = ((2 + lh) ^ swz) << 4, x2 = ((4 + lh) ^ swz) << 4, x3 = ((6 + lh) ^ swz) << 4;
    const char* const fa0 = smem + arow + x0; const char* const fa1 = smem + arow + x1;
    const char* const fa2 = smem + arow + x2; const char* const fa3 = smem + arow + x3;
    const char* const fb0 = smem + brow + x0; const char* const fb1 = smem + brow + x1;
    const char* const fb2 = smem + brow + x2; const char* const fb3 = smem + brow + x3;
#define G8_LD(ptr_, off_) (*reinterpret_cast<const bf16x8*>((ptr_) + (off_)))
#define G8_LOAD_B(dst_, j_, par_)                                                      \
    dst_##0 = G8_LD(fb0, (j_) * 32768 + (par_) * 16384); dst_##1 = G8_LD(fb1, (j_) * 32768 + (par_) * 16384); \
    dst_##2 = G8_LD(fb2, (j_) * 32768 + (par_) * 16384); dst_##3 = G8_LD(fb3, (j_) * 32768 + (par_) * 16384);
#define G8_LOAD_A(i_, par_)                                                            \
    a00 = G8_LD(fa0, (i_) * 32768 + (par_) * 16384);        a01 = G8_LD(fa1, (i_) * 32768 + (par_) * 16384);        \
    a02 = G8_LD(fa2, (i_) * 32768 + (par_) * 16384);        a03 = G8_LD(fa3, (i_) * 32768 + (par_) * 16384);        \
    a10 = G8_LD(fa0, (i_) * 32768 + (par_) * 16384 + 4096); a11 = G8_LD(fa1, (i_) * 32768 + (par_) * 16384 + 4096); \
    a12 = G8_LD(fa2, (i_) * 32768 + (par_) * 16384 + 4096); a13 = G8_LD(fa3, (i_) * 32768 + (par_) * 16384 + 4096);
#define G8_MFMA(b_, j_, i_)                                                                                     \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        acc[j_][2 * (i_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##0, a00, acc[j_][2 * (i_)], 0, 0, 0);          \
        acc[j_][2 * (i_) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##0, a10, acc[j_][2 * (i_) + 1], 0, 0, 0);  \
        acc[j_][2 * (i_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##1, a01, acc[j_][2 * (i_)], 0, 0, 0);          \
        acc[j_][2 * (i_) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##1, a11, acc[j_][2 * (i_) + 1], 0, 0, 0);  \
        acc[j_][2 * (i_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##2, a02, acc[j_][2 * (i_)], 0, 0, 0);          \
        acc[j_][2 * (i_) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##2, a12, acc[j_][2 * (i_) + 1], 0, 0, 0);  \
        acc[j_][2 * (i_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##3, a03, acc[j_][2 * (i_)], 0, 0, 0);          \
        acc[j_][2 * (i_) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_##3, a13, acc[j_][2 * (i_) + 1], 0, 0, 0);  \
        __builtin_amdgcn_s_setprio(0);                                                                          \
    } while (0)
    bf16x8 a00, a01, a02, a03, a10, a11, a12, a13;       // a{f}{kk}: current A piece (64 rows x 64 k)
    bf16x8 bz0, bz1, bz2, bz3, bo0, bo1, bo2, bo3;       // B pieces j = 0 (bz) and j = 1 (bo)

    // one K-tile (4 phases); par_ = t & 1 is a literal so every LDS offset is an immediate
#define G8_KTILE(t_, par_)                                                                         \
    do {                                                                                           \
        const int t__ = (t_);                                                                      \
        /* q0 */                                                                                   \
        G8_LOAD_B(bz, 0, par_)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        G8_LOAD_A(0, par_)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (t__ + 1 < nk) G8_STAGE(t__ + 1, 0, 1);                                                 \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                         \
        G8_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        G8_MFMA(bz, 0, 0);                                                                         \
        G8_BAR();                                                                                  \
        /* q1 */                                                                                   \
        G8_LOAD_B(bo, 1, par_)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (t__ + 2 < nk) G8_STAGE(t__ + 2, 1, 0);                                                 \
        G8_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        G8_MFMA(bo, 1, 0);                                                                         \
        G8_BAR();                                                                                  \
        /* q2 */                                                                                   \
        G8_LOAD_A(1, par_)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (t__ + 2 < nk) G8_STAGE(t__ + 2, 0, 0);                                                 \
        G8_BAR();                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        G8_MFMA(bo, 1, 1);                                                                         \
        G8_BAR();                                                                                  \
        /* q3 */                                                                                   \
        if (t__ + 2 < nk) {                                                                        \
            G8_STAGE(t__ + 2, 1, 1);                                                               \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                       \
        } else {                                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        }                                                                                          \
        G8_BAR();                                                                                  \
        G8_MFMA(bz, 0, 1);                                                                         \
        G8_BAR();                                                                                  \
    } while (0)

    // ---- prologue: K-tile 0 complete, B0/A0/B1 of K-tile 1 in flight
    G8_STAGE(0, 1, 0); G8_STAGE(0, 0, 0); G8_STAGE(0, 1, 1); G8_STAGE(0, 0, 1);
    if (nk > 1) {
        G8_STAGE(1, 1, 0); G8_STAGE(1, 0, 0); G8_STAGE(1, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G8_BAR();
    if (wr == 1) G8_BAR();          // stagger: the wr = 1 waves run one barrier behind
    for (int t = 0; t < nk; t += 2) {
        G8_KTILE(t, 0);
        if (t + 1 < nk) G8_KTILE(t + 1, 1);
    }
    if (wr == 0) G8_BAR();          // re-align the barrier count

    if (ks > 1) {
        float* mine = sp.part + ((long)item * ks + split) * 65536 + tid;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + ((j * 4 + i) * 16 + r) * 512, acc[j][i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the write-through stores have been acknowledged
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) *flag = (__hip_atomic_fetch_add(sp.tick + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ks - 1) ? 1 : 0;
        __syncthreads();
        const int last = *flag;
        __syncthreads();                 // smem is reused by the epilogue
        if (!last) return;
        const float* base = sp.part + (long)item * ks * 65536 + tid;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sum = 0.f;
                    for (int q = 0; q < ks; ++q)
                        sum += __hip_atomic_load(base + (long)q * 65536 + ((j * 4 + i) * 16 + r) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc[j][i][r] = sum;
                }
    }

    // ---- epilogue: two chunks of 128 tile rows (chunk i = A half i) through LDS; every wave has rows in both
    const int rslot = tid % (BN / 8);
    const int rrow0 = tid / (BN / 8);
    constexpr int RPP = NT / (BN / 8);
    const int gn0 = n0 + rslot * 8;
    float gv[8];
    if (pgate) {
        const uint4 graw = *reinterpret_cast<const uint4*>(pgate + gn0);
        const uint32_t gw[4] = {graw.x, graw.y, graw.z, graw.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { gv[2 * c] = bf2f((uint16_t)(gw[c] & 0xffff)); gv[2 * c + 1] = bf2f((uint16_t)(gw[c] >> 16)); }
    }
    bf16_t* cbase; long ldc; int cn;
    if (gn0 >= p.n_split) { cbase = pC1; ldc = p.ldc1; cn = gn0 - p.n_split; }
    else { cbase = pC; ldc = p.ldc; cn = gn0; }
    uint2 bq[2][4];
    {
        const bf16_t* bsrc = pbias ? pbias : (const bf16_t*)p.B;   // always readable: no branch around the array writes
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) bq[j][a] = *reinterpret_cast<const uint2*>(bsrc + n0 + j * 128 + wc * 32 + 8 * a + 4 * lh);
    }
    static_assert(128 / RPP == 8, "8 row passes per 128-row chunk");
#define G8_RROW(it_) ((m0 + 128 * chunk + rrow0 + RPP * (it_) > p.M - 1) ? p.M - 1 : m0 + 128 * chunk + rrow0 + RPP * (it_))
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
        uint4 rq0, rq1, rq2, rq3, rq4, rq5, rq6, rq7;
        if (pgate) {
            GM_RES_LOAD1(0, G8_RROW(0)) GM_RES_LOAD1(1, G8_RROW(1)) GM_RES_LOAD1(2, G8_RROW(2)) GM_RES_LOAD1(3, G8_RROW(3))
            GM_RES_LOAD1(4, G8_RROW(4)) GM_RES_LOAD1(5, G8_RROW(5)) GM_RES_LOAD1(6, G8_RROW(6)) GM_RES_LOAD1(7, G8_RROW(7))
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int nl = j * 128 + wc * 32 + 8 * a + 4 * lh;
                const int gn = n0 + nl;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (pbias) {
                    const uint2 braw = bq[j][a];
                    bv[0] = bf2f((uint16_t)(braw.x & 0xffff)); bv[1] = bf2f((uint16_t)(braw.x >> 16));
                    bv[2] = bf2f((uint16_t)(braw.y & 0xffff)); bv[3] = bf2f((uint16_t)(braw.y >> 16));
                }
                const bool do_gelu = gn >= p.gelu_from;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    float y[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = rbf(acc[j][2 * chunk + f][4 * a + c] * p.alpha + bv[c]);
                        if (do_gelu) v = gelu_tanh(v);
                        y[c] = v;
                    }
                    uint2 o; o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]);
                    const int ml = wr * 64 + f * 32 + l31;
                    *reinterpret_cast<uint2*>(smem + ml * CROW + nl * 2) = o;
                }
            }
        }
        __syncthreads();
        GM_ROW_PASS1(0, gm < p.M) GM_ROW_PASS1(1, gm < p.M) GM_ROW_PASS1(2, gm < p.M) GM_ROW_PASS1(3, gm < p.M)
        GM_ROW_PASS1(4, gm < p.M) GM_ROW_PASS1(5, gm < p.M) GM_ROW_PASS1(6, gm < p.M) GM_ROW_PASS1(7, gm < p.M)
    }
}

// ---------------------------------------------------------------------------------------------
// Small-M path (M <= 8): y[m, n] = act_out( sum_k act_in(x[m,k]) * W[n,k] + b[n] ), one wave per n.
// Used for the timestep / guidance / pooled-text embedders and all AdaLN modulation linears
// (M = batch = 1): pure weight streaming, HBM-bound.

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemvParams p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const bf16_t* wrow = (const bf16_t*)p.W + (long)n * p.ldw;
    const bf16_t* pbias = (const bf16_t*)p.bias;
    for (int m = 0; m < p.M; ++m) {
        const bf16_t* xr = (const bf16_t*)p.x + (long)m * p.ldx;
        float s = 0.f;
        for (int k = lane * 8; k < p.K; k += 64 * 8) {
            const uint4 wv = *reinterpret_cast<const uint4*>(wrow + k);
            const uint4 xv = *reinterpret_cast<const uint4*>(xr + k);
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x0 = bf2f((uint16_t)(xw[c] & 0xffff)), x1 = bf2f((uint16_t)(xw[c] >> 16));
                if (p.silu_in) { x0 = rbf(silu_f(x0)); x1 = rbf(silu_f(x1)); }
                s += x0 * bf2f((uint16_t)(ww[c] & 0xffff));
                s += x1 * bf2f((uint16_t)(ww[c] >> 16));
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            float v = s + (pbias ? bf2f(pbias[n]) : 0.f);
            v = rbf(v);
            if (p.silu_out) v = silu_f(v);
            ((bf16_t*)p.y)[(long)m * p.ldy + n] = f2bf(v);
        }
    }
}

template <int BM, int BN, int NWM, int NWN, bool CONV = false, bool MX8 = false>
static int launch_gemm(GemmParams p, hipStream_t stream, int group_env, int dbg_env) {
    constexpr int LDS = 2 * (BM + BN) * GM_BK * 2;
    UTX_ONCE_PER_DEVICE(attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN, NWM, NWN, CONV, MX8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    const int ntm = (p.M + BM - 1) / BM;
    const int ntn = (p.N + BN - 1) / BN;
    int group_m = group_env > 0 ? group_env : GM_GROUP_M;
    if (group_m > ntm) group_m = ntm;
    p.ntn = ntn | (group_m << 16) | (dbg_env << 24);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, NWM, NWN, CONV, MX8>), dim3(ntm * ntn), dim3(64 * NWM * NWN), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

static int launch_gemm8(GemmParams p, hipStream_t stream, int group_env, int dbg_env) {
    constexpr int LDS = 131072;
    UTX_ONCE_PER_DEVICE(attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_8ph_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    const int ntm = (p.M + 255) / 256, ntn = p.N / 256;
    int group_m = group_env > 0 ? group_env : GM_GROUP_M;
    if (group_m > ntm) group_m = ntm;
    p.ntn = ntn | (group_m << 16) | (dbg_env << 24);
    G8Split sp = {1, 0, nullptr, nullptr};
    const int tiles = ntm * ntn;
    // Tail split, opt-in (UTX_GEMM_TAILSPLIT=1): see G8Split.  Candidate when the last round is sparsely filled and K is long
    // enough that a tile's fixed cost (prologue + epilogue, ~10 us against ~1.4 us per K-tile) does not dominate.
    const int ncu = utx_ncu();
    const bool enabled = g_utx_opt.gemm_tailsplit == 1 && dbg_env == 0;
    const int nfull = (tiles / ncu) * ncu, r = tiles - nfull;
    int best_ks = 1;
    if (enabled && nfull > 0 && r > 0) {
        const int nkt = p.K / GM_BK;
        const double f = 10.0 / (1.4 * nkt + 10.0);
        double best = 0.88;
        for (int ks = 2; ks <= 8; ++ks) {
            if (ks * 3 > nkt || r * ks > 512) break;
            const double cost = (double)((r * ks + ncu - 1) / ncu) * ((1.0 - f) / ks + f + 0.02 * ks);
            if (cost < best) { best = cost; best_ks = ks; }
        }
    }
    if (best_ks == 1) {
        hipLaunchKernelGGL(gemm256_8ph_kernel, dim3(tiles), dim3(512), LDS, stream, p, sp);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    // Workspace: one (partials, tickets) pair PER STREAM -- FluxDiT launches GEMMs on two streams at once and a shared pair would race
    // between them -- grown outside of any capture: a launch that would have to allocate while its stream is being captured into a
    // graph falls back to the unsplit launch (same bits: the split changes only who sums the K ranges).
    struct TailWs { int dev; hipStream_t stream; float* part; int* tick; size_t cap; };
    static TailWs ws_tab[32] = {};
    static int ws_n = 0;
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return -5;
    const size_t need = (size_t)r * best_ks;
    TailWs* ws = nullptr;
    for (int i = 0; i < ws_n; ++i) if (ws_tab[i].dev == dev && ws_tab[i].stream == stream) ws = &ws_tab[i];
    if (!ws || ws->cap < need) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        if (capturing || (!ws && ws_n == 32)) {
            hipLaunchKernelGGL(gemm256_8ph_kernel, dim3(tiles), dim3(512), LDS, stream, p, sp);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
        if (!ws) { ws = &ws_tab[ws_n++]; *ws = TailWs{dev, stream, nullptr, nullptr, 0}; }
        if (ws->part) (void)hipFree(ws->part);        // hipFree waits for the launches that still read it
        if (ws->tick) (void)hipFree(ws->tick);
        ws->part = nullptr; ws->tick = nullptr; ws->cap = 0;
        if (hipMalloc((void**)&ws->part, need * 65536 * sizeof(float)) != hipSuccess) return -5;
        if (hipMalloc((void**)&ws->tick, 1024 * sizeof(int)) != hipSuccess) return -5;
        ws->cap = need;
    }
    if (hipMemsetAsync(ws->tick, 0, (size_t)r * sizeof(int), stream) != hipSuccess) return -5;
    hipLaunchKernelGGL(gemm256_8ph_kernel, dim3(nfull), dim3(512), LDS, stream, p, sp);
    G8Split st = {best_ks, nfull, ws->part, ws->tick};
    hipLaunchKernelGGL(gemm256_8ph_kernel, dim3(r * best_ks), dim3(512), LDS, stream, p, st);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// Which kernel a descriptor goes to, and how the one-wave-per-SIMD kernel would cut its last round -- pure host arithmetic (no HIP call), shared by
// the launcher below and by utx_gemm_plan (C ABI: the host side and the CPU tests read the decision instead of restating it).
// kernel: 0 = 128^2 (incl. the implicit convolution / MX fp8 forms), 1 = one wave per SIMD (gemm_w4.hip), 2 = persistent 8-wave, 3 = per-tile 8-phase,
// 4 = 2-barrier 256^2.  Shape validation stays in the launcher.
extern "C" void utx_gemm_plan_impl(const GemmParams* pp, int ncu, int sk_has_work, int out[4]) {
    const GemmParams& p = *pp;
    const int dbg_env = g_utx_opt.gemm_debug_abl, tile_env = g_utx_opt.gemm_tile;
    out[0] = 0; out[1] = ((p.M + 127) / 128) * ((p.N + 127) / 128); out[2] = 0; out[3] = 0;
    if (p.conv_Wo > 0 || p.mx8 == 1) return;
    if (p.mx8 == 2) {     // MX fp8 with tile-packed scales: the one-wave-per-SIMD kernel only (gemm_w4.hip, MX); the launcher refuses other shapes
        out[0] = 1; out[1] = ((p.M + 255) / 256) * (p.N / 256);
        int grid = out[1] < ncu ? out[1] : ncu;
        if (g_utx_opt.gemm_pers_grid > 0 && g_utx_opt.gemm_pers_grid < grid) grid = g_utx_opt.gemm_pers_grid;
        utx_gemm_w4_split_plan(&p, out[1], grid, sk_has_work, &out[2], &out[3]);
        return;
    }
    // 256^2 tiles need every column boundary on a 256 multiple and enough tiles to fill the chip
    const bool ok256 = (p.N % 256 == 0) && (p.n_split >= p.N || p.n_split % 256 == 0) &&
                       (p.gelu_from >= p.N || p.gelu_from % 256 == 0) &&
                       (p.K2 == 0 || (p.lora_seg_n % 256 == 0 && p.lora_n_limit % 256 == 0));
    const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    bool use256 = ok256 && tiles256 >= 192;
    if (tile_env == 128) use256 = false;
    if ((tile_env == 256 || tile_env == 2562 || tile_env == 2560 || tile_env == 2564) && ok256) use256 = true;   // 2562 = the 2-barrier 256^2 kernel, 2560 = force the persistent kernel (A/B testing)
    if (!use256) return;
    out[1] = (int)tiles256;
    // default for the large-M linears: the persistent one-wave-per-SIMD kernel (gemm_w4.hip; +5...10 % over the persistent 8-wave kernel on the FLUX
    // shapes, profiles/r02_gemm_w4_check_v7.log); UTX_GEMM_TILE=2560 keeps the persistent 8-wave kernel (gemm_pers.hip), 256 the per-tile-launch
    // 8-phase kernel, for A/B (all bit-identical); the timing ablations / the 8-phase tail split only exist in the latter.
    const bool plain_opts = (dbg_env & 3) == 0 && g_utx_opt.gemm_tailsplit == 0;
    if ((tile_env == 0 || tile_env == 2564) && plain_opts) {
        out[0] = 1;
        int grid = tiles256 < ncu ? (int)tiles256 : ncu;
        if (g_utx_opt.gemm_pers_grid > 0 && g_utx_opt.gemm_pers_grid < grid) grid = g_utx_opt.gemm_pers_grid;
        utx_gemm_w4_split_plan(&p, (int)tiles256, grid, sk_has_work, &out[2], &out[3]);
    } else if (tile_env == 2560 && plain_opts) out[0] = 2;
    else if (tile_env != 2562) out[0] = 3;
    else out[0] = 4;
}

extern "C" int utx_launch_gemm_bf16(const GemmParams* hp, hipStream_t stream) {
    GemmParams p = *hp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
    if (p.mx8) {
        // MX fp8 base segment: K counts fp8 elements (multiple of 128 = one K-tile), lda / ldb are bytes; scales [rows][K/32] bytes
        if (!p.a_scale || !p.b_scale || (p.K % 128) || (p.lda & 15) || (p.ldb & 15) || p.conv_Wo > 0 || p.mx8 > 2) return -2;
        if (p.mx8 == 1 && ((p.lds_a & 3) || (p.lds_b & 3) || p.lds_a < p.K / 32 || p.lds_b < p.K / 32)) return -2;
        p.K /= 2; p.lda /= 2; p.ldb /= 2;          // bf16 units: the staging code is byte-identical
    }
    if ((p.K % GM_BK) || (p.K2 % GM_BK) || (p.N % 8)) return -2;
    if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 7)) return -2;
    if (p.K2 > 0 && (!p.A2 || !p.B2 || (p.lda2 & 7) || (p.ldb2 & 7) || p.lora_seg_n <= 0 || (p.lora_seg_n % 128))) return -2;
    if (p.gate && (!p.res || (p.ldres & 7))) return -2;
    if (p.n_split < p.N && (!p.C1 || (p.n_split % 128) || (p.ldc1 & 7))) return -2;
    const int group_env = g_utx_opt.gemm_group_m, dbg_env = g_utx_opt.gemm_debug_abl, tile_env = g_utx_opt.gemm_tile;
    // fused q / k post-processing exists only in the bf16 one-wave-per-SIMD kernel: refuse every other form here, in front of the early returns of
    // the convolution / MX forms, instead of silently dropping it
    if (p.qk_cols > 0 && (p.mx8 || p.conv_Wo > 0 || p.gate)) return -2;
    if (p.q_out && p.mx8 != 2) return -2;      // fp8 output of the GELU tiles exists in the MX one-wave-per-SIMD kernel only: refused elsewhere, never dropped
    if (p.conv_Wo > 0) {   // implicit 3x3 convolution: 128^2 kernel, A rows gathered per tap
        if (p.K2 > 0 || !p.zero_page || p.conv_cin_log2 < 6 || p.K != (9 << p.conv_cin_log2) || p.conv_Hi <= 0 || p.conv_Wi <= 0 ||
            p.conv_stride < 1 || p.conv_stride > 2 || p.conv_pad < 0 || p.conv_pad > 1 || (p.conv_up & ~1) || (p.M % p.conv_Wo))
            return -2;
        return launch_gemm<128, 128, 2, 2, true>(p, stream, 0, 0);
    }
    if (p.mx8 == 2) {
        // tile-packed scales: 256 x 256 tiles only (every column boundary on a 256 multiple), no LoRA segment (merge the adapters into the weights
        // before quantising: unitex_amd/flux/transformer.py), no fused q / k epilogue
        if ((p.N % 256) || (p.n_split < p.N && p.n_split % 256) || (p.gelu_from < p.N && p.gelu_from % 256) || p.K2 > 0 || p.qk_cols > 0) return -2;
        return utx_launch_gemm_w4(p, stream);
    }
    if (p.mx8) return launch_gemm<128, 128, 2, 2, false, true>(p, stream, group_env, 0);
    int plan[4];
    utx_gemm_plan_impl(&p, 256, 1, plan);      // the kernel choice does not depend on the CU count (only the split of the last round does)
    if (p.qk_cols > 0) {
        // fused q / k post-processing: only the one-wave-per-SIMD kernel has it (a wave owns a whole head there); refuse instead of dropping it
        if ((p.qk_cols % 256) || p.qk_cols > p.N || p.qk_cols > p.n_split || p.qk_cols > p.gelu_from ||
            !p.qk_wq || !p.qk_wk || !p.qk_cos || !p.qk_sin || !p.qk_Qh || !p.qk_Kh || p.qk_hs <= 0 || p.qk_tok_off < 0) return -2;
        if (plan[0] != 1) return -2;
    }
    if (plan[0] == 1) return utx_launch_gemm_w4(p, stream);
    if (plan[0] == 2) return utx_launch_gemm_pers(p, stream);
    if (plan[0] == 3) return launch_gemm8(p, stream, group_env, dbg_env);
    if (plan[0] == 4) return launch_gemm<256, 256, 2, 4>(p, stream, group_env, dbg_env);
    return launch_gemm<128, 128, 2, 2>(p, stream, group_env, dbg_env);
}

extern "C" int utx_launch_gemv_bf16(const GemvParams* hp, hipStream_t stream) {
    GemvParams p = *hp;
    if (p.M <= 0 || p.M > 8 || p.N <= 0 || p.K <= 0 || (p.K % 8) || (p.ldw & 7) || (p.ldx & 7)) return -2;
    hipLaunchKernelGGL(gemv_bf16_kernel, dim3((p.N + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
