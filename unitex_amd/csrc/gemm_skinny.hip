// bf16 GEMM for SKINNY outputs over many rows:  C[M, N] = alpha * A[M, K] . B[N, K]^T (+ bias[N]),  N = 32 NI <= 192,  M large -- the LoRA-down products
// x . A_lora^T of the FLUX linears (N = 64 per adapter and fused projection: 64 / 192; peft computes lora_A(x) as a linear of its own, [3p]; the call site is
// unitex_amd/flux/transformer.py `_gemm`, the reference's is diffusers' LoRA-wrapped nn.Linear inside flux_piplines/texturing/pipeline.py:646-656).
//
// Why a kernel of its own (profiles/r06_small_gemm_census.log): these launches are one pass over the activations -- 309 MB at K = 3072, 1.2 GB at K = 12 288 -- and nothing
// else; the 128 x 128 tile kernel of gemm.hip ran them at 95.6 / 143 / 272 us (N = 64 / 192 at K = 3072, N = 64 at K = 12 288) against 49 / 49 / 196 us for that pass at
// 6.3 TB/s: N = 192 is two column tiles, i.e. A is read twice, and its two-stage ring waits for every load of a K-step at one barrier (the HBM queue runs half empty).
// Here a workgroup owns 128 rows x ALL N columns (A is read once; N <= 192), a wave owns 32 of those rows, and
//   * A goes STRAIGHT INTO REGISTERS in the MFMA operand layout (lane (m, h): 8 bf16 at k = 16 kk + 8 h of row m -- no wave shares a row, so LDS would add nothing),
//     requested THREE K-steps ahead (four register sets rotating by name): 48 KB per workgroup in flight, waited for by hipcc's own counted vmcnt;
//   * B (the N x K adapter, L2-resident: <= 1.5 MB) goes through registers into a two-slot LDS ring, one 64-k slab of N rows per K-step, requested a step before it is
//     written, XOR-swizzled like gemm.hip.
// Arithmetic = gemm.hip's: the same swapped MFMA (A-operand = B rows), the same K order, the same epilogue expression -- bit-identical outputs
// (tests/test_dit_ops_gpu.py::test_skinny_gemm_matches_the_tile_kernel_bitwise).
#include "common.h"
#include "kernels.h"

#define GS_BK 64
typedef int gs_i32x4 __attribute__((ext_vector_type(4)));      // a native vector (HIP's uint4 is a struct: arrays of it stayed in scratch)
// the loop's barrier WITHOUT __syncthreads()'s fence: hipcc puts s_waitcnt vmcnt(0) in front of that one, which would drain the loads requested K-steps ahead at every
// K-step.  What the barrier has to order is stated by hand: this wave's LDS writes of the next slab and its reads of the current one are complete (s_waitcnt lgkmcnt(0) in front
// of it); the compiler may move nothing across it.  Every vector-memory operation of the loop is a plain register load, so hipcc's own COUNTED vmcnt waits are exact (a form with
// LDS-DMA for B had to write them by hand and hipcc's -- blind to the DMAs -- then over-waited: profiles/r06_gemm_skinny_ab.log, first table)
#define GS_BAR()                                      \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                 \
        asm volatile("" ::: "memory");                \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)

template <int NI>
__global__ __launch_bounds__(256, 2) void gemm_skinny_n_kernel(GemmParams p) {
    constexpr int N_ROWS = 32 * NI;                 // rows of B staged per K-step (= the tile's columns)
    constexpr int STAGE = N_ROWS * 128;             // one 64-k slab of B: N_ROWS rows of 128 bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* pA = (const bf16_t*)p.A; const bf16_t* pB = (const bf16_t*)p.B; const bf16_t* pbias = (const bf16_t*)p.bias;
    bf16_t* pC = (bf16_t*)p.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int nk = p.K / GS_BK;

    // ---- A: this lane's row (clamped: rows past M are computed and never stored), 16 bytes at k = 64 kt + 16 kk + 8 lh
    int arow = m0 + l31; if (arow > p.M - 1) arow = p.M - 1;
    const bf16_t* pa = pA + (long)arow * p.lda + 8 * lh;
    // ---- B: piece j of this thread = slab row 8 (wave + 4 j) + lane / 8, global chunk lane % 8 (a row's 128 bytes by eight neighbouring lanes); it lands in the LDS slot
    // chunk ^ ((row >> 1) & 7) of its row (gemm.hip's swizzle: the fragment reads below are conflict-free)
    const int srow_in = lane >> 3, schunk = lane & 7;
    const bf16_t* pb[NI];
    int bdst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = 8 * (wave + 4 * j) + srow_in;
        pb[j] = pB + (long)r * p.ldb + (schunk << 3);       // r < N: the slab has exactly N rows
        bdst[j] = r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4);
    }
#define GS_LOAD_B(dst_, kt_)                                                                                         \
    do {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) dst_[j] = *reinterpret_cast<const gs_i32x4*>(pb[j] + (long)(kt_) * GS_BK); \
    } while (0)
#define GS_WRITE_B(src_, buf_)                                                                                       \
    do {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) *reinterpret_cast<gs_i32x4*>(smem + (buf_) * STAGE + bdst[j]) = src_[j]; \
    } while (0)
#define GS_LOAD_A(dst_, kt_)                                                                                         \
    do {                                                                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) dst_[kk] = *reinterpret_cast<const bf16x8*>(pa + (long)(kt_) * GS_BK + 16 * kk); \
    } while (0)

    f32x16 acc[NI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // fragment read of B: row l31 of column block i, chunk 2 kk + lh -> i * 4096 + l31 * 128 + ((chunk ^ swz) << 4); 32 i leaves (row >> 1) & 7 unchanged
    const int boff0 = l31 * 128, bswz0 = (l31 >> 1) & 7;

    // ---- the pipeline.  Requests: A(kt + 3) and B(kt + 2) go out at the head of step kt (indices clamp at nk - 1: the tail re-requests the last K-step, results unused);
    // B(kt + 1) -- requested at the head of the previous step -- is written to LDS slot (kt + 1) & 1 behind the step's MFMAs (that slot was last read in step kt - 1, which every wave
    // left through the barrier), then the barrier.  A's four register sets and B's two rotate BY NAME over a four-step loop body (no register copies: a first form that reused one B set made
    // hipcc rename it and copy the fresh loads home, i.e. wait for them in the step that requested them).
    bf16x8 a0[4], a1[4], a2[4], a3[4];
    gs_i32x4 b0[NI], b1[NI];
    const int last = nk - 1;
    GS_LOAD_B(b0, 0);
    GS_LOAD_A(a0, 0);
    GS_LOAD_B(b1, (1 < last ? 1 : last));
    GS_LOAD_A(a1, (1 < last ? 1 : last));
    GS_LOAD_A(a2, (2 < last ? 2 : last));
    GS_WRITE_B(b0, 0);
    GS_BAR();
#define GS_STEP(cur_, nxt3_, bw_, bl_, kt_)      /* bw_: holds B(kt + 1), written to LDS behind the MFMAs; bl_: the other set, free since the previous step's write, receives B(kt + 2) at the head */ \
    do {                                                                                                             \
        const int ktb_ = ((kt_) + 2 < last) ? (kt_) + 2 : last, kta_ = ((kt_) + 3 < last) ? (kt_) + 3 : last;        \
        GS_LOAD_B(bl_, ktb_);                                                                                        \
        GS_LOAD_A(nxt3_, kta_);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);      /* the requests go out in FRONT of the step's MFMAs (hipcc sank them behind) */ \
        const char* st_ = smem + ((kt_) & 1) * STAGE;                                                                \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                           \
            const char* pbk_ = st_ + boff0 + (((2 * kk + lh) ^ bswz0) << 4);                                         \
            _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                           \
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pbk_ + i * 4096), cur_[kk], acc[i], 0, 0, 0); \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        GS_WRITE_B(bw_, ((kt_) + 1) & 1);                                                                            \
        GS_BAR();                                                                                                    \
    } while (0)
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
        GS_STEP(a0, a3, b1, b0, kt);
        GS_STEP(a1, a0, b0, b1, kt + 1);
        GS_STEP(a2, a1, b1, b0, kt + 2);
        GS_STEP(a3, a2, b0, b1, kt + 3);
    }
    if (kt < nk) { GS_STEP(a0, a3, b1, b0, kt); ++kt; }
    if (kt < nk) { GS_STEP(a1, a0, b0, b1, kt); ++kt; }
    if (kt < nk) { GS_STEP(a2, a1, b1, b0, kt); ++kt; }

    // ---- epilogue: gemm.hip's expression y = bf16(alpha * acc + bias); lane (m = l31, h) holds acc[ni][4 a + c] = C[m][32 ni + 8 a + 4 h + c]: 8-byte stores
    uint2 bq[NI][4];
    {
        const bf16_t* bsrc = pbias ? pbias : pB;      // always a readable address: no branch around the loads
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int a = 0; a < 4; ++a) bq[ni][a] = *reinterpret_cast<const uint2*>(bsrc + 32 * ni + 8 * a + 4 * lh);
    }
    const int gm = m0 + l31;
    if (gm < p.M) {
        bf16_t* crow = pC + (long)gm * p.ldc;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n = 32 * ni + 8 * a + 4 * lh;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (pbias) {
                    const uint2 braw = bq[ni][a];
                    bv[0] = bf2f((uint16_t)(braw.x & 0xffff)); bv[1] = bf2f((uint16_t)(braw.x >> 16));
                    bv[2] = bf2f((uint16_t)(braw.y & 0xffff)); bv[3] = bf2f((uint16_t)(braw.y >> 16));
                }
                float y[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) y[c] = rbf(acc[ni][4 * a + c] * p.alpha + bv[c]);
                uint2 o; o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]);
                *reinterpret_cast<uint2*>(crow + n) = o;
            }
    }
}

template <int NI>
static int launch_skinny(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 32 * NI * 128;
    UTX_ONCE_PER_DEVICE(attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_n_kernel<NI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        UTX_ONCE_DONE(attr_set);
    }
    hipLaunchKernelGGL((gemm_skinny_n_kernel<NI>), dim3((p.M + 127) / 128), dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// the launches this kernel takes: a plain bf16 product (optional bias) with N = 32 NI <= 256 columns over >= 4096 rows and >= 4 K-steps -- no LoRA segment, gate, GELU,
// column split, fused q / k epilogue, convolution gather or MX operands (those stay with gemm.hip's kernels)
extern "C" int utx_gemm_skinny_shape(const GemmParams* p) {      // the shape part (what utx_gemm_plan reports: no pointer is looked at)
    return p->M >= 4096 && p->N >= 64 && p->N <= 192 && (p->N % 32) == 0 && (p->K % GS_BK) == 0 && p->K >= 4 * GS_BK && p->K2 == 0 && !p->gate && p->gelu_from >= p->N &&
           p->n_split >= p->N && p->qk_cols == 0 && p->conv_Wo == 0 && !p->mx8 && !p->q_out && (p->lda & 7) == 0 && (p->ldb & 7) == 0 && (p->ldc & 3) == 0;
}
extern "C" int utx_gemm_skinny_takes(const GemmParams* p) {
    return utx_gemm_skinny_shape(p) && ((((uintptr_t)p->A) | ((uintptr_t)p->B)) & 15) == 0 && (((uintptr_t)p->C) & 7) == 0 && (!p->bias || (((uintptr_t)p->bias) & 7) == 0);
}

extern "C" int utx_launch_gemm_skinny(const GemmParams* p, hipStream_t stream) {
    if (!utx_gemm_skinny_takes(p)) return -2;
    switch (p->N / 32) {
        case 2: return launch_skinny<2>(*p, stream);
        case 3: return launch_skinny<3>(*p, stream);
        case 4: return launch_skinny<4>(*p, stream);
        case 5: return launch_skinny<5>(*p, stream);
        case 6: return launch_skinny<6>(*p, stream);
        default: return -2;
    }
}
