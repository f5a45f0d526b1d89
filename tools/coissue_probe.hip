// Does the MFMA stream of one wave overlap the VALU / transcendental stream of ANOTHER wave on the same SIMD?
// (design question behind the attention kernel: QK^T / PV of one wave group vs softmax of the other.)
//   hipcc --offload-arch=gfx950 -O2 tools/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
// 512-thread workgroups, one per CU: waves 0-3 = role A (MFMA chain), waves 4-7 = role B (VALU), so every SIMD
// holds one A and one B wave.  mode bit0 = A active, bit1 = B active; B's mix: 32 v_exp_f32 + NADD plain VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
static int g_grid = 256;   // workgroups launched (one per CU); argv[1] -- fewer active CUs = more power/clock headroom per CU
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NEXP, int NADD, int PRIO>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool roleA = wave < 4;
    if (roleA) {
        if (!(mode & 1)) return;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        float s = 0;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        float x[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = 0.01f * (threadIdx.x + j);
        float acc = 0.f;
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < NEXP; ++j) x[j & 31] = __builtin_amdgcn_exp2f(x[j & 31]) - 1.0f;   // 1 exp + 1 add
#pragma unroll
            for (int j = 0; j < NADD; ++j) x[j & 31] = x[j & 31] * 0.999f + 0.001f;               // plain VALU (fma)
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += x[j];
        out[blockIdx.x * 512 + threadIdx.x] = acc;
    }
}

// same-wave interleave: every wave issues 16 MFMA and 2 x 16 (exp + add) per iteration, VALU slotted between MFMAs.
// what: 1 = MFMA only, 2 = VALU only, 3 = both interleaved;  KIND 0: v_exp + v_add per element, 1: two v_fma, 2: one v_fma
#define VOP(v_) (KIND == 0 ? __builtin_amdgcn_exp2f(v_) - 1.0f : KIND == 1 ? ((v_) * 0.999f + 0.001f) * 1.001f - 0.001f : (v_) * 0.999f + 0.001f)
template <int PER, int what, int KIND>
__global__ __launch_bounds__(512) void probe_same(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = 0.01f * (threadIdx.x + j);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (what & 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + j) & 31] = VOP(x[(8 * k + j) & 31]);
            }
            if (what & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 2 + j) & 31] = VOP(x[(8 * k + 2 + j) & 31]);
            }
            if (what & 1) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 4 + j) & 31] = VOP(x[(8 * k + 4 + j) & 31]);
            }
            if (what & 1) c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 6 + j) & 31] = VOP(x[(8 * k + 6 + j) & 31]);
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
#pragma unroll
    for (int j = 0; j < 32; ++j) s += x[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int PER, int KIND>
static void run_same(const char* name, float* d, int threads) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[4] = {0, 0, 0, 0};
#define RUN_SAME(W_)                                                                                  \
    hipLaunchKernelGGL((probe_same<PER, W_, KIND>), dim3(g_grid), dim3(threads), 0, 0, d, 1000);               \
    hipDeviceSynchronize();                                                                           \
    hipEventRecord(a);                                                                                \
    hipLaunchKernelGGL((probe_same<PER, W_, KIND>), dim3(g_grid), dim3(threads), 0, 0, d, iters);              \
    hipEventRecord(b); hipEventSynchronize(b);                                                        \
    hipEventElapsedTime(&ms[W_], a, b);
    RUN_SAME(1) RUN_SAME(2) RUN_SAME(3)
    printf("%-28s  mfma %.3f ms   valu %.3f ms   interleaved %.3f ms   overlap = %.2f\n",
           name, ms[1], ms[2], ms[3], (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]));
}


// same as probe_same<PER, what, KIND> but the four accumulators are pinned in AGPRs (MFMA C/D in the AGPR file, the VALU
// stream only touches arch VGPRs): separates "issue port" contention from "VGPR bank/port" contention.
template <int PER, int what, int KIND>
__global__ __launch_bounds__(256) void probe_same_agpr(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = 0.01f * (threadIdx.x + j);
#define MF(c_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(a), "v"(b))
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (what & 1) MF(c0);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + j) & 31] = VOP(x[(8 * k + j) & 31]);
            }
            if (what & 1) MF(c1);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 2 + j) & 31] = VOP(x[(8 * k + 2 + j) & 31]);
            }
            if (what & 1) MF(c2);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 4 + j) & 31] = VOP(x[(8 * k + 4 + j) & 31]);
            }
            if (what & 1) MF(c3);
            if (what & 2) {
#pragma unroll
                for (int j = 0; j < PER; ++j) x[(8 * k + 6 + j) & 31] = VOP(x[(8 * k + 6 + j) & 31]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
#pragma unroll
    for (int j = 0; j < 32; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PER, int KIND>
static void run_same_agpr(const char* name, float* d) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[4] = {0, 0, 0, 0};
#define RUN_SAMEA(W_)                                                                                 \
    hipLaunchKernelGGL((probe_same_agpr<PER, W_, KIND>), dim3(g_grid), dim3(256), 0, 0, d, 1000);     \
    hipDeviceSynchronize();                                                                           \
    hipEventRecord(a);                                                                                \
    hipLaunchKernelGGL((probe_same_agpr<PER, W_, KIND>), dim3(g_grid), dim3(256), 0, 0, d, iters);    \
    hipEventRecord(b); hipEventSynchronize(b);                                                        \
    hipEventElapsedTime(&ms[W_], a, b);
    RUN_SAMEA(1) RUN_SAMEA(2) RUN_SAMEA(3)
    printf("%-40s  mfma %.3f ms   valu %.3f ms   interleaved %.3f ms   overlap = %.2f\n",
           name, ms[1], ms[2], ms[3], (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]));
}

template <int NEXP, int NADD, int PRIO>
static void run(const char* name, float* d) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((probe<NEXP, NADD, PRIO>), dim3(g_grid), dim3(512), 0, 0, d, 1000, mode);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<NEXP, NADD, PRIO>), dim3(g_grid), dim3(512), 0, 0, d, iters, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[mode], a, b);
    }
    // per iteration: A = 16 MFMA (ideal 512 cycles), B = NEXP exp+add, NADD fma
    printf("%-28s  A(mfma) %.3f ms   B(valu) %.3f ms   both %.3f ms   overlap = %.2f (1 = perfect, 0 = serial)  | ns/iter A %.1f B %.1f\n",
           name, ms[1], ms[2], ms[3], (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]),
           ms[1] * 1e6 / iters, ms[2] * 1e6 / iters);
}

int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    printf("grid = %d workgroups\n", g_grid);
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<32, 0, 0>("32 exp+add, prio0", d);
    run<32, 0, 1>("32 exp+add, prio1", d);
    run<0, 64, 0>("64 fma, prio0", d);
    run<0, 64, 1>("64 fma, prio1", d);
    run<32, 0, 2>("32 exp+add, VALU wave prio3", d);
    run<0, 64, 2>("64 fma, VALU wave prio3", d);
    run<32, 64, 2>("32 exp+add+64 fma, VALU prio3", d);
    run<32, 64, 1>("32 exp+add + 64 fma, prio1", d);
    run<16, 32, 1>("16 exp+add + 32 fma, prio1", d);
    run_same<2, 0>("same wave, 1 w/SIMD, 2 exp+add/MFMA", d, 256);
    run_same<2, 0>("same wave, 2 w/SIMD, 2 exp+add/MFMA", d, 512);
    run_same<1, 0>("same wave, 2 w/SIMD, 1 exp+add/MFMA", d, 512);
    run_same<2, 1>("same wave, 1 w/SIMD, 4 fma/MFMA", d, 256);
    run_same<2, 1>("same wave, 2 w/SIMD, 4 fma/MFMA", d, 512);
    run_same<2, 2>("same wave, 1 w/SIMD, 2 fma/MFMA", d, 256);
    run_same<1, 2>("same wave, 1 w/SIMD, 1 fma/MFMA", d, 256);
    run_same_agpr<2, 0>("AGPR acc, 1 w/SIMD, 2 exp+add/MFMA", d);
    run_same_agpr<2, 1>("AGPR acc, 1 w/SIMD, 4 fma/MFMA", d);
    run_same_agpr<2, 2>("AGPR acc, 1 w/SIMD, 2 fma/MFMA", d);
    run_same_agpr<1, 2>("AGPR acc, 1 w/SIMD, 1 fma/MFMA", d);
    return 0;
}
