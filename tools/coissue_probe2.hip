// How many single-issue instructions does one wave hide behind its own v_mfma_f32_32x32x16_bf16 (32 cycles of matrix pipe)?
// Everything is asm volatile, so the instruction order below IS the issue order (the first probe's same-wave rows were
// re-ordered and SLP-packed by hipcc).  One wave per SIMD (256 threads, one workgroup per CU), 4 independent accumulators.
//   FORM 0: accumulators in AGPRs, 1: in VGPRs.   KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_bf16_f32, 3: v_add_f32 chain
//   hipcc --offload-arch=gfx950 -O2 tools/coissue_probe2.hip -o /tmp/coissue_probe2 && /tmp/coissue_probe2 [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int FORM, int NFILL, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x0 = 0.1f, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, x4 = 0.5f, x5 = 0.6f, x6 = 0.7f, x7 = 0.8f;
    const float k0 = 0.999f, k1 = 0.001f;
#define FILL1(x_)                                                                                          \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x_) : "v"(k0), "v"(k1));                 \
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x_));                                       \
    else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x_) : "v"(k0));                 \
    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x_) : "v"(k1));
#define FILL(base_)                                                                                        \
    { if (NFILL > 0) FILL1(x0) if (NFILL > 1) FILL1(x1) if (NFILL > 2) FILL1(x2) if (NFILL > 3) FILL1(x3)  \
      if (NFILL > 4) FILL1(x4) if (NFILL > 5) FILL1(x5) if (NFILL > 6) FILL1(x6) if (NFILL > 7) FILL1(x7) }
#define MF(c_)                                                                                             \
    if (FORM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(a), "v"(b));    \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c_) : "v"(a), "v"(b));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { MF(c0) FILL(0) MF(c1) FILL(0) MF(c2) FILL(0) MF(c3) FILL(0) }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_grid = 8;
template <int FORM, int NFILL, int KIND>
static float run(float* d) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<FORM, NFILL, KIND>), dim3(g_grid), dim3(256), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<FORM, NFILL, KIND>), dim3(g_grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / (iters * 16.0f);     // ns per MFMA
}
template <int FORM, int KIND>
static void sweep(const char* name, float* d) {
    const float r0 = run<FORM, 0, KIND>(d);
    printf("%-34s ns/MFMA with 0..8 fillers per gap: %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f\n", name, r0,
           run<FORM, 1, KIND>(d), run<FORM, 2, KIND>(d), run<FORM, 3, KIND>(d), run<FORM, 4, KIND>(d), run<FORM, 5, KIND>(d),
           run<FORM, 6, KIND>(d), run<FORM, 7, KIND>(d), run<FORM, 8, KIND>(d));
}
int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    printf("grid = %d workgroups (one wave per SIMD); 32 cycles at 2.4 GHz = 13.3 ns\n", g_grid);
    sweep<0, 0>("AGPR acc, v_fma_f32 fillers", d);
    sweep<1, 0>("VGPR acc, v_fma_f32 fillers", d);
    sweep<0, 1>("AGPR acc, v_exp_f32 fillers", d);
    sweep<1, 1>("VGPR acc, v_exp_f32 fillers", d);
    sweep<0, 2>("AGPR acc, v_cvt_pk_bf16_f32 fillers", d);
    sweep<1, 3>("VGPR acc, v_add_f32 fillers", d);
    return 0;
}
