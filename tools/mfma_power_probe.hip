// Sustained MFMA rate of the whole chip on RANDOM operands (power / clock limited), by instruction shape: the ceiling any GEMM or
// attention kernel can approach on this board.  One or two waves per SIMD, register-resident operand fragments rotated GEMM-style
// (8 A x 4 B fragments per pass), no memory traffic in the loop.   out: TF/s per variant.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int MODE>   // 0: 32x32x16 bf16, 1: 16x16x32 bf16, 2: 32x32x64 MX fp8
__global__ __launch_bounds__(512, 2) void burn(float* out, int iters, int zero) {
    uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    bf16x8 a[8], b[4];
    i32x8 a8[4], b8[2];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = zero ? (__bf16)0.f : (__bf16)(((int)(rnd(s) >> 16) - 32768) / 32768.0f);
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = zero ? (__bf16)0.f : (__bf16)(((int)(rnd(s) >> 16) - 32768) / 32768.0f);
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a8[i][e] = zero ? 0 : (int)(rnd(s) & 0x7f7f7f7f) ;   // positive / small exponents avoided? random bytes
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b8[i][e] = zero ? 0 : (int)(rnd(s) & 0x3f3f3f3f);
    float r = 0.f;
    if (MODE == 0) {
        f32x16 c[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[(i & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], c[(i & 1) * 4 + j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) r += c[i][e];
    } else if (MODE == 1) {
        f32x4 c[16];
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[(i & 3) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[(i & 3) * 4 + j], 0, 0, 0);
                    c[(i & 3) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[(i & 3) * 4 + j], 0, 0, 0);
                }
        }
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) r += c[i][e];
    } else {
        f32x16 c[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], c[i * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) r += c[i][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

extern "C" double mfma_burn(int mode, int threads, int iters, int zero, int nwg) {
    float* d; hipMalloc(&d, (size_t)nwg * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(burn<0>, dim3(nwg), dim3(threads), 0, 0, d, iters, zero);
        else if (mode == 1) hipLaunchKernelGGL(burn<1>, dim3(nwg), dim3(threads), 0, 0, d, iters, zero);
        else hipLaunchKernelGGL(burn<2>, dim3(nwg), dim3(threads), 0, 0, d, iters, zero);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    const double waves = (double)nwg * threads / 64;
    const double flops_per_iter = mode == 0 ? 32.0 * 32768 : mode == 1 ? 64.0 * 16384 : 8.0 * 131072;
    return waves * iters * flops_per_iter / (ms * 1e-3) / 1e12;
}
