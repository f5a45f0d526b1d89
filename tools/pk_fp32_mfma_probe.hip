// Probe (gfx950): do PACKED fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of one kernel go wrong while ANOTHER queue's MFMA kernel shares the
// SIMDs?  Round 4 observation: utx_qkv_post, whose rotary arithmetic hipcc compiles to packed fp32 instructions, produced a wrong LOW element in
// lanes 48-63 of a wave about once per 100 forwards while the other stream's 128^2-tile MFMA GEMM ran beside it; built without packed instructions: 0 of 11 000.
// This is the two-kernel experiment outside the library:
//   victim    : every lane runs a chain of packed operations on values whose results are known exactly (small integers / powers of two), recomputes the same
//               chain with scalar v_mul / v_add / v_fma, and counts lanes whose packed and scalar results differ in any bit;
//   aggressor : workgroups of 4 waves looping over v_mfma_f32_32x32x16_bf16 (no memory traffic), few registers, so both kernels are co-resident on every SIMD.
// Arms: victim alone (control), victim beside the aggressor on a second stream, victim beside a VALU-only aggressor.
// build + run:  hipcc --offload-arch=gfx950 -O2 tools/pk_fp32_mfma_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned long long* counts, int iters) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // per-lane operands: small integers (exact in every operation below)
    float a0 = in[(gtid * 4 + 0) & 4095], a1 = in[(gtid * 4 + 1) & 4095], c = in[(gtid * 4 + 2) & 4095], s = in[(gtid * 4 + 3) & 4095];
    unsigned long long bad_lo = 0, bad_hi = 0, bad_top = 0;
    for (int it = 0; it < iters; ++it) {
        // the rotary form: r0 = a0 c - a1 s ; r1 = a1 c + a0 s, packed and scalar
        f2 p = {a0, a1}, q = {-a1, a0}, cc = {c, c}, ss = {s, s}, t0, t1, r;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(p), "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(q), "v"(ss));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(t0), "v"(t1));
        f2 r2;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(p), "v"(cc), "v"(t1));
        float u0, u1, w0, w1, e0, e1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(a0), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(a1), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(-a1), "v"(s));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a0), "v"(s));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(u0), "v"(w0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(u1), "v"(w1));
        const bool lo = (__float_as_uint(r[0]) != __float_as_uint(e0)) || (__float_as_uint(r2[0]) != __float_as_uint(e0));
        const bool hi = (__float_as_uint(r[1]) != __float_as_uint(e1)) || (__float_as_uint(r2[1]) != __float_as_uint(e1));
        bad_lo += lo; bad_hi += hi; bad_top += (lo || hi) && lane >= 48;
        // next operands: keep them small integers, vary them
        a0 = (float)((int)(e0 + (float)it) & 7) - 3.f; a1 = (float)((int)(e1 - (float)it) & 7) - 4.f;
    }
    if (bad_lo) atomicAdd(&counts[0], bad_lo);
    if (bad_hi) atomicAdd(&counts[1], bad_hi);
    if (bad_top) atomicAdd(&counts[2], bad_top);
}

// victim, second form: like utx_qkv_post, the operands of the packed operations arrive by 16-byte global loads in every iteration (a table of small integers)
__global__ __launch_bounds__(256) void victim_loads(const float* __restrict__ in, unsigned long long* counts, int iters) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned long long bad_lo = 0, bad_hi = 0, bad_top = 0;
    unsigned idx = (unsigned)gtid * 4u;
    for (int it = 0; it < iters; ++it) {
        idx = (idx * 1664525u + 1013904223u);
        const float4 v = *reinterpret_cast<const float4*>(in + ((idx >> 8) & 4092u & ~3u));
        const float a0 = v.x, a1 = v.y, c = v.z, s = v.w;
        f2 p = {a0, a1}, q = {-a1, a0}, cc = {c, c}, ss = {s, s}, t0, t1, r;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(p), "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(q), "v"(ss));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(t0), "v"(t1));
        float u0, u1, w0, w1, e0, e1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(a0), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(a1), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(-a1), "v"(s));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a0), "v"(s));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(u0), "v"(w0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(u1), "v"(w1));
        const bool lo = __float_as_uint(r[0]) != __float_as_uint(e0), hi = __float_as_uint(r[1]) != __float_as_uint(e1);
        bad_lo += lo; bad_hi += hi; bad_top += (lo || hi) && lane >= 48;
    }
    if (bad_lo) atomicAdd(&counts[0], bad_lo);
    if (bad_hi) atomicAdd(&counts[1], bad_hi);
    if (bad_top) atomicAdd(&counts[2], bad_top);
}

// aggressor, second form: the shape of a tiled GEMM's K loop -- LDS-DMA of a tile, barrier, fragment reads, MFMAs, barrier
__global__ __launch_bounds__(256) void aggressor_gemm_like(const float* __restrict__ src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char tile[32768];
    const int tid = threadIdx.x;
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((tid * 4 + j * 1024 + it * 64) & 1048572)),
                                             (__attribute__((address_space(3))) void*)(tile + j * 4096 + (tid >> 6) * 1024), 16, 0, 0);
        __syncthreads();
        for (int k = 0; k < 4; ++k) {
            const bf8 a = *reinterpret_cast<const bf8*>(tile + ((tid * 16 + k * 4096) & 32752)), b = *reinterpret_cast<const bf8*>(tile + ((tid * 16 + k * 4096 + 16384) & 32752));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i];
    if (t == 12345.678f) sink[0] = t;
}

__global__ __launch_bounds__(256) void aggressor_mfma(float* sink, int iters) {
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i];
    if (t == 12345.678f) sink[0] = t;
}

__global__ __launch_bounds__(256) void aggressor_valu(float* sink, int iters) {
    float x = (float)threadIdx.x, y = 1.0001f;
    for (int it = 0; it < iters * 16; ++it) { x = x * y + 0.5f; y = y * 0.99999f + 1e-6f; }
    if (x == 12345.678f) sink[0] = x;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    float* in; unsigned long long* counts; float* sink;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&counts, 32)); CK(hipMalloc(&sink, 16));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 7 + 3) % 9 - 4);
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const int nv = 1024, na = 1024;      // 4 workgroups of each kernel per CU: both co-resident on every SIMD
    float* big; CK(hipMalloc(&big, 1048576 * 4 + 65536)); CK(hipMemset(big, 0, 1048576 * 4 + 65536));
    for (int arm = 0; arm < 6; ++arm) {
        CK(hipMemset(counts, 0, 32));
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        if (arm == 1) hipLaunchKernelGGL(aggressor_mfma, dim3(na), dim3(256), 0, s2, sink, iters * 3);
        if (arm == 2) hipLaunchKernelGGL(aggressor_valu, dim3(na), dim3(256), 0, s2, sink, iters);
        if (arm == 4 || arm == 5) hipLaunchKernelGGL(aggressor_gemm_like, dim3(na), dim3(256), 0, s2, big, sink, iters / 4);
        CK(hipEventRecord(a, s1));
        if (arm >= 3 && arm != 5) hipLaunchKernelGGL(victim_loads, dim3(nv), dim3(256), 0, s1, in, counts, iters);
        else hipLaunchKernelGGL(victim, dim3(nv), dim3(256), 0, s1, in, counts, iters);
        CK(hipEventRecord(b, s1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long c[4]; CK(hipMemcpy(c, counts, 32, hipMemcpyDeviceToHost));
        printf("{\"arm\": \"%s\", \"lane_iterations\": %.0f, \"packed_low_element_wrong\": %llu, \"packed_high_element_wrong\": %llu, \"of_which_in_lanes_48_63\": %llu, \"victim_ms\": %.1f}\n",
               arm == 0 ? "victim alone" : arm == 1 ? "victim beside an MFMA kernel on a second stream" : arm == 2 ? "victim beside a VALU-only kernel on a second stream"
               : arm == 3 ? "victim with per-iteration 16-byte loads, alone" : arm == 4 ? "victim with loads beside a GEMM-shaped kernel (LDS-DMA, barriers, MFMA)" : "register victim beside the GEMM-shaped kernel",
               (double)nv * 256 * iters, c[0], c[1], c[2], ms);
        fflush(stdout);
    }
    return 0;
}
