// Probe (gfx950), round 5: the DEPENDENT packed-fp32 pair whose consumer takes the producer's HIGH half into its LOW lane, issued back to back -- the instruction pattern hipcc
// emits for utx_qkv_post's rotary arithmetic and the one the two-stream corruption was traced to (tools/two_stream_dissect.py: the wrong Q elements equal a0 c q, i.e. the
// low lane of `v_pk_add_f32 r, t0, t1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]` saw 0 for t1's high half, which the v_pk_mul_f32 right in front of it had just
// written; lanes 50-63 of a wave; 1-2 % of the forwards of the two-stream fp8 plan, 0 of 5000 once the products and the sums are separated -- profiles/r05_two_stream_*.log).
// round 4's probe (tools/pk_fp32_mfma_probe.hip) used plain packed operations in separate asm statements (hipcc pads statement boundaries) and saw nothing.
//   victim    : ONE asm statement = the library's four instructions, back to back, on exactly representable operands; a scalar re-computation; lanes whose packed and scalar
//               results differ are counted, and how many of those equal the "second product missing" signature;
//   aggressors: as in round 4's probe (MFMA loop, VALU loop, GEMM-shaped loop with LDS-DMA / barriers / MFMA) on a second stream, co-resident on every SIMD.
// build + run:  hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_probe.hip -o /tmp/pk_opsel_probe && /tmp/pk_opsel_probe [iters]
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
    float a0 = in[(gtid * 4 + 0) & 4095], a1 = in[(gtid * 4 + 1) & 4095], c = in[(gtid * 4 + 2) & 4095], s = in[(gtid * 4 + 3) & 4095];
    unsigned long long bad_lo = 0, bad_hi = 0, bad_top = 0, bad_sig = 0;
    for (int it = 0; it < iters; ++it) {
        f2 a = {a0, a1}, cs = {c, 7.f}, sn = {s, -5.f}, t0, r0, r1;
        // the library's stream (csrc/dit_elementwise.hip built with packed fp32, qkv_post_kernel): t0 = c a ; a = s a (in place) ; r0.lo = t0.lo - a.hi ; r1.hi = t0.hi + a.lo
        asm volatile("v_pk_mul_f32 %[t0], %[cs], %[a] op_sel_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %[a], %[sn], %[a] op_sel_hi:[0,1]\n\t"
                     "v_pk_add_f32 %[r0], %[t0], %[a] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %[r1], %[t0], %[a] op_sel:[0,1] op_sel_hi:[1,0]"
                     : [t0] "=&v"(t0), [a] "+v"(a), [r0] "=&v"(r0), [r1] "=&v"(r1) : [cs] "v"(cs), [sn] "v"(sn));
        float u0, u1, w0, w1, e0, e1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(a0), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(a1), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(a1), "v"(s));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a0), "v"(s));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(u0), "v"(w0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(u1), "v"(w1));
        const bool lo = __float_as_uint(r0[0]) != __float_as_uint(e0);
        const bool hi = __float_as_uint(r1[1]) != __float_as_uint(e1);
        bad_lo += lo; bad_hi += hi; bad_top += (lo || hi) && lane >= 48;
        bad_sig += lo && (__float_as_uint(r0[0]) == __float_as_uint(u0));      // the second product missing: r0 = c a0
        a0 = (float)((int)(e0 + (float)it) & 7) - 3.f; a1 = (float)((int)(e1 - (float)it) & 7) - 4.f;
        if (a1 == 0.f) a1 = 2.f;
        if (s == 0.f) s = 1.f;                                                   // the signature must differ from the right answer
    }
    if (bad_lo) atomicAdd(&counts[0], bad_lo);
    if (bad_hi) atomicAdd(&counts[1], bad_hi);
    if (bad_top) atomicAdd(&counts[2], bad_top);
    if (bad_sig) atomicAdd(&counts[3], bad_sig);
}

// victim, second form: the two table operands (c, s) arrive by 16-byte global loads issued INSIDE the statement and waited for with counted s_waitcnt vmcnt right in front of
// the packed multiplies that consume them -- the library's stream again (its cos / sin float4 loads, `s_waitcnt vmcnt(2)` / `vmcnt(1)` in front of the two v_pk_mul_f32): here
// the waits really stall (the loads are issued just before), so a multiply issues in the cycles in which the returning load still writes its last lanes
__global__ __launch_bounds__(256) void victim_loads(const float* __restrict__ in, unsigned long long* counts, int iters) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned long long bad_lo = 0, bad_hi = 0, bad_top = 0, bad_sig = 0;
    unsigned idx = (unsigned)gtid * 4u;
    float a0 = in[(gtid * 4 + 0) & 4095], a1 = in[(gtid * 4 + 1) & 4095];
    if (a1 == 0.f) a1 = 2.f;
    for (int it = 0; it < iters; ++it) {
        idx = (idx * 1664525u + 1013904223u);
        const float* pc = in + ((idx >> 8) & 4092u & ~3u);
        const float* ps = in + ((idx >> 20) & 4092u & ~3u);
        f2 a = {a0, a1}, t0, r0, r1;
        float c, s;
        asm volatile("global_load_dwordx4 v[200:203], %[pc], off\n\t"
                     "global_load_dwordx4 v[204:207], %[ps], off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_pk_mul_f32 %[t0], v[200:201], %[a] op_sel_hi:[0,1]\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_pk_mul_f32 %[a], v[204:205], %[a] op_sel_hi:[0,1]\n\t"
                     "v_cvt_pk_bf16_f32 v208, v202, v203\n\t"
                     "v_pk_add_f32 %[r0], %[t0], %[a] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %[r1], %[t0], %[a] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                     "v_mov_b32 %[c], v200\n\t"
                     "v_mov_b32 %[s], v204"
                     : [t0] "=&v"(t0), [a] "+v"(a), [r0] "=&v"(r0), [r1] "=&v"(r1), [c] "=&v"(c), [s] "=&v"(s) : [pc] "v"(pc), [ps] "v"(ps)
                     : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "memory");
        float u0, u1, w0, w1, e0, e1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(a0), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(a1), "v"(c));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(a1), "v"(s));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a0), "v"(s));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(u0), "v"(w0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(u1), "v"(w1));
        const bool lo = __float_as_uint(r0[0]) != __float_as_uint(e0), hi = __float_as_uint(r1[1]) != __float_as_uint(e1);
        bad_lo += lo; bad_hi += hi; bad_top += (lo || hi) && lane >= 48;
        bad_sig += lo && (__float_as_uint(r0[0]) == __float_as_uint(u0)) && w0 != 0.f;
        a0 = (float)((int)(e0 + (float)it) & 7) - 3.f; a1 = (float)((int)(e1 - (float)it) & 7) - 4.f;
        if (a1 == 0.f) a1 = 2.f;
    }
    if (bad_lo) atomicAdd(&counts[0], bad_lo);
    if (bad_hi) atomicAdd(&counts[1], bad_hi);
    if (bad_top) atomicAdd(&counts[2], bad_top);
    if (bad_sig) atomicAdd(&counts[3], bad_sig);
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

// aggressor, MX form: the scaled fp8 MFMA of the fp8 plan's text-half GEMMs (v_mfma_scale_f32_32x32x64_f8f6f4: 16 passes, eight operand registers a side + scales) -- the
// library's failure rate is 1-2 % of the forwards with these neighbours and 1 in 2000 with the bf16 MFMA
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void aggressor_mx(float* sink, int iters) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + (int)((threadIdx.x + i) & 3) * 0x01010101; b[i] = 0x3c3c3c3c - (int)((threadIdx.x * 3 + i) & 3) * 0x01010101; }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
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
    for (int use_loads = 0; use_loads < 2; ++use_loads)
    for (int arm = 0; arm < 9; ++arm) {
        CK(hipMemset(counts, 0, 32));
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        if (arm == 1) hipLaunchKernelGGL(aggressor_mfma, dim3(na), dim3(256), 0, s2, sink, iters * 3);
        if (arm == 2) hipLaunchKernelGGL(aggressor_valu, dim3(na), dim3(256), 0, s2, sink, iters);
        if (arm == 3) hipLaunchKernelGGL(aggressor_gemm_like, dim3(na), dim3(256), 0, s2, big, sink, iters / 4);
        if (arm == 7) hipLaunchKernelGGL(aggressor_mx, dim3(na), dim3(256), 0, s2, sink, iters * 2);
        CK(hipEventRecord(a, s1));
        if (use_loads) hipLaunchKernelGGL(victim_loads, dim3(nv), dim3(256), 0, s1, in, counts, iters / 8);
        else hipLaunchKernelGGL(victim, dim3(nv), dim3(256), 0, s1, in, counts, iters);
        CK(hipEventRecord(b, s1));
        // arms 4-6: a STORM of short-lived workgroups beside the victim (waves of another kernel being launched onto / retired from the victim's SIMDs all the time, as the text
        // half's small GEMMs are at the tail of the image half's projection): 3000 launches of ~20 us each
        if (arm >= 4 && arm != 7) for (int k = 0; k < 3000; ++k) {
            if (arm == 4) hipLaunchKernelGGL(aggressor_mfma, dim3(256), dim3(256), 0, s2, sink, 1500);
            if (arm == 5) hipLaunchKernelGGL(aggressor_gemm_like, dim3(256), dim3(256), 0, s2, big, sink, 40);
            if (arm == 6) hipLaunchKernelGGL(aggressor_valu, dim3(512), dim3(256), 0, s2, sink, 300);
            if (arm == 8) hipLaunchKernelGGL(aggressor_mx, dim3(256), dim3(256), 0, s2, sink, 800);
        }
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long c[4]; CK(hipMemcpy(c, counts, 32, hipMemcpyDeviceToHost));
        printf("{\"victim\": \"%s\", \"arm\": \"%s\", \"lane_iterations\": %.0f, \"low_lane_wrong\": %llu, \"of_which_second_product_missing\": %llu, \"high_lane_wrong\": %llu, \"wrong_in_lanes_48_63\": %llu, \"victim_ms\": %.1f}\n",
               use_loads ? "operands by in-statement loads + counted waits" : "operands in registers", arm == 0 ? "victim alone" : arm == 1 ? "victim beside an MFMA kernel on a second stream" : arm == 2 ? "victim beside a VALU-only kernel on a second stream"
               : arm == 3 ? "victim beside a GEMM-shaped kernel (LDS-DMA, barriers, MFMA)" : arm == 4 ? "victim beside 3000 short MFMA launches" : arm == 5 ? "victim beside 3000 short GEMM-shaped launches"
               : arm == 6 ? "victim beside 3000 short VALU launches" : arm == 7 ? "victim beside an MX scaled-MFMA kernel (v_mfma_scale_f32_32x32x64_f8f6f4)" : "victim beside 3000 short MX scaled-MFMA launches", (double)nv * 256 * (use_loads ? iters / 8 : iters), c[0], c[3], c[1], c[2], ms);
        fflush(stdout);
    }
    return 0;
}
