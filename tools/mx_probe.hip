// Probe of v_mfma_scale_f32_32x32x64_f8f6f4's operand layout (one wave): D = A(32x64 fp8, scaled) x B(64x32 fp8, scaled).
// in: a[64 lanes][8 dwords], b[64][8], sa[64] (scale dword per lane), sb[64], opsel_a/b compile-time variants; out d[64][16].
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int OA, int OB>
__global__ void probe(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 0, 0, OA, sa[l], OB, sb[l]);
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}
extern "C" int mx_probe(const void* a, const void* b, const void* sa, const void* sb, void* d, int oa, int ob) {
    if (oa == 0 && ob == 0) hipLaunchKernelGGL((probe<0, 0>), 1, 64, 0, 0, (const i32x8*)a, (const i32x8*)b, (const int*)sa, (const int*)sb, (float*)d);
    else if (oa == 1 && ob == 1) hipLaunchKernelGGL((probe<1, 1>), 1, 64, 0, 0, (const i32x8*)a, (const i32x8*)b, (const int*)sa, (const int*)sb, (float*)d);
    else if (oa == 2 && ob == 2) hipLaunchKernelGGL((probe<2, 2>), 1, 64, 0, 0, (const i32x8*)a, (const i32x8*)b, (const int*)sa, (const int*)sb, (float*)d);
    else if (oa == 3 && ob == 3) hipLaunchKernelGGL((probe<3, 3>), 1, 64, 0, 0, (const i32x8*)a, (const i32x8*)b, (const int*)sa, (const int*)sb, (float*)d);
    else return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
}
