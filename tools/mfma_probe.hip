// Empirical check of the gfx950 MFMA 32x32x16 bf16 fragment layouts assumed by attention.hip / gemm.hip.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const float* A, const float* B, float* Cout) {
    // assumed operand layout: lane l holds A[l&31][8*(l>>5)+e], B[8*(l>>5)+e][l&31]
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + e];
        b[e] = (__bf16)B[(8 * (l >> 5) + e) * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    // assumed C layout: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
    for (int r = 0; r < 16; ++r) Cout[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
    float hA[32 * 16], hB[16 * 32], hC[32 * 32], ref[32 * 32];
    srand(1);
    for (int i = 0; i < 32 * 16; ++i) hA[i] = (float)((rand() % 17) - 8);
    for (int i = 0; i < 16 * 32; ++i) hB[i] = (float)((rand() % 13) - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32 * 32; ++i) if (fabsf(hC[i] - ref[i]) > 1e-3f) ++bad;
    printf("MFMA_32x32x16_bf16 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    if (bad) { for (int i = 0; i < 4; ++i) { for (int j = 0; j < 8; ++j) printf("%6.0f/%-6.0f ", hC[i*32+j], ref[i*32+j]); printf("\n"); } }
    return bad ? 1 : 0;
}
