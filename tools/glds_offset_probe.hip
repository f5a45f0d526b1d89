// Where does the instruction offset of an LDS-DMA go?  global_load_lds_dwordx4 v_off, s[base:base+1] offset:N with M0 = D:
// does the wave's 1 KB land at LDS D or D + N, and is it fetched from base + v_off or base + v_off + N?  (The 4 x 64 attention stream writes M0 once per 1 KB piece;
// if the offset moves BOTH addresses, one M0 write serves the four pieces of a tile -- six SALU instructions fewer per tile.)
//   hipcc --offload-arch=gfx950 -O2 tools/glds_offset_probe.hip -o tools/bin/glds_offset_probe && tools/bin/glds_offset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint32_t* src, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t voff = threadIdx.x * 16;
    const uint32_t d = 2048;      // M0: LDS byte address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(d), "s"(src) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *src, *out, h[4096], hs[8192];
    for (int i = 0; i < 8192; ++i) hs[i] = i;      // dword i holds i: a fetched value names its global byte address / 4
    hipMalloc(&src, sizeof(hs)); hipMalloc(&out, sizeof(h));
    hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, src, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeefu) { if (first < 0) first = i; ++n; }
    printf("M0 = 2048, inst offset 1024, lane voff = 16 lane: %d dwords written, first at LDS byte %d holding global byte %u (lane 0)\n", n, first * 4, first >= 0 ? h[first] * 4 : 0);
    printf("  -> LDS address %s the instruction offset; global address %s it\n", first * 4 == 2048 ? "IGNORES" : first * 4 == 3072 ? "INCLUDES" : "?", first >= 0 && h[first] * 4 == 0 ? "IGNORES" : first >= 0 && h[first] * 4 == 1024 ? "INCLUDES" : "?");
    return 0;
}
