// L2 -> LDS fill rate of one CU with every CU streaming (gfx950): what access pattern of the 1 KB LDS-DMA pieces sustains the most bytes per
// clock?  The 256x256x64 GEMM needs 64 KB per 2048 MFMA cycles = 32 B/clk/CU; the kernels measure ~23 (profiles/r02_gemm_w4_trace_v1.log).
//   hipcc --offload-arch=gfx950 -O3 tools/l2_fill_probe.hip -o /tmp/l2_fill_probe && /tmp/l2_fill_probe
// Patterns (each wave-instruction moves 64 lanes x 16 B = 1 KB into a lane-linear LDS image):
//   0  contiguous 1 KB                                     (pre-tiled operand)
//   1  8 rows x 128 B, row stride 6144 B                   (gemm_pers.hip: K = 3072 bf16 rows, 64-k K-tile)
//   2  16 rows x 64 B, row stride 6144 B                   (gemm_w4.hip: 32-k sub-stage)
//   3  8 rows x 128 B, row stride 8192 B                   (K = 4096)
//   4  4 rows x 256 B, row stride 6144 B
// Footprint per workgroup: its own 256-row x 6144-B window re-walked along K, as a GEMM tile does (L2 / MALL resident after the first pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

template <int PAT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fill_kernel(const char* buf, long wg_stride, int iters, int depth, long long* cycles, int share, int passes) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // share > 1: `share` workgroups OF THE SAME XCD (blockIdx % 8) read the same window at the same time, as the tiles of a tile row / column
    // do (the operand reuse that makes ~80 % of a GEMM's DMA requests L2 hits); share < 0: |share| neighbouring block ids (= different XCDs)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const long win = share > 0 ? (long)(xcd * 32 + (jx / share) * share) : (long)((blockIdx.x / (-share)) * (-share));
    const char* base0 = buf + win * wg_stride;
    const long pass_stride = 256 * wg_stride;          // a new set of windows every 48 K-tiles (a GEMM moves on to its next tile)
    // window of a workgroup: 512 operand rows (A 256 + B 256), row stride STR; one "K-tile" = 64 KB = 64 pieces of 1 KB
    const long STR = (PAT == 3) ? 8192 : 6144;
    unsigned voff;
    if (PAT == 0) voff = lane * 16;
    else if (PAT == 1 || PAT == 3) voff = (unsigned)((lane >> 3) * STR + (lane & 7) * 16);      // 8 rows x 128 B
    else if (PAT == 2) voff = (unsigned)((lane >> 2) * STR + (lane & 3) * 16);                  // 16 rows x 64 B
    else voff = (unsigned)((lane >> 4) * STR + (lane & 15) * 16);                               // 4 rows x 256 B
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + wave * 1024;
    const int PIECES = 64 / WAVES;          // pieces per wave per 64 KB "K-tile"
    const long long t0 = __builtin_readcyclecounter();
    int outstanding = 0;
    for (int it = 0; it < iters; ++it) {
        const int kt = it % 48;
        const char* base = base0 + (long)((it / 48) % passes) * pass_stride;
        for (int pc = 0; pc < PIECES; ++pc) {
            const int p = wave * PIECES + pc;          // piece 0..63 of this K-tile
            long off;
            if (PAT == 0) off = (long)kt * 65536 + (long)p * 1024;                                   // contiguous stream
            else if (PAT == 1 || PAT == 3) off = (long)(8 * p) * STR + (long)kt * 128;               // rows 8p.., 128 B at column kt
            else if (PAT == 2) off = (long)(16 * (p & 31)) * STR + (long)kt * 128 + 64 * (p >> 5);   // two 32-k sub-stages per K-tile
            else off = (long)(4 * p) * STR + (long)kt * 256;                                         // 256 rows x 256 B
            dma16(base + off, voff, lds0 + ((it & 1) * 65536 + pc * WAVES * 1024) % 131072);
            if (++outstanding >= depth) {     // between depth/2 and depth pieces of this wave in flight (waiting after every piece costs a
                                              // wake-up per piece: 35 instead of 100 GB/s/CU -- the kernels wait once per 8 or 16 pieces)
                if (depth == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (depth == 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (depth == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (depth == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (depth == 32) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                outstanding = depth == 2 ? 1 : (depth == 28 ? 24 : depth / 2);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PAT, int WAVES>
static void run(const char* name, const char* buf, long wg_stride, int grid, int share, int depth = 28) {
    const int iters = 480;
    long long* d; hipMalloc(&d, grid * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<PAT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8192);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL((fill_kernel<PAT, WAVES>), dim3(grid), dim3(64 * WAVES), 131072 + 8192, 0, buf, wg_stride, iters, depth, d, share, 2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(grid); hipMemcpy(h.data(), d, grid * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= grid;
        const double bytes = (double)iters * 65536;
        if (rep == 1)
            printf("%-44s waves %d grid %3d share %2d depth %2d (%3d KB in flight/CU) | %7.1f B/clk/CU (shader clk) | %6.1f GB/s/CU | %5.2f TB/s chip | %.3f ms | clk %.2f GHz\n", name, WAVES, grid, share, depth, depth * WAVES,
                   bytes / avg, bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12, ms, avg / (ms * 1e-3) / 1e9);
    }
    hipFree(d);
}

int main() {
    const long wg_stride = 512L * 8192;                 // every workgroup its own 512-row window (rows up to 8192 B apart)
    char* buf; hipMalloc(&buf, 2 * 256 * wg_stride + (1 << 22)); hipMemset(buf, 1, 2 * 256 * wg_stride + (1 << 22));
    for (int share : {1, 8, 32}) {
        run<0, 8>("contiguous 1 KB pieces", buf, wg_stride, 256, share);
        run<1, 8>("8 rows x 128 B, stride 6144 (pers)", buf, wg_stride, 256, share);
        run<2, 8>("16 rows x 64 B, stride 6144 (w4)", buf, wg_stride, 256, share);
        run<3, 8>("8 rows x 128 B, stride 8192", buf, wg_stride, 256, share);
        run<4, 8>("4 rows x 256 B, stride 6144", buf, wg_stride, 256, share);
        run<0, 4>("contiguous 1 KB pieces", buf, wg_stride, 256, share);
        run<1, 4>("8 rows x 128 B, stride 6144", buf, wg_stride, 256, share);
    }
    for (int depth : {2, 4, 8, 16, 32}) run<1, 4>("8 rows x 128 B, stride 6144: latency curve", buf, wg_stride, 256, 8, depth);
    for (int depth : {2, 4, 8, 16, 32}) run<1, 8>("8 rows x 128 B, stride 6144: latency curve", buf, wg_stride, 256, 8, depth);
    run<0, 8>("contiguous, 32 CUs only", buf, wg_stride, 32, 1);
    run<1, 8>("8 rows x 128 B, 32 CUs only", buf, wg_stride, 32, 1);
    return 0;
}
