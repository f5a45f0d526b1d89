// Probe (gfx950): do vector-memory operations of a wave retire IN ORDER through vmcnt, stores included?
//
// The counted waits of the one-wave-per-SIMD GEMM's epilogues (gemm_w4.hip: W4_GATE_PIECE, W4_QK_PIECE) wait for an OLDER load with a literal
// `s_waitcnt vmcnt(N)` whose N counts the younger loads AND stores issued behind it.  That is only sound if a younger store cannot leave the counter
// ahead of the older load.  LLVM's own s_waitcnt insertion for gfx9 / CDNA (one VM_CNT counter, no separate store counter) relies on the same rule;
// this probe looks for a counterexample on the hardware:
//
//   v_mov   d, SENTINEL
//   global_load_dword d <- cold[random line]        (older: HBM + TLB miss, ~ 1-3 us)
//   K x global_store_dword hot[...]                 (younger: L2-resident lines, plain or nontemporal)
//   s_waitcnt vmcnt(K)                              (in-order retirement => the load has landed)
//   v_mov   c, d                                    (read the destination right behind the wait)
//   s_waitcnt vmcnt(0)
//   violation  <=>  c != d  (c still holds the sentinel)
//
// Control arm: the same sequence with vmcnt(K + 1) (= no wait at all) must show stale reads on (nearly) every trial, else the probe proves nothing.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/vmcnt_order_probe.hip -o /tmp/vmcnt_probe && /tmp/vmcnt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr unsigned SENT = 0x7fc0dead;

// MODE 0: K = 1 plain store | 1: K = 4 plain stores | 2: K = 4 nontemporal stores | 3: K = 2 stores + 2 hot (L2-hit) loads, younger than the cold one
// CONTROL: the wait is vmcnt(K + 1): nothing is waited for
template <int MODE, bool CONTROL>
__global__ __launch_bounds__(256) void probe_kernel(const unsigned* __restrict__ cold, size_t cold_mask, unsigned* hot, unsigned long long* counts, int iters) {
    const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long state = 0x9E3779B97F4A7C15ull * (gtid + 1) + 12345;
    unsigned* const h = hot + (size_t)gtid * 4;      // this lane's four hot words (a wave's 256 words = four 256-byte rows... L2 resident after the first trips)
    unsigned long long stale = 0, wrong = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const size_t idx = (size_t)(state >> 20) & cold_mask;
        const unsigned* const src = cold + idx;
        unsigned d, c, e0 = 0, e1 = 0;
        const unsigned x = (unsigned)it;
        if constexpr (MODE == 0) {
            asm volatile("v_mov_b32 %0, %4\n\ts_nop 1\n\t"
                         "global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %5, off\n\t"
                         "s_waitcnt vmcnt(%6)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(d), "=&v"(c) : "v"(src), "v"(h), "v"(SENT), "v"(x), "n"(CONTROL ? 2 : 1) : "memory");
        } else if constexpr (MODE == 1) {
            asm volatile("v_mov_b32 %0, %4\n\ts_nop 1\n\t"
                         "global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %5, off\n\tglobal_store_dword %3, %5, off offset:4\n\t"
                         "global_store_dword %3, %5, off offset:8\n\tglobal_store_dword %3, %5, off offset:12\n\t"
                         "s_waitcnt vmcnt(%6)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(d), "=&v"(c) : "v"(src), "v"(h), "v"(SENT), "v"(x), "n"(CONTROL ? 5 : 4) : "memory");
        } else if constexpr (MODE == 2) {
            asm volatile("v_mov_b32 %0, %4\n\ts_nop 1\n\t"
                         "global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %5, off nt\n\tglobal_store_dword %3, %5, off offset:4 nt\n\t"
                         "global_store_dword %3, %5, off offset:8 nt\n\tglobal_store_dword %3, %5, off offset:12 nt\n\t"
                         "s_waitcnt vmcnt(%6)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(d), "=&v"(c) : "v"(src), "v"(h), "v"(SENT), "v"(x), "n"(CONTROL ? 5 : 4) : "memory");
        } else {
            asm volatile("v_mov_b32 %0, %6\n\ts_nop 1\n\t"
                         "global_load_dword %0, %4, off\n\t"
                         "global_store_dword %5, %7, off\n\tglobal_load_dword %2, %5, off offset:8\n\t"
                         "global_store_dword %5, %7, off offset:4\n\tglobal_load_dword %3, %5, off offset:12\n\t"
                         "s_waitcnt vmcnt(%8)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(d), "=&v"(c), "=&v"(e0), "=&v"(e1) : "v"(src), "v"(h), "v"(SENT), "v"(x), "n"(CONTROL ? 5 : 4) : "memory");
        }
        const unsigned expect = (unsigned)(idx * 2654435761u) ^ 0x5bd1e995u;
        if (c != d) ++stale;
        if (d != expect) ++wrong;
        state += e0 + e1;      // keep the extra loads alive
    }
    if (stale) atomicAdd(&counts[0], stale);
    if (wrong) atomicAdd(&counts[1], wrong);
}

__global__ void fill_kernel(unsigned* cold, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        cold[i] = (unsigned)(i * 2654435761u) ^ 0x5bd1e995u;
}

template <int MODE, bool CONTROL>
static void run(const char* what, const unsigned* cold, size_t mask, unsigned* hot, unsigned long long* counts, int blocks, int iters) {
    CK(hipMemset(counts, 0, 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe_kernel<MODE, CONTROL>), dim3(blocks), dim3(256), 0, 0, cold, mask, hot, counts, iters);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2];
    CK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
    const double trials = (double)blocks * 256 * iters;
    printf("{\"arm\": \"%s\", \"control\": %s, \"lane_trials\": %.0f, \"stale_reads\": %llu, \"wrong_final_values\": %llu, \"ms\": %.1f, \"us_per_trip\": %.2f}\n",
           what, CONTROL ? "true" : "false", trials, h[0], h[1], ms, ms * 1e3 / iters);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
    const size_t n = (size_t)1 << 31;      // 2^31 words = 8 GB: far beyond the L2s / MALL and the TLB reach
    unsigned *cold, *hot;
    unsigned long long* counts;
    CK(hipMalloc(&cold, n * 4));
    CK(hipMalloc(&hot, (size_t)blocks * 256 * 16));
    CK(hipMalloc(&counts, 16));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, cold, n);
    CK(hipMemset(hot, 0, (size_t)blocks * 256 * 16));
    CK(hipDeviceSynchronize());
    run<0, true>("load; 1 store; vmcnt(2) [no wait]", cold, n - 1, hot, counts, blocks, iters);
    run<0, false>("load; 1 store; vmcnt(1)", cold, n - 1, hot, counts, blocks, iters);
    run<1, true>("load; 4 stores; vmcnt(5) [no wait]", cold, n - 1, hot, counts, blocks, iters);
    run<1, false>("load; 4 stores; vmcnt(4)", cold, n - 1, hot, counts, blocks, iters);
    run<2, false>("load; 4 nontemporal stores; vmcnt(4)", cold, n - 1, hot, counts, blocks, iters);
    run<3, true>("load; store, hot load, store, hot load; vmcnt(5) [no wait]", cold, n - 1, hot, counts, blocks, iters);
    run<3, false>("load; store, hot load, store, hot load; vmcnt(4)", cold, n - 1, hot, counts, blocks, iters);
    return 0;
}
