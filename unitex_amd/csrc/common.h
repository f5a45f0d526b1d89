// Shared device helpers for the gfx950 kernels of libunitex_hip.so.
// CDNA4 only: wave = 64 lanes, MFMA 32x32x16 bf16, 160 KiB LDS / CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifndef UTX_BF16_T
#define UTX_BF16_T
typedef uint16_t bf16_t;  // raw storage type used across the C ABI
#endif

#define UTX_WAVE 64

// fp32 -> bf16 round-to-nearest-even on raw bits (NaN kept quiet). Matches torch's
// float->bfloat16 conversion, which the oracle uses.
__device__ __forceinline__ uint16_t f2bf(float f) {
    // gfx950 has a hardware RNE convert (v_cvt_pk_bf16_f32); same rounding as torch's float->bfloat16
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, v);
}
// round an fp32 value through bf16 (used to mirror the reference's bf16 tensor boundaries)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap (8 XCDs; block b is observed to land on XCD b % 8 -- speed only,
// never correctness). Returns the logical work id for hardware block id `bid` so that each
// XCD walks a contiguous chunk of the logical grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, j = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// ---- host side: one-time launch setup is PER DEVICE (hipFuncSetAttribute applies to the current device; CU counts may differ between
// partitions): call sites keep a bit mask indexed by device instead of a process-wide flag
// Thread-safe (a multi-GPU C host runs one thread per device): the mask and the table are atomics; a lost race only repeats the idempotent set-up.
// Device ids >= 64 do not alias slot 0: utx_device() returns -1 for them, the set-up then runs on every call and utx_ncu() asks the runtime each time.
#include <atomic>
static inline int utx_device() { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : -1; }
static inline int utx_ncu() {
    static std::atomic<int> tab[64] = {};
    const int d = utx_device();
    int n = d >= 0 ? tab[d].load(std::memory_order_relaxed) : 0;
    if (!n) {
        int cur = 0; hipDeviceProp_t pr;
        n = (hipGetDevice(&cur) == hipSuccess && hipGetDeviceProperties(&pr, cur) == hipSuccess) ? pr.multiProcessorCount : 256;
        if (d >= 0) tab[d].store(n, std::memory_order_relaxed);
    }
    return n;
}
#define UTX_ONCE_PER_DEVICE(flag_) static std::atomic<unsigned long long> flag_{0}; const int flag_##_dev = utx_device(); \
    if (flag_##_dev < 0 || !((flag_.load(std::memory_order_acquire) >> flag_##_dev) & 1ull))
#define UTX_ONCE_DONE(flag_) do { if (flag_##_dev >= 0) flag_.fetch_or(1ull << flag_##_dev, std::memory_order_release); } while (0)
