// LBVH ray-mesh intersector (gfx950) -- replaces the Slang/CUDA kernels of
// TextureTools/texturetools/raytracing/rt_aprmis (bvhworkers/*.slang, bvhhelpers.py:20-84), which
// PBRMesh.optix.intersects_closest reaches (mesh/structure_v2.py:73-77, renderer_inverse.py:321).
//
// Build (bit-identical tree to oracle/geom_ref.c::utxref_bvh_build):
//   per-triangle AABB + scene extent (get_elements.slang:1-39, bvhhelpers.py:29-35)
//   -> 30-bit Morton code of the AABB centre (lbvh_morton_codes.slang:24-79)
//   -> STABLE radix sort by code (the reference uses a single-workgroup sort that hard-codes 32-wide
//      subgroups, lbvh_single_radixsort.slang:3 -- not usable on wave64; here: rocPRIM device radix sort)
//   -> Karras-2012 hierarchy with index tie-break (lbvh_hierarchy.slang:31-244)
//   -> bottom-up AABB union in ONE launch with per-node arrival counters (the reference launches one
//      kernel per tree level, bvhhelpers.py:74-78); inter-workgroup hand-off follows the agent-scope
//      release / acquire recipe (guide G16): stores -> __threadfence() -> atomic arrive; the second
//      arriver reads the sibling box with agent-scope (L1-bypassing) loads.  min/max unions are exact,
//      so the result does not depend on arrival order.
// Traversal reproduces intersect_test2.slang:14-146,270-309 including its quirks (see oracle header).
// Compiled with -ffp-contract=off.
#include "common.h"
#include "kernels.h"
#include <mutex>
#include <vector>
#include "bvh_device.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>


__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __uint_as_float(u); }

__global__ __launch_bounds__(256) void bvh_init_kernel(unsigned* extent, int* counter, int n_internal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3) extent[i] = 0xffffffffu;       // mins
    else if (i < 6) extent[i] = 0u;            // maxs
    for (int j = i; j < n_internal; j += gridDim.x * blockDim.x) counter[j] = 0;
}

__global__ __launch_bounds__(256) void bvh_elements_kernel(const float* vert, const int* faces, int F, float* ebox, unsigned* extent) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (f < F) {
        float a[3] = {1e9f, 1e9f, 1e9f}, b[3] = {-1e9f, -1e9f, -1e9f};
        for (int i = 0; i < 3; ++i) {
            const float* v = vert + 3 * (long)faces[3 * f + i];
            for (int k = 0; k < 3; ++k) { a[k] = fminf(a[k], v[k]); b[k] = fmaxf(b[k], v[k]); }
        }
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(a[k], b[k]); mx[k] = fmaxf(a[k], b[k]);
            ebox[6 * (long)f + k] = mn[k]; ebox[6 * (long)f + 3 + k] = mx[k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        float lo = mn[k], hi = mx[k];
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&extent[k], f2ord(lo)); atomicMax(&extent[3 + k], f2ord(hi)); }
    }
}

__device__ __forceinline__ unsigned expand_bits(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ unsigned morton3d(float x, float y, float z) {
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((unsigned)x) * 4 + expand_bits((unsigned)y) * 2 + expand_bits((unsigned)z);
}

__global__ __launch_bounds__(256) void bvh_morton_kernel(const float* ebox, int F, const unsigned* extent, unsigned* codes, int* idx) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float m[3];
    for (int k = 0; k < 3; ++k) {
        const float gmin = ord2f(extent[k]), gmax = ord2f(extent[3 + k]);
        const float lo = ebox[6 * (long)f + k], hi = ebox[6 * (long)f + 3 + k];
        const float center = lo + 0.5f * (hi - lo);
        m[k] = (center - gmin) / (gmax - gmin);
    }
    codes[f] = morton3d(m[0], m[1], m[2]);
    idx[f] = f;
}

__device__ __forceinline__ int find_msb(unsigned v) { return v ? 31 - __clz(v) : -1; }
__device__ __forceinline__ int delta_fn(int i, unsigned codeI, int j, int n, const unsigned* codes) {
    if (j < 0 || j > n - 1) return -1;
    const unsigned codeJ = codes[j];
    if (codeI == codeJ) return 32 + 31 - find_msb((unsigned)i ^ (unsigned)j);
    return 31 - find_msb(codeI ^ codeJ);
}

__global__ __launch_bounds__(256) void bvh_hierarchy_kernel(int n, const unsigned* codes, const int* sorted_idx, const float* ebox,
                                                            int* info, float* aabb, int* parent) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int LEAF = n - 1;
    {   // leaf
        const int e = sorted_idx[g];
        info[3 * (long)(LEAF + g) + 0] = 0; info[3 * (long)(LEAF + g) + 1] = 0; info[3 * (long)(LEAF + g) + 2] = e;
        for (int k = 0; k < 6; ++k) aabb[6 * (long)(LEAF + g) + k] = ebox[6 * (long)e + k];
    }
    if (g == 0) parent[0] = 0;
    if (g >= n - 1) return;
    const int idx = g;
    const unsigned code = codes[idx];
    const int dL = delta_fn(idx, code, idx - 1, n, codes), dR = delta_fn(idx, code, idx + 1, n, codes);
    const int d = (dR >= dL) ? 1 : -1;
    const int dmin = dL < dR ? dL : dR;
    int lmax = 2;
    while (delta_fn(idx, code, idx + lmax * d, n, codes) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t > 0; t >>= 1)
        if (delta_fn(idx, code, idx + (l + t) * d, n, codes) > dmin) l += t;
    const int jdx = idx + l * d;
    const int first = idx < jdx ? idx : jdx, last = idx > jdx ? idx : jdx;
    const unsigned fcode = codes[first];
    const int common = delta_fn(first, fcode, last, n, codes);
    int split = first, stride = last - first;
    do {
        stride = (stride + 1) >> 1;
        const int ns = split + stride;
        if (ns < last) { if (delta_fn(first, fcode, ns, n, codes) > common) split = ns; }
    } while (stride > 1);
    const int cA = (split == first) ? LEAF + split : split;
    const int cB = (split + 1 == last) ? LEAF + split + 1 : split + 1;
    info[3 * (long)idx + 0] = cA; info[3 * (long)idx + 1] = cB; info[3 * (long)idx + 2] = 0;
    parent[cA] = idx; parent[cB] = idx;
}

__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void bvh_refit_kernel(int n, const int* info, float* aabb, const int* parent, int* counter) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n || n < 2) return;
    int node = parent[(n - 1) + g];
    // the leaf boxes were written by the previous launch -> visible
    for (;;) {
        __threadfence();  // release: this thread's box stores (if any) before the arrival
        const int arrived = atomicAdd(&counter[node], 1);
        if (arrived == 0) return;  // first arriver leaves; the second one owns the node
        __threadfence();  // acquire side (plus L1-bypassing loads below)
        const int L = info[3 * (long)node], R = info[3 * (long)node + 1];
        float bb[6];
        for (int k = 0; k < 3; ++k) {
            bb[k] = fminf(ld_agent(&aabb[6 * (long)L + k]), ld_agent(&aabb[6 * (long)R + k]));
            bb[3 + k] = fmaxf(ld_agent(&aabb[6 * (long)L + 3 + k]), ld_agent(&aabb[6 * (long)R + 3 + k]));
        }
        for (int k = 0; k < 6; ++k) __hip_atomic_store(&aabb[6 * (long)node + k], bb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (node == 0) return;
        node = parent[node];
    }
}

// packed tree for the stackless traversal (bvh_device.h): one thread per node; esc by walking up until the node is a second child
__global__ __launch_bounds__(256) void bvh_pack_kernel(int n, const int* info, const float* aabb, const int* parent, const float* vert,
                                                       const int* faces, float4* nodes, float4* tris, int* depth) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int nn = 2 * n - 1;
    if (g >= nn) return;
    const int L = info[3 * (long)g], R = info[3 * (long)g + 1];
    const bool leaf = (L == 0 && R == 0);
    int esc = -1, steps = 0;
    for (int cur = g; cur != 0; ++steps) {
        const int P = parent[cur];
        if (cur == info[3 * (long)P + 1]) { esc = info[3 * (long)P]; break; }
        cur = P;
    }
    if (leaf) {
        int d = 0;
        for (int cur = g; cur != 0; cur = parent[cur]) ++d;
        atomicMax(depth, d);
        const int prim = info[3 * (long)g + 2];
        const int* f = faces + 3 * (long)prim;
        for (int k = 0; k < 3; ++k)
            tris[3 * (long)prim + k] = make_float4(vert[3 * (long)f[k]], vert[3 * (long)f[k] + 1], vert[3 * (long)f[k] + 2], 0.f);
    }
    const int link = leaf ? ~info[3 * (long)g + 2] : R;
    nodes[2 * (long)g] = make_float4(aabb[6 * (long)g], aabb[6 * (long)g + 1], aabb[6 * (long)g + 2], aabb[6 * (long)g + 3]);
    nodes[2 * (long)g + 1] = make_float4(aabb[6 * (long)g + 4], aabb[6 * (long)g + 5], __int_as_float(link), __int_as_float(esc));
}

// ---------------------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bvh_trace_packed_kernel(const float4* nodes, const float4* tris, const float* ro, const float* rd, long R,
                                                               int* tid, unsigned long long* visited) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nv = 0;
    if (i < R) {
        const float o[3] = {ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]};
        const float d[3] = {rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]};
        tid[i] = bvh_trace_packed(nodes, tris, o, d, visited ? &nv : nullptr);
    }
    if (visited) {      // diagnostics (bench: nodes visited per ray): wave sum, one atomic per wave
        for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(visited, (unsigned long long)nv);
    }
}

__global__ __launch_bounds__(256) void bvh_trace_kernel(const int* info, const float* aabb, const float* vert, const int* faces,
                                                        const float* ro, const float* rd, long R, int* tid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float o[3] = {ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]};
    const float d[3] = {rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]};
    tid[i] = bvh_trace_one(info, aabb, vert, faces, o, d);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define HCHK(e) do { if ((e) != hipSuccess) return -7; } while (0)

// Every device array of the tree lives in ONE block of memory: the caller's (utx_bvh_build_ws: a torch allocation, nothing allocated or freed on the path, no host
// synchronisation) or, for the plain utx_bvh_build, one hipMalloc owned by the handle.  The tree depth -- which decides the traversal the launches use -- comes back
// through a pinned host word behind an event and is read LAZILY, at the first launch that needs it (bvh_depth below).
static size_t bvh_align(size_t x) { return (x + 255) & ~(size_t)255; }
namespace {
// pinned words the builds copy their tree depth into.  Grows in chunks of 1024 words (a long-lived service may hold any number of meshes); chunks are never freed and the
// pool itself is a leaked heap singleton: a utx_bvh_free that runs during static destruction still finds its mutex and vectors alive.
struct DepthPool {
    std::mutex mu;
    std::vector<int*> chunks;
    std::vector<int*> free_slots;
    int* take() {
        std::lock_guard<std::mutex> lock(mu);
        if (free_slots.empty()) {
            int* c = nullptr;
            if (hipHostMalloc((void**)&c, 1024 * sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            chunks.push_back(c);
            for (int i = 1023; i >= 0; --i) free_slots.push_back(c + i);
        }
        int* const p = free_slots.back(); free_slots.pop_back();
        return p;
    }
    void give(int* p) {
        std::lock_guard<std::mutex> lock(mu);
        for (int* c : chunks)
            if (p >= c && p < c + 1024) { free_slots.push_back(p); return; }
    }
};
static DepthPool& depth_pool() { static DepthPool* const pool = new DepthPool(); return *pool; }
}
static size_t bvh_sort_bytes(int F) {
    size_t n = 0;
    (void)rocprim::radix_sort_pairs(nullptr, n, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)F, 0, 32, (hipStream_t)0);
    return n ? n : 16;
}
extern "C" size_t utx_bvh_workspace_bytes_impl(int F) {
    if (F < 1) return 0;
    const size_t nn = 2 * (size_t)F - 1;
    return bvh_align(nn * 3 * sizeof(int)) + bvh_align(nn * 6 * sizeof(float)) + bvh_align((size_t)F * 6 * sizeof(float)) + 4 * bvh_align((size_t)F * 4) +
           bvh_align(nn * sizeof(int)) + bvh_align((size_t)(F > 1 ? F - 1 : 1) * sizeof(int)) + bvh_align(8 * sizeof(unsigned)) + bvh_align(nn * 2 * sizeof(float4)) +
           bvh_align((size_t)F * 3 * sizeof(float4)) + bvh_align(sizeof(int)) + bvh_align(bvh_sort_bytes(F));
}

extern "C" int utx_bvh_build_ws_impl(const float* verts, int V, const int* faces, int F, void* work, size_t work_bytes, utx_bvh** out, hipStream_t stream) {
    (void)V;
    if (F < 1 || !out || !work || ((uintptr_t)work & 255) || work_bytes < utx_bvh_workspace_bytes_impl(F)) return -2;
    utx_bvh* b = new utx_bvh();      // value-initialised: every pointer null, the event null
    b->F = F; b->verts = verts; b->faces = faces; b->depth.store(-1, std::memory_order_relaxed);
    const size_t nn = 2 * (size_t)F - 1;
    char* w = (char*)work;
    auto take = [&](size_t bytes) { void* p_ = w; w += bvh_align(bytes); return p_; };
    b->info = (int*)take(nn * 3 * sizeof(int));
    b->aabb = (float*)take(nn * 6 * sizeof(float));
    b->ebox = (float*)take((size_t)F * 6 * sizeof(float));
    b->codes = (unsigned*)take((size_t)F * 4); b->codes_sorted = (unsigned*)take((size_t)F * 4);
    b->idx = (int*)take((size_t)F * 4); b->idx_sorted = (int*)take((size_t)F * 4);
    b->parent = (int*)take(nn * sizeof(int));
    b->counter = (int*)take((size_t)(F > 1 ? F - 1 : 1) * sizeof(int));
    b->extent = (unsigned*)take(8 * sizeof(unsigned));
    b->nodes = (float4*)take(nn * 2 * sizeof(float4));
    b->tris = (float4*)take((size_t)F * 3 * sizeof(float4));
    b->depth_dev = (int*)take(sizeof(int));
    b->sort_tmp_bytes = bvh_sort_bytes(F);
    b->sort_tmp = take(b->sort_tmp_bytes);
#define BCHK(e) do { if ((e) != hipSuccess) { utx_bvh_free_impl(b); return -7; } } while (0)
    b->depth_host = depth_pool().take();      // a pinned word of the (growing) pool; nullptr only when pinned host memory itself is exhausted
    if (!b->depth_host) { utx_bvh_free_impl(b); return -7; }
    BCHK(hipEventCreateWithFlags(&b->depth_ready, hipEventDisableTiming));
    BCHK(hipMemsetAsync(b->depth_dev, 0, sizeof(int), stream));
    const int nb = (F + 255) / 256;
    hipLaunchKernelGGL(bvh_init_kernel, dim3(nb), dim3(256), 0, stream, b->extent, b->counter, F - 1);
    hipLaunchKernelGGL(bvh_elements_kernel, dim3(nb), dim3(256), 0, stream, verts, faces, F, b->ebox, b->extent);
    hipLaunchKernelGGL(bvh_morton_kernel, dim3(nb), dim3(256), 0, stream, b->ebox, F, b->extent, b->codes, b->idx);
    BCHK(rocprim::radix_sort_pairs(b->sort_tmp, b->sort_tmp_bytes, b->codes, b->codes_sorted, b->idx, b->idx_sorted, (size_t)F, 0, 32, stream));
    hipLaunchKernelGGL(bvh_hierarchy_kernel, dim3(nb), dim3(256), 0, stream, F, b->codes_sorted, b->idx_sorted, b->ebox, b->info, b->aabb, b->parent);
    hipLaunchKernelGGL(bvh_refit_kernel, dim3(nb), dim3(256), 0, stream, F, b->info, b->aabb, b->parent, b->counter);
    hipLaunchKernelGGL(bvh_pack_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, stream, F, b->info, b->aabb, b->parent, verts, faces,
                       b->nodes, b->tris, b->depth_dev);
    if (hipGetLastError() != hipSuccess) { utx_bvh_free_impl(b); return -4; }
    // the tree depth decides, once per mesh, which traversal the launches use: copied to the pinned word, read by bvh_depth() when a launch first asks (no wait here)
    BCHK(hipMemcpyAsync(b->depth_host, b->depth_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    if (hipEventRecord(b->depth_ready, stream) != hipSuccess) {
        // cold path: the copy into the pinned word is enqueued but no event stands behind it -- drain the stream before the word goes back to the pool (utx_bvh_free's
        // hipEventSynchronize on a never-recorded event would return at once and a later tree could read a stale or foreign depth from the slot)
        (void)hipStreamSynchronize(stream);
        utx_bvh_free_impl(b);
        return -7;
    }
#undef BCHK
    *out = b;
    return 0;
}

// the allocating form: ONE hipMalloc owned by the handle (utx_bvh_free releases it); depth known on return, as this entry point always promised
extern "C" int utx_bvh_build_impl(const float* verts, int V, const int* faces, int F, utx_bvh** out, hipStream_t stream) {
    if (F < 1 || !out) return -2;
    void* blk = nullptr;
    const size_t bytes = utx_bvh_workspace_bytes_impl(F);
    HCHK(hipMalloc(&blk, bytes));
    utx_bvh* b = nullptr;
    const int rc = utx_bvh_build_ws_impl(verts, V, faces, F, blk, bytes, &b, stream);
    if (rc != 0) { (void)hipFree(blk); return rc; }
    b->owned = blk;
    if (utx_bvh_depth_impl(b) < 0) { utx_bvh_free_impl(b); return -7; }
    *out = b;
    return 0;
}

extern "C" void utx_bvh_free_impl(utx_bvh* b) {
    if (!b) return;
    if (b->depth_ready && b->depth.load(std::memory_order_relaxed) < 0 && b->depth_host) (void)hipEventSynchronize(b->depth_ready);      // the copy into the pinned word must have landed before the word is handed on
    if (b->depth_ready) (void)hipEventDestroy(b->depth_ready);
    if (b->depth_host) depth_pool().give(b->depth_host);
    if (b->owned) (void)hipFree(b->owned);      // the caller's workspace (utx_bvh_build_ws) is the caller's to release
    delete b;
}

extern "C" int utx_bvh_arrays_impl(utx_bvh* b, int** info, float** aabb, unsigned** codes_sorted, int** idx_sorted) {
    if (!b) return -2;
    if (info) *info = b->info;
    if (aabb) *aabb = b->aabb;
    if (codes_sorted) *codes_sorted = b->codes_sorted;
    if (idx_sorted) *idx_sorted = b->idx_sorted;
    return b->F;
}

// longest root-to-leaf path; the first call waits for the build's copy of it (an event behind the build's last kernel), later calls are a load.  -1: the wait failed.
// The lazy write of b->depth is a relaxed atomic store of a value every thread computes identically: two threads launching on one fresh handle both wait on the event and
// both store the same word (utx_bvh.depth is a std::atomic<int>).
extern "C" int utx_bvh_depth_impl(utx_bvh* b) {
    int d = b->depth.load(std::memory_order_relaxed);
    if (d < 0) {
        if (hipEventSynchronize(b->depth_ready) != hipSuccess) return -1;
        d = *b->depth_host;
        b->depth.store(d, std::memory_order_relaxed);
    }
    return d;
}

extern "C" int utx_bvh_trace_impl(utx_bvh* b, const float* ro, const float* rd, long R, int* tid, unsigned long long* visited, int force_stack,
                                  hipStream_t stream) {
    if (!b || R <= 0) return -2;
    const int depth = utx_bvh_depth_impl(b);
    if (depth < 0) return -7;      // the build's depth never arrived (event wait failed): no traversal may be chosen on a guess -- as utx_launch_backproject
    if (depth <= UTX_BVH_PACKED_MAX_DEPTH && !force_stack) {
        hipLaunchKernelGGL(bvh_trace_packed_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, b->nodes, b->tris, ro, rd, R, tid, visited);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    if (visited) return -2;     // the node count is a diagnostic of the packed traversal
    hipLaunchKernelGGL(bvh_trace_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, b->info, b->aabb, b->verts, b->faces, ro, rd, R, tid);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
