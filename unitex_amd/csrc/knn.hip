// Exact k-nearest-neighbour gather in 3-D for the 'kdtree' back-projection variant
// (reference: TextureTools/texturetools/render/nvdiffrast/renderer_inverse.py:367-433 bake_mv_to_uv_kdtree; the search
// itself is torch_kdtree [3p], pcd/knn/__init__.py:103-113).  HBM-bound integer / fp32 work, no MFMA.
//
// Same structure as the 1-NN fill of texture_post.hip: the source points are binned into a uniform grid (G^3 cells over
// [-1,1]^3, cell ids radix-sorted with rocPRIM), every query walks outward ring by ring and keeps its k best candidates in
// a sorted list; ring r+1 can only hold points at distance >= r * cell, which gives an exact stopping rule.  Source and
// query sets are given densely with byte masks (view pixels / atlas texels), so nothing is compacted on the way in or out.
//   d2 = (dx*dx + dy*dy) + dz*dz in float32; ties are broken towards the lower source index.
// Output per query: the mean of the neighbours' attributes in ascending-distance order (mode 0), or the MVPaint weighting
// (mode 1, arXiv 2411.02336 sec. 3.2 as the reference states it: normalised inverse score times normal cosine).
#include "common.h"
#include "kernels.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#define KNN_G 128
#define KNN_KMAX 32

__device__ __forceinline__ int knn_cell1(float v) {
    int c = (int)floorf((v + 1.0f) * (KNN_G * 0.5f));
    return c < 0 ? 0 : (c > KNN_G - 1 ? KNN_G - 1 : c);
}
__global__ __launch_bounds__(256) void knn_keys_kernel(const float* pos, const unsigned char* mask, long N, unsigned* keys, int* vals) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    unsigned k = 0xffffffffu;
    if (!mask || mask[t]) {
        const int cx = knn_cell1(pos[3 * t]), cy = knn_cell1(pos[3 * t + 1]), cz = knn_cell1(pos[3 * t + 2]);
        k = (unsigned)((cz * KNN_G + cy) * KNN_G + cx);
    }
    keys[t] = k; vals[t] = (int)t;
}
__global__ __launch_bounds__(256) void knn_bounds_kernel(const unsigned* keys, long N, int* cell_start, int* cell_end, int* count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned k = keys[i];
    if (k == 0xffffffffu) return;
    if (i == 0 || keys[i - 1] != k) cell_start[k] = (int)i;
    if (i == N - 1 || keys[i + 1] != k) { cell_end[k] = (int)i + 1; }
    if (i == N - 1 || keys[i + 1] == 0xffffffffu) *count = (int)i + 1;     // number of valid source points (they sort first)
}

template <int MODE>
__global__ __launch_bounds__(128) void knn_query_kernel(KnnParams p, const int* vals, const int* cell_start, const int* cell_end, const int* count) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.M) return;
    if (p.dst_mask && !p.dst_mask[t]) return;
    const int n_src = *count;
    const int k = p.k < n_src ? p.k : n_src;        // fewer sources than k: use what exists (the reference would index out of range)
    float bd[KNN_KMAX]; int bi[KNN_KMAX];
    int have = 0;
    const float qx = p.dst_pos[3 * t], qy = p.dst_pos[3 * t + 1], qz = p.dst_pos[3 * t + 2];
    const int cx = knn_cell1(qx), cy = knn_cell1(qy), cz = knn_cell1(qz);
    const float cs = 2.0f / KNN_G;
    for (int r = 0; r < KNN_G && k > 0; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz; if (z < 0 || z >= KNN_G) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy; if (y < 0 || y >= KNN_G) continue;
                const bool shell_zy = (dz == -r || dz == r || dy == -r || dy == r);
                const int step = (shell_zy || r == 0) ? 1 : 2 * r;
                for (int dx = -r; dx <= r; dx += step) {
                    const int x = cx + dx; if (x < 0 || x >= KNN_G) continue;
                    const int cell = (z * KNN_G + y) * KNN_G + x;
                    const int s = cell_start[cell];
                    if (s < 0) continue;
                    const int e = cell_end[cell];
                    for (int i = s; i < e; ++i) {
                        const int j = vals[i];
                        const float ddx = p.src_pos[3 * (long)j] - qx, ddy = p.src_pos[3 * (long)j + 1] - qy, ddz = p.src_pos[3 * (long)j + 2] - qz;
                        const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        if (have == k && !(d2 < bd[k - 1] || (d2 == bd[k - 1] && j < bi[k - 1]))) continue;
                        int q = have < k ? have : k - 1;            // insertion into the sorted list
                        while (q > 0 && (d2 < bd[q - 1] || (d2 == bd[q - 1] && j < bi[q - 1]))) { bd[q] = bd[q - 1]; bi[q] = bi[q - 1]; --q; }
                        bd[q] = d2; bi[q] = j;
                        if (have < k) ++have;
                    }
                }
            }
        }
        const float rr = (float)r * cs;
        if (have == k && bd[k - 1] < rr * rr * 0.99999f) break;
    }
    if (p.out_idx) for (int q = 0; q < p.k; ++q) p.out_idx[t * p.k + q] = q < have ? bi[q] : -1;
    if (p.out_d2) for (int q = 0; q < p.k; ++q) p.out_d2[t * p.k + q] = q < have ? bd[q] : INFINITY;
    if (!p.out_attr || have == 0) return;
    const int C = p.C;
    if (MODE == 0) {
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int q = 0; q < have; ++q) s += p.src_attr[(long)bi[q] * C + c];
            p.out_attr[t * C + c] = s / (float)have;
        }
    } else {
        // w_q = (1/score_q) / sum(1/score) * cos(n_src_q, n_dst); out = sum(w c) / sum(w); non-finite results -> 0
        float inv[KNN_KMAX], wsum_inv = 0.f;
        for (int q = 0; q < have; ++q) { const float v = 1.0f / bd[q]; inv[q] = (v != v) ? 0.f : v; wsum_inv += fabsf(inv[q]); }
        const float nx = p.dst_nrm[3 * t], ny = p.dst_nrm[3 * t + 1], nz = p.dst_nrm[3 * t + 2];
        const float nd = fmaxf(sqrtf((nx * nx + ny * ny) + nz * nz), 1e-8f);
        float w[KNN_KMAX], wsum = 0.f;
        for (int q = 0; q < have; ++q) {
            const float* n = p.src_nrm + 3 * (long)bi[q];
            const float ns = fmaxf(sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]), 1e-8f);
            const float cs_ = ((n[0] * nx + n[1] * ny) + n[2] * nz) / (ns * nd);
            w[q] = inv[q] / fmaxf(wsum_inv, 1e-12f) * cs_;
            wsum += w[q];
        }
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int q = 0; q < have; ++q) s += p.src_attr[(long)bi[q] * C + c] * w[q];
            float o = s / wsum;
            if (!(fabsf(o) <= 3.0e38f)) o = 0.f;
            p.out_attr[t * C + c] = o;
        }
    }
}

extern "C" size_t utx_knn_workspace_bytes_impl(long N) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)N, 0, 32, (hipStream_t)0);
    return (size_t)N * 16 + (size_t)KNN_G * KNN_G * KNN_G * 8 + 256 + tmp + 256;
}

extern "C" int utx_launch_knn(const KnnParams* pp, void* work, size_t work_bytes, hipStream_t stream) {
    KnnParams p = *pp;
    if (p.N <= 0 || p.M <= 0 || p.k < 1 || p.k > KNN_KMAX || p.C < 0) return -2;
    if (p.mode != 0 && p.mode != 1) return -2;
    if (p.out_attr && (!p.src_attr || p.C < 1)) return -2;
    if (p.mode == 1 && p.out_attr && (!p.src_nrm || !p.dst_nrm)) return -2;
    if (work_bytes < utx_knn_workspace_bytes_impl(p.N)) return -2;
    const long N = p.N;
    unsigned* keys = (unsigned*)work; unsigned* keys_s = keys + N;
    int* vals = (int*)(keys_s + N); int* vals_s = vals + N;
    int* cell_start = vals_s + N; int* cell_end = cell_start + KNN_G * KNN_G * KNN_G;
    int* count = cell_end + KNN_G * KNN_G * KNN_G;
    void* tmp = (void*)(((uintptr_t)(count + 64) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_s, vals, vals_s, (size_t)N, 0, 32, stream) != hipSuccess) return -7;
    const unsigned nb = (unsigned)((N + 255) / 256);
    if (hipMemsetAsync(cell_start, 0xff, (size_t)KNN_G * KNN_G * KNN_G * 4, stream) != hipSuccess) return -7;
    if (hipMemsetAsync(count, 0, 4, stream) != hipSuccess) return -7;
    hipLaunchKernelGGL(knn_keys_kernel, dim3(nb), dim3(256), 0, stream, p.src_pos, p.src_mask, N, keys, vals);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, (size_t)N, 0, 32, stream) != hipSuccess) return -7;
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(nb), dim3(256), 0, stream, keys_s, N, cell_start, cell_end, count);
    const unsigned nq = (unsigned)((p.M + 127) / 128);
    if (p.mode == 0) hipLaunchKernelGGL(knn_query_kernel<0>, dim3(nq), dim3(128), 0, stream, p, vals_s, cell_start, cell_end, count);
    else hipLaunchKernelGGL(knn_query_kernel<1>, dim3(nq), dim3(128), 0, stream, p, vals_s, cell_start, cell_end, count);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
