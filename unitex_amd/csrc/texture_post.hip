// Atlas post-processing of bake_mv_to_uv_reproject_blur (gfx950):
//   seam mask, exact 3-D nearest-neighbour fill of unseen texels, lens blur on the seam, pull-push.
// Reference: TextureTools/texturetools/render/nvdiffrast/renderer_inverse.py:603-627,
// image/lens_blur.py:260-280, texture/stitching/mip.py:9-95, pcd/knn/__init__.py:103-113 (torch_kdtree [3p]).
// All HBM-bound; -ffp-contract=off so the float sequences match oracle/geom_ref.py where it is bit-exact.
#include "common.h"
#include "kernels.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

// ---------------------------------------------------------------------------------------------
// seam mask.  The reference ORs, over the views in priority order, the inner+outer 3x3 boundary of the
// region newly claimed by each view (renderer_inverse.py:602 + get_boundary_mask :435-444).  With
// winner[p] = claiming view (or -1) this is exactly: bnd(p) <=> some in-image 8-neighbour has a different
// winner.  seam = maxpool3(bnd) AND (7x7 erosion of the coverage mask)  (:603-604).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seam_bnd_kernel(const signed char* winner, int Hh, int Ww, unsigned char* bnd) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Hh * Ww) return;
    const int y = (int)(t / Ww), x = (int)(t % Ww);
    const signed char w0 = winner[t];
    int diff = 0;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= Hh || xx < 0 || xx >= Ww) continue;
        diff |= (winner[(long)yy * Ww + xx] != w0);
    }
    bnd[t] = (unsigned char)diff;
}
__global__ __launch_bounds__(256) void seam_final_kernel(const unsigned char* bnd, const float4* rast2d, int Hh, int Ww, unsigned char* seam) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Hh * Ww) return;
    const int y = (int)(t / Ww), x = (int)(t % Ww);
    int any = 0, all = 1;
    for (int dy = -3; dy <= 3; ++dy) for (int dx = -3; dx <= 3; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= Hh || xx < 0 || xx >= Ww) continue;
        const long q = (long)yy * Ww + xx;
        if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) any |= bnd[q];
        all &= (rast2d[q].w > 0.f);
    }
    seam[t] = (unsigned char)(any && all);
}
extern "C" int utx_launch_seam_mask(const void* winner, const float* rast2d, int Hh, int Ww, void* tmp, void* seam, hipStream_t stream) {
    const long T = (long)Hh * Ww;
    if (T <= 0) return -2;
    const unsigned nb = (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(seam_bnd_kernel, dim3(nb), dim3(256), 0, stream, (const signed char*)winner, Hh, Ww, (unsigned char*)tmp);
    hipLaunchKernelGGL(seam_final_kernel, dim3(nb), dim3(256), 0, stream, (const unsigned char*)tmp, (const float4*)rast2d, Hh, Ww, (unsigned char*)seam);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// exact 1-NN fill in 3-D (renderer_inverse.py:606-615).  Uniform grid (G^3 cells over [-1,1]^3):
// seen texels are sorted by cell id (rocPRIM radix sort), unseen covered texels search outward ring by
// ring; ring r+1 can only hold points at distance >= r*cell, which gives an exact stopping rule.
// d2 = (dx*dx + dy*dy) + dz*dz in float32; ties -> lowest texel index (= lowest index in the reference's
// row-major compaction).
// ---------------------------------------------------------------------------------------------
// Round 4: the time of this stage was the QUERY kernel (5.4 of 5.5 ms at 50 k faces / 2048^2 -- rocprofv3, profiles/r04_rocprofv3_kernel_stats_backprojection.csv),
// not the sort: with 128^3 cells a surface cell holds ~100 seen texels and every query scans 27 cells of them through an index indirection (a
// dependent, scattered 12-byte load per candidate).  Now (a) 256^3 cells (a quarter of the candidates per ring volume on a surface), (b) the seen texels'
// positions are GATHERED into cell order once ({x, y, z, texel index} as one 16-byte record), so a query streams each cell's candidates from
// consecutive addresses.  Same exact search, same stopping rule, same tie rule: the result does not depend on the grid.
// Round 5: the resolution follows the atlas: the cell table is 8 G^3 bytes (134 MB / 17 MB / 2 MB) and is cleared on every call, which a small atlas should not pay for.
// Round 6: the steps sit where the candidate count per searched cell volume stays level -- 256 from 512^2 texels (a 1000^2 atlas, just under 2^20, used to drop to 128
// and scan 8x the candidates per ring), 128 from 128^2, 64 below.  The result does not depend on the grid: UTX_NN_GRID (64 | 128 | 256) forces one, and
// tests/test_geometry_gpu.py compares nn_index across all three on one input.  Positions outside [-1, 1] are clamped into edge cells; the stopping rule survives that: a
// clamped point is at least as far from any query as its cell's inner face, and a clamped query only loses rings that hold no cells.
static inline int nn_grid(long T) {
    const int f = g_utx_opt.nn_grid;
    if (f == 64 || f == 128 || f == 256) return f;
    return T >= (1L << 18) ? 256 : (T >= (1L << 14) ? 128 : 64);
}
__device__ __forceinline__ int nn_cell1(float v, int G) {
    int c = (int)floorf((v + 1.0f) * ((float)G * 0.5f));
    return c < 0 ? 0 : (c > G - 1 ? G - 1 : c);
}
__global__ __launch_bounds__(256) void nn_keys_kernel(const float* pos, const signed char* winner, long T, unsigned* keys, int* vals, int NN_G) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    unsigned k = 0xffffffffu;
    if (winner[t] >= 0) {
        const int cx = nn_cell1(pos[3 * t], NN_G), cy = nn_cell1(pos[3 * t + 1], NN_G), cz = nn_cell1(pos[3 * t + 2], NN_G);
        k = (unsigned)((cz * NN_G + cy) * NN_G + cx);
    }
    keys[t] = k; vals[t] = (int)t;
}
// cell bounds + the candidates' records in cell order: spos[i] = {pos[vals[i]], vals[i]}
__global__ __launch_bounds__(256) void nn_bounds_kernel(const unsigned* keys, const int* vals, const float* pos, long T, int* cell_start, int* cell_end, float4* spos) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const unsigned k = keys[i];
    if (k == 0xffffffffu) return;
    if (i == 0 || keys[i - 1] != k) cell_start[k] = (int)i;
    if (i == T - 1 || keys[i + 1] != k) cell_end[k] = (int)i + 1;
    const int j = vals[i];
    spos[i] = make_float4(pos[3 * (long)j], pos[3 * (long)j + 1], pos[3 * (long)j + 2], __int_as_float(j));
}
__global__ __launch_bounds__(256) void nn_query_kernel(const float* pos, const signed char* winner, const float4* rast2d, long T,
                                                       const float4* __restrict__ spos, const int* __restrict__ cell_start, const int* __restrict__ cell_end,
                                                       float* atlas, int* nn_index, int NN_G) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (nn_index) nn_index[t] = -1;
    if (winner[t] >= 0 || !(rast2d[t].w > 0.f)) return;
    const float qx = pos[3 * t], qy = pos[3 * t + 1], qz = pos[3 * t + 2];
    const int cx = nn_cell1(qx, NN_G), cy = nn_cell1(qy, NN_G), cz = nn_cell1(qz, NN_G);
    const float cs = 2.0f / (float)NN_G;
    float best = 3.0e38f; int bi = -1;
    for (int r = 0; r < NN_G; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz; if (z < 0 || z >= NN_G) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy; if (y < 0 || y >= NN_G) continue;
                const bool shell_zy = (dz == -r || dz == r || dy == -r || dy == r);
                if (shell_zy || r == 0) {
                    // a whole x-row of the shell: its cells are consecutive cell ids, and consecutive NON-EMPTY cells are consecutive ranges of spos --
                    // one pass over [start of the first non-empty cell, end of the last one)
                    int x0 = cx - r, x1 = cx + r;
                    if (x0 < 0) x0 = 0;
                    if (x1 > NN_G - 1) x1 = NN_G - 1;
                    const int base = (z * NN_G + y) * NN_G;
                    int s = -1, e = -1;
                    for (int x = x0; x <= x1; ++x) {
                        const int cs_ = cell_start[base + x];
                        if (cs_ < 0) continue;
                        if (s < 0) s = cs_;
                        e = cell_end[base + x];
                    }
                    for (int i = s; i < e; ++i) {
                        const float4 c = spos[i];
                        const int j = __float_as_int(c.w);
                        const float ddx = c.x - qx, ddy = c.y - qy, ddz = c.z - qz;
                        const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        if (d2 < best || (d2 == best && j < bi)) { best = d2; bi = j; }
                    }
                } else {
                    for (int dx = -r; dx <= r; dx += 2 * r) {
                        const int x = cx + dx; if (x < 0 || x >= NN_G) continue;
                        const int cell = (z * NN_G + y) * NN_G + x;
                        const int s = cell_start[cell];
                        if (s < 0) continue;
                        const int e = cell_end[cell];
                        for (int i = s; i < e; ++i) {
                            const float4 c = spos[i];
                            const int j = __float_as_int(c.w);
                            const float ddx = c.x - qx, ddy = c.y - qy, ddz = c.z - qz;
                            const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                            if (d2 < best || (d2 == best && j < bi)) { best = d2; bi = j; }
                        }
                    }
                }
            }
        }
        const float rr = (float)r * cs;
        if (bi >= 0 && best < rr * rr * 0.99999f) break;
    }
    if (bi >= 0) { atlas[3 * t] = atlas[3 * (long)bi]; atlas[3 * t + 1] = atlas[3 * (long)bi + 1]; atlas[3 * t + 2] = atlas[3 * (long)bi + 2]; }
    if (nn_index) nn_index[t] = bi;
}

extern "C" size_t utx_nn_fill_workspace_bytes_impl(long T) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)T, 0, 32, (hipStream_t)0);
    const size_t NN_G = (size_t)nn_grid(T);
    return (size_t)T * 16 + (size_t)T * 16 + NN_G * NN_G * NN_G * 8 + tmp + 512;
}

extern "C" int utx_launch_nn_fill(const float* pos, const void* winner, const float* rast2d, long T, float* atlas, int* nn_index,
                                  void* work, size_t work_bytes, hipStream_t stream) {
    if (T <= 0) return -2;
    if (work_bytes < utx_nn_fill_workspace_bytes_impl(T)) return -2;
    const int NN_G = nn_grid(T);
    unsigned* keys = (unsigned*)work; unsigned* keys_s = keys + T;
    int* vals = (int*)(keys_s + T); int* vals_s = vals + T;
    float4* spos = (float4*)(((uintptr_t)(vals_s + T) + 15) & ~(uintptr_t)15);
    int* cell_start = (int*)(spos + T); int* cell_end = cell_start + (size_t)NN_G * NN_G * NN_G;
    void* tmp = (void*)(((uintptr_t)(cell_end + (size_t)NN_G * NN_G * NN_G) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_s, vals, vals_s, (size_t)T, 0, 32, stream) != hipSuccess) return -7;
    const unsigned nb = (unsigned)((T + 255) / 256);
    if (hipMemsetAsync(cell_start, 0xff, (size_t)NN_G * NN_G * NN_G * 4, stream) != hipSuccess) return -7;
    hipLaunchKernelGGL(nn_keys_kernel, dim3(nb), dim3(256), 0, stream, pos, (const signed char*)winner, T, keys, vals, NN_G);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, (size_t)T, 0, 32, stream) != hipSuccess) return -7;
    hipLaunchKernelGGL(nn_bounds_kernel, dim3(nb), dim3(256), 0, stream, keys_s, vals_s, pos, T, cell_start, cell_end, spos);
    hipLaunchKernelGGL(nn_query_kernel, dim3(nb), dim3(256), 0, stream, pos, (const signed char*)winner, (const float4*)rast2d, T, spos,
                       cell_start, cell_end, atlas, nn_index, NN_G);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// lens blur, evaluated only where it is consumed (the seam mask).  lens_blur_torch is linear between the
// x^gamma and ^(1/gamma) maps: the 5 separable complex components collapse to ONE real 7x7 kernel
//   K[dy][dx] = sum_c A_c Re(k_c[dy] k_c[dx]) + B_c Im(k_c[dy] k_c[dx])      (host-computed, float32)
// out = clamp( max(sum K * x^5, 0)^(1/5), 0, 1 ), zero padding, x^5 = ((x*x)*(x*x))*x.
// ---------------------------------------------------------------------------------------------
struct BlurK { float k[49]; };
__global__ __launch_bounds__(256) void lens_blur_seam_kernel(const float* src, const unsigned char* seam, int Hh, int Ww, BlurK K, float* dst) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Hh * Ww) return;
    float o0 = src[3 * t], o1 = src[3 * t + 1], o2 = src[3 * t + 2];
    if (seam[t]) {
        const int y = (int)(t / Ww), x = (int)(t % Ww);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int dy = -3; dy <= 3; ++dy) for (int dx = -3; dx <= 3; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= Hh || xx < 0 || xx >= Ww) continue;
            const float kk = K.k[(dy + 3) * 7 + (dx + 3)];
            const float* c = src + 3 * ((long)yy * Ww + xx);
            const float c0 = c[0], c1 = c[1], c2 = c[2];
            a0 = a0 + kk * (((c0 * c0) * (c0 * c0)) * c0);
            a1 = a1 + kk * (((c1 * c1) * (c1 * c1)) * c1);
            a2 = a2 + kk * (((c2 * c2) * (c2 * c2)) * c2);
        }
        o0 = fminf(fmaxf(powf(fmaxf(a0, 0.f), 0.2f), 0.f), 1.f);
        o1 = fminf(fmaxf(powf(fmaxf(a1, 0.f), 0.2f), 0.f), 1.f);
        o2 = fminf(fmaxf(powf(fmaxf(a2, 0.f), 0.2f), 0.f), 1.f);
    }
    dst[3 * t] = o0; dst[3 * t + 1] = o1; dst[3 * t + 2] = o2;
}
extern "C" int utx_launch_lens_blur_seam(const float* src, const void* seam, int Hh, int Ww, const float* k49_host, float* dst, hipStream_t stream) {
    const long T = (long)Hh * Ww;
    if (T <= 0 || !k49_host) return -2;
    BlurK K; for (int i = 0; i < 49; ++i) K.k[i] = k49_host[i];
    hipLaunchKernelGGL(lens_blur_seam_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, stream, src, (const unsigned char*)seam, Hh, Ww, K, dst);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// pull-push (mip.py:51-95).  Level l has size (H>>l, W>>l); colour interleaved [h][w][3] f32, mask u8.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_pull_kernel(const float* kd, const unsigned char* mask, int Hh, int Ww, int zero_outside,
                                                      float* kd_mip, unsigned char* mask_mip) {
    const int h2 = Hh / 2, w2 = Ww / 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)h2 * w2) return;
    const int y = (int)(t / w2), x = (int)(t % w2);
    float a = 0.f, k[3] = {0.f, 0.f, 0.f};
    for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
        const long q = (long)(2 * y + dy) * Ww + (2 * x + dx);
        const float m = mask[q] ? 1.0f : 0.0f;
        a = a + m;
        for (int c = 0; c < 3; ++c) { const float vv = (zero_outside && !mask[q]) ? 0.f : kd[3 * q + c]; k[c] = k[c] + vv; }
    }
    a = a * 0.25f;
    for (int c = 0; c < 3; ++c) k[c] = k[c] * 0.25f;
    if (a > 0.f && a < 1.f) for (int c = 0; c < 3; ++c) k[c] = k[c] / a;
    for (int c = 0; c < 3; ++c) kd_mip[3 * t + c] = k[c];
    mask_mip[t] = a > 0.f ? 1 : 0;
}
// out(fine) = mask ? kd : bilinear_up(kd_mip)  (replicate padding; tap order = the reference's conv2d kernels)
__global__ __launch_bounds__(256) void pp_push_kernel(const float* kd, const unsigned char* mask, int Hh, int Ww, int zero_outside,
                                                      const float* kd_mip, float* out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Hh * Ww) return;
    const int y = (int)(t / Ww), x = (int)(t % Ww);
    if (mask[t]) { for (int c = 0; c < 3; ++c) out[3 * t + c] = kd[3 * t + c]; return; }
    (void)zero_outside;
    const int h2 = Hh / 2, w2 = Ww / 2;
    const int i = y >> 1, j = x >> 1, py = y & 1, px = x & 1;
    const int iy = py ? min(i + 1, h2 - 1) : max(i - 1, 0);
    const int jx = px ? min(j + 1, w2 - 1) : max(j - 1, 0);
    const float* c_ = kd_mip + 3 * ((long)i * w2 + j);
    const float* cx = kd_mip + 3 * ((long)i * w2 + jx);
    const float* cy = kd_mip + 3 * ((long)iy * w2 + j);
    const float* cxy = kd_mip + 3 * ((long)iy * w2 + jx);
    const float w9 = 0.5625f, w3 = 0.1875f, w1 = 0.0625f;
    for (int c = 0; c < 3; ++c) {
        float r;
        if (!py && !px) r = ((cxy[c] * w1 + cy[c] * w3) + cx[c] * w3) + c_[c] * w9;
        else if (!py && px) r = ((cy[c] * w3 + cxy[c] * w1) + c_[c] * w9) + cx[c] * w3;
        else if (py && !px) r = ((cx[c] * w3 + c_[c] * w9) + cxy[c] * w1) + cy[c] * w3;
        else r = ((c_[c] * w9 + cx[c] * w3) + cy[c] * w3) + cxy[c] * w1;
        out[3 * t + c] = r;
    }
}

extern "C" size_t utx_pull_push_workspace_bytes_impl(int Hh, int Ww) {
    size_t tot = 0;
    int h = Hh, w = Ww;
    for (int l = 0; l < 32 && h >= 2 && w >= 2; ++l) { h /= 2; w /= 2; tot += (size_t)h * w * (2 * 12 + 1) + 64; }
    return tot + 256;
}

extern "C" int utx_launch_pull_push(const float* kd, const void* mask, int Hh, int Ww, float* out, void* work, hipStream_t stream) {
    int n = 0;
    { int lh = 0, lw = 0; while ((1 << (lh + 1)) <= Hh) ++lh; while ((1 << (lw + 1)) <= Ww) ++lw; n = (lh < lw ? lh : lw) - 2; if (n < 0) n = 0; }
    const long T = (long)Hh * Ww;
    if (n == 0) { return hipMemcpyAsync(out, kd, T * 12, hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : -7; }
    if (n > 16) return -2;
    float* kds[17]; float* fill[17]; unsigned char* ms[17]; int hs[17], wsz[17];
    char* p = (char*)work;
    hs[0] = Hh; wsz[0] = Ww; kds[0] = (float*)kd; ms[0] = (unsigned char*)mask; fill[0] = out;
    for (int l = 1; l <= n; ++l) {
        hs[l] = hs[l - 1] / 2; wsz[l] = wsz[l - 1] / 2;
        const size_t px = (size_t)hs[l] * wsz[l];
        kds[l] = (float*)p; p += px * 12;
        fill[l] = (float*)p; p += px * 12;
        ms[l] = (unsigned char*)p; p += (px + 63) & ~(size_t)63;
    }
    for (int l = 1; l <= n; ++l) {
        const long px = (long)hs[l] * wsz[l];
        hipLaunchKernelGGL(pp_pull_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, stream, kds[l - 1], ms[l - 1], hs[l - 1], wsz[l - 1],
                           l == 1 ? 1 : 0, kds[l], ms[l]);
    }
    // coarsest level is its own fill; push down to level 0
    const float* cur = kds[n];
    for (int l = n - 1; l >= 0; --l) {
        const long px = (long)hs[l] * wsz[l];
        hipLaunchKernelGGL(pp_push_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, stream, kds[l], ms[l], hs[l], wsz[l], l == 0 ? 1 : 0, cur, fill[l]);
        cur = fill[l];
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// tensor_to_image (renderer_utils.py:62-83): clamp(0,1)*255 -> uint8 by truncation; optional vertical flip
// (link_rgb_to_mesh flips the atlas when attaching it: io/link_pbr_to_mesh.py:17).
__global__ __launch_bounds__(256) void to_u8_kernel(const float* src, long n_rows, long row_elems, int flip, unsigned char* dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * row_elems) return;
    const long r = i / row_elems, c = i % row_elems;
    const float v = fminf(fmaxf(src[i], 0.f), 1.f) * 255.0f;
    dst[(flip ? (n_rows - 1 - r) : r) * row_elems + c] = (unsigned char)v;
}
extern "C" int utx_launch_to_u8(const float* src, long n_rows, long row_elems, int flip, void* dst, hipStream_t stream) {
    const long n = n_rows * row_elems;
    if (n <= 0) return -2;
    hipLaunchKernelGGL(to_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, n_rows, row_elems, flip, (unsigned char*)dst);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
