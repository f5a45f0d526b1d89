// Fused UV back-projection (gfx950): per (view, texel) colour gather + ray visibility, the visibility
// hole-filling, and the priority composite.
//
// Replaces NVDiffRendererInverse.uv_to_pcd (TextureTools/texturetools/render/nvdiffrast/
// renderer_inverse.py:262-343) and the composite loop of bake_mv_to_uv_reproject_blur (:591-602).
// The reference materialises ~10 tensors of shape [6,2048,2048,3] f32 (302 MB each) on the way
// (rays_o, rays_d, ndc, sampled image, masked_select copies ...); here one pass reads the 16-byte UV
// raster record per texel, recomputes position / ray / NDC in registers, bilinearly samples the view
// (4 taps x float4, L2/MALL resident: 6 x 512^2 x 16 B = 25 MB), walks the LBVH and writes 12 B colour
// + 2 flag bytes per (view, texel).  HBM-bound gather: no LDS, coalesced texel-major reads/writes.
// Same float expressions, in the same order, as oracle/geom_ref.c::utxref_backproject (-ffp-contract=off).
#include "common.h"
#include "kernels.h"
#include "bvh_device.h"

__device__ __forceinline__ float4 tap4(const float4* img, int H, int W, int x, int y) {
    if (x < 0 || x >= W || y < 0 || y >= H) return make_float4(0.f, 0.f, 0.f, 0.f);
    return img[(long)y * W + x];
}

// MODE 1: stackless thread-per-ray walk over the packed tree (bvh_device.h); MODE 0: the reference's 64-entry stack walk, kept for trees deeper than
// UTX_BVH_PACKED_MAX_DEPTH (where the reference's stack overflow quirk could matter) and for A/B tests; MODE 2 (round 4, the default): a wave owns an
// 8 x 8 TEXEL TILE of one view and its 64 parallel rays walk the packed tree as ONE PACKET (bvh_trace_packet: nodes through the scalar cache, per-lane
// box / triangle tests, bit-identical results); 256 threads = a 16 x 16 block of texels, tiles without a covered texel skip the walk.
template <int MODE>
__global__ __launch_bounds__(256) void backproject_kernel(utx_backproject_desc p, const int* info, const float* aabb, const float4* nodes, const float4* tris) {
    __shared__ int pstack[MODE == 2 ? 4 * 192 : 1];      // MODE 2: the packets' DFS stacks, 64 x {node, mask lo, mask hi} per wave
    const long T = (long)p.T_h * p.T_w;
    long t;
    bool inside = true;
    if constexpr (MODE == 2) {
        const int tiles_x = (p.T_w + 15) / 16;
        const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const int x = bx * 16 + (wv & 1) * 8 + (ln & 7), y = by * 16 + (wv >> 1) * 8 + (ln >> 3);
        inside = x < p.T_w && y < p.T_h;
        t = inside ? (long)y * p.T_w + x : 0;
    } else {
        t = (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= T) return;
    }
    const int vw = p.view_begin + blockIdx.y;
    const float4 r = inside ? ((const float4*)p.rast2d)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int id = (int)r.w - 1;
    float* oc = (float*)p.color + ((long)vw * T + t) * 3;
    unsigned char* rv = (unsigned char*)p.rayvis + (long)vw * T + t;
    unsigned char* ao = (unsigned char*)p.alphaok + (long)vw * T + t;
    if constexpr (MODE == 2) {
        if (__ballot(id >= 0) == 0) {      // nothing of this tile is covered
            if (inside) { oc[0] = 0.f; oc[1] = 0.f; oc[2] = 0.f; *rv = 0; *ao = 0; }
            return;
        }
        if (id < 0) {      // an uncovered texel of a covered tile stays in the wave (the packet walk is wave-wide) with its ray switched off
            if (inside) { oc[0] = 0.f; oc[1] = 0.f; oc[2] = 0.f; *rv = 0; *ao = 0; }
            return;      // (the packet walk below ballots over the lanes that are still here)
        }
    } else if (id < 0) { oc[0] = 0.f; oc[1] = 0.f; oc[2] = 0.f; *rv = 0; *ao = 0; return; }
    const float* vert = (const float*)p.verts;
    const int* faces = (const int*)p.faces;
    const float u = r.x, v = r.y, w = (1.0f - u) - v;
    const int f0 = faces[3 * (long)id], f1 = faces[3 * (long)id + 1], f2 = faces[3 * (long)id + 2];
    float pos[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pos[a] = (vert[3 * (long)f0 + a] * u + vert[3 * (long)f1 + a] * v) + vert[3 * (long)f2 + a] * w;
    const float* d_in = (const float*)p.dirs + 3 * vw;
    const float two_sqrt3 = p.two_sqrt3;
    const float ro[3] = {pos[0] - two_sqrt3 * d_in[0], pos[1] - two_sqrt3 * d_in[1], pos[2] - two_sqrt3 * d_in[2]};
    float dn = sqrtf(dot3(d_in, d_in)); if (dn < 1e-12f) dn = 1e-12f;
    const float d[3] = {d_in[0] / dn, d_in[1] / dn, d_in[2] / dn};
    const float* n = (const float*)p.fnormal + 3 * (long)id;
    float ld = sqrtf(dot3(d, d)); if (ld < 1e-8f) ld = 1e-8f;
    const float nn[3] = {n[0], n[1], n[2]};
    float ln = sqrtf(dot3(nn, nn)); if (ln < 1e-8f) ln = 1e-8f;
    const float cs = dot3(d, nn) / (ld * ln);
    const float* nd = (const float*)p.vndc + (long)vw * p.V * 2;
    const float gx = (nd[2 * (long)f0] * u + nd[2 * (long)f1] * v) + nd[2 * (long)f2] * w;
    const float gy = (nd[2 * (long)f0 + 1] * u + nd[2 * (long)f1 + 1] * v) + nd[2 * (long)f2 + 1] * w;
    const int H = p.H, W = p.W;
    const float ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty), w10 = (1.0f - tx) * ty, w11 = tx * ty;
    const float4* img = (const float4*)p.images + (long)vw * H * W;
    const float4 a = tap4(img, H, W, x0, y0), b = tap4(img, H, W, x0 + 1, y0);
    const float4 c = tap4(img, H, W, x0, y0 + 1), e = tap4(img, H, W, x0 + 1, y0 + 1);
    oc[0] = ((a.x * w00 + b.x * w01) + c.x * w10) + e.x * w11;
    oc[1] = ((a.y * w00 + b.y * w01) + c.y * w10) + e.y * w11;
    oc[2] = ((a.z * w00 + b.z * w01) + c.z * w10) + e.z * w11;
    const float sa = ((a.w * w00 + b.w * w01) + c.w * w10) + e.w * w11;
    *ao = sa > 0.999f ? 1 : 0;
    const int hit = MODE == 2 ? bvh_trace_packet(nodes, tris, ro, d, true, pstack + (threadIdx.x >> 6) * 192, nullptr)
                  : MODE == 1 ? bvh_trace_packed(nodes, tris, ro, d, nullptr) : bvh_trace_one(info, aabb, vert, faces, ro, d);
    *rv = (hit == id && hit != -1 && cs < p.cos_thresh) ? 1 : 0;
}

extern "C" int utx_launch_backproject(const utx_backproject_desc* hp, const utx_bvh* bvh, hipStream_t stream) {
    utx_backproject_desc p = *hp;
    if (!bvh || p.T_h <= 0 || p.T_w <= 0 || p.view_count <= 0) return -2;
    const long T = (long)p.T_h * p.T_w;
    dim3 grid((unsigned)((T + 255) / 256), p.view_count);
    const int depth = utx_bvh_depth_impl(const_cast<utx_bvh*>(bvh));      // first use after a build: waits for the build's depth word
    if (depth < 0) return -7;
    if (depth <= UTX_BVH_PACKED_MAX_DEPTH && !g_utx_opt.bvh_stack_walk && g_utx_opt.bvh_packet) {
        dim3 gridp((unsigned)(((p.T_w + 15) / 16) * ((p.T_h + 15) / 16)), p.view_count);
        hipLaunchKernelGGL(backproject_kernel<2>, gridp, dim3(256), 0, stream, p, bvh->info, bvh->aabb, bvh->nodes, bvh->tris);
    } else if (depth <= UTX_BVH_PACKED_MAX_DEPTH && !g_utx_opt.bvh_stack_walk)
        hipLaunchKernelGGL(backproject_kernel<1>, grid, dim3(256), 0, stream, p, bvh->info, bvh->aabb, bvh->nodes, bvh->tris);
    else
        hipLaunchKernelGGL(backproject_kernel<0>, grid, dim3(256), 0, stream, p, bvh->info, bvh->aabb, bvh->nodes, bvh->tris);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// visibility hole filling (renderer_inverse.py:327-343, kernel_mode 7 => k = 3 then k = 5; A13):
//   k=3: m |= (9 * #set 8-neighbours - m) >= 3          k=5: m |= (25 * #set rim(16) - #set core(9)) >= 135
// zero padding outside the atlas.  pass 0: k=3 (src -> dst); pass 1: k=5 + AND coverage + AND alpha.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dilate_kernel(const unsigned char* src, unsigned char* dst, int n_views, int Hh, int Ww,
                                                     int pass, const float4* rast2d, const unsigned char* alphaok) {
    const long T = (long)Hh * Ww;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int vw = blockIdx.y;
    if (t >= T) return;
    const int y = (int)(t / Ww), x = (int)(t % Ww);
    const unsigned char* m = src + (long)vw * T;
    auto at = [&](int yy, int xx) -> int { return (yy < 0 || yy >= Hh || xx < 0 || xx >= Ww) ? 0 : (int)m[(long)yy * Ww + xx]; };
    int self = at(y, x), out;
    if (pass == 0) {
        int s = 0;
        for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if (dy || dx) s += at(y + dy, x + dx);
        out = (self || (9 * s - self) >= 3) ? 1 : 0;
    } else {
        int rim = 0, core = 0;
        for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) {
            const int vv = at(y + dy, x + dx);
            if (dy == -2 || dy == 2 || dx == -2 || dx == 2) rim += vv; else core += vv;
        }
        out = (self || (25 * rim - core) >= 135) ? 1 : 0;
        const bool cov = rast2d[t].w > 0.f;
        out = (out && cov && alphaok[(long)vw * T + t]) ? 1 : 0;
    }
    dst[(long)vw * T + t] = (unsigned char)out;
}

extern "C" int utx_launch_dilate_visibility(const void* rayvis, const void* alphaok, const void* rast2d, int n_views, int Hh, int Ww,
                                            void* tmp, void* vis_out, hipStream_t stream) {
    if (n_views <= 0 || Hh <= 0 || Ww <= 0) return -2;
    const long T = (long)Hh * Ww;
    dim3 grid((unsigned)((T + 255) / 256), n_views);
    hipLaunchKernelGGL(dilate_kernel, grid, dim3(256), 0, stream, (const unsigned char*)rayvis, (unsigned char*)tmp, n_views, Hh, Ww, 0,
                       (const float4*)rast2d, (const unsigned char*)alphaok);
    hipLaunchKernelGGL(dilate_kernel, grid, dim3(256), 0, stream, (const unsigned char*)tmp, (unsigned char*)vis_out, n_views, Hh, Ww, 1,
                       (const float4*)rast2d, (const unsigned char*)alphaok);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// priority composite (renderer_inverse.py:595-602): first view in `order` that sees the texel wins.
// A per-texel scan over <= 8 views -- no atomics needed.  winner = -1 where no view sees the texel.
// ---------------------------------------------------------------------------------------------
struct OrderArg { int v[8]; int n; };

__global__ __launch_bounds__(256) void composite_kernel(const float* colors, const unsigned char* vis, OrderArg ord, long T,
                                                        float* atlas, signed char* winner) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int wv = -1;
    for (int i = 0; i < ord.n; ++i) { const int vw = ord.v[i]; if (vis[(long)vw * T + t]) { wv = vw; break; } }
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (wv >= 0) { const float* c = colors + ((long)wv * T + t) * 3; c0 = c[0]; c1 = c[1]; c2 = c[2]; }
    atlas[3 * t] = c0; atlas[3 * t + 1] = c1; atlas[3 * t + 2] = c2;
    winner[t] = (signed char)wv;
}

extern "C" int utx_launch_composite(const float* colors, const void* vis, const int* order, int n_order, long T, float* atlas,
                                    void* winner, hipStream_t stream) {
    if (n_order <= 0 || n_order > 8 || T <= 0) return -2;
    OrderArg o; o.n = n_order;
    for (int i = 0; i < 8; ++i) o.v[i] = i < n_order ? order[i] : 0;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, stream, colors, (const unsigned char*)vis, o, T,
                       atlas, (signed char*)winner);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
