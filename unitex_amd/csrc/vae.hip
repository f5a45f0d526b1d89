// FLUX AutoencoderKL pieces that are not GEMM-shaped (gfx950): GroupNorm(32)+SiLU, row softmax of the single-head
// mid-block attention, and the two thin-input 3x3 convolutions (image -> 128 ch, latent -> 512 ch).
// Every other convolution of the VAE runs as an implicit GEMM in gemm.hip (CONV mode), the attention projections /
// score / PV products as plain GEMMs.  Activations are NHWC bf16: [H*W, C], pixel-major.
//
// Replaces (diffusers [3p], called at /root/reference/flux_piplines/texturing/pipeline.py:226-238 and :683-692):
// torch.nn.GroupNorm(32, C, eps=1e-6) + SiLU, F.scaled_dot_product_attention of the VAE mid block, Conv2d(3|16, C, 3).
// All three are HBM-bound streaming kernels: 16-byte vector I/O, fp32 statistics, deterministic reduction order.
#include "common.h"
#include "kernels.h"

#define GN_GROUPS 32
#define GN_MAXBLK 1024

// ---- GroupNorm pass 1: per-block partial (sum, sumsq) for each of the 32 groups.
// thread -> (pixel, channel octet); an octet's first / last 4 channels may belong to different groups (C = 128).
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, long npix, int C, float* __restrict__ partial) {
    __shared__ float part[256][4];
    const int oct_per_pix = C >> 3;
    const int cpg = C / GN_GROUPS;
    const int pix_per_it = 256 / oct_per_pix;
    const int o = threadIdx.x % oct_per_pix, po = threadIdx.x / oct_per_pix;
    const long slab = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * slab;
    const long p1 = (p0 + slab < npix) ? p0 + slab : npix;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    for (long p = p0 + po; p < p1; p += pix_per_it) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + p * C + o * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float a = bf2f((uint16_t)(w[c] & 0xffff)), b = bf2f((uint16_t)(w[c] >> 16));
            s0 += a + b; q0 += a * a + b * b;
        }
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            const float a = bf2f((uint16_t)(w[c] & 0xffff)), b = bf2f((uint16_t)(w[c] >> 16));
            s1 += a + b; q1 += a * a + b * b;
        }
    }
    part[threadIdx.x][0] = s0; part[threadIdx.x][1] = q0; part[threadIdx.x][2] = s1; part[threadIdx.x][3] = q1;
    __syncthreads();
    if (threadIdx.x < GN_GROUPS) {
        const int g = threadIdx.x;
        float s = 0.f, q = 0.f;
        for (int t = 0; t < 256; ++t) {   // fixed order -> deterministic
            const int to = t % oct_per_pix;
            if ((to * 8) / cpg == g) { s += part[t][0]; q += part[t][1]; }
            if ((to * 8 + 4) / cpg == g) { s += part[t][2]; q += part[t][3]; }
        }
        partial[((long)blockIdx.x * GN_GROUPS + g) * 2 + 0] = s;
        partial[((long)blockIdx.x * GN_GROUPS + g) * 2 + 1] = q;
    }
}

// ---- GroupNorm pass 2: finish the statistics (every block, same fixed order), normalise, affine, optional SiLU.
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, long npix, int C, const float* __restrict__ partial,
                                                       int nblk_stats, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       float eps, int silu, bf16_t* __restrict__ y) {
    __shared__ float mean_s[GN_GROUPS], rstd_s[GN_GROUPS];
    const int cpg = C / GN_GROUPS;
    if (threadIdx.x < GN_GROUPS) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nblk_stats; ++b) {
            s += (double)partial[((long)b * GN_GROUPS + threadIdx.x) * 2 + 0];
            q += (double)partial[((long)b * GN_GROUPS + threadIdx.x) * 2 + 1];
        }
        const double n = (double)npix * cpg;
        const double m = s / n;
        double var = q / n - m * m;
        if (var < 0.0) var = 0.0;
        mean_s[threadIdx.x] = (float)m;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int oct_per_pix = C >> 3;
    const long total = npix * oct_per_pix;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % oct_per_pix);
        const uint4 v = *reinterpret_cast<const uint4*>(x + i * 8);
        const uint4 gv = *reinterpret_cast<const uint4*>(gamma + o * 8);
        const uint4 bv = *reinterpret_cast<const uint4*>(beta + o * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
        uint32_t ow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int g = (o * 8 + 2 * c) / cpg;   // both halves of a dword share the group (cpg >= 4, even offset)
            const float m = mean_s[g], r = rstd_s[g];
            float a = (bf2f((uint16_t)(w[c] & 0xffff)) - m) * r * bf2f((uint16_t)(gw[c] & 0xffff)) + bf2f((uint16_t)(bw[c] & 0xffff));
            float b = (bf2f((uint16_t)(w[c] >> 16)) - m) * r * bf2f((uint16_t)(gw[c] >> 16)) + bf2f((uint16_t)(bw[c] >> 16));
            if (silu) {   // GroupNorm output is a bf16 tensor in the reference; SiLU is applied to that
                a = rbf(a); b = rbf(b);
                a = a / (1.0f + __expf(-a)); b = b / (1.0f + __expf(-b));
            }
            ow[c] = pack2bf(a, b);
        }
        *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ---- row softmax, bf16 in place: P[i, :] = softmax(S[i, :]) (the 1/sqrt(C) scale was applied by the score GEMM).
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, long ld, int ncol) {
    __shared__ float red[4];
    bf16_t* row = s + (long)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int c = threadIdx.x * 8; c < ncol; c += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, fmaxf(bf2f((uint16_t)(w[k] & 0xffff)), bf2f((uint16_t)(w[k] >> 16))));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x * 8; c < ncol; c += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            sum += __expf(bf2f((uint16_t)(w[k] & 0xffff)) - mx) + __expf(bf2f((uint16_t)(w[k] >> 16)) - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = threadIdx.x * 8; c < ncol; c += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            ow[k] = pack2bf(__expf(bf2f((uint16_t)(w[k] & 0xffff)) - mx) * inv, __expf(bf2f((uint16_t)(w[k] >> 16)) - mx) * inv);
        *reinterpret_cast<uint4*>(row + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ---- 3x3, stride 1, pad 1 convolution for thin inputs (Cin = 3 or 16): y[p, co] = b[co] + sum_{tap, ci} x[p+tap, ci] w[tap, ci, co]
// x NHWC [H*W, Cin] bf16, wt [9*Cin][Cout] bf16 (tap-major, co contiguous), thread = (pixel, 8 output channels).
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_thin_kernel(const bf16_t* __restrict__ x, int H, int W, const bf16_t* __restrict__ wt,
                                                           const bf16_t* __restrict__ bias, int Cout, bf16_t* __restrict__ y) {
    const int oct = Cout >> 3;
    const long total = (long)H * W * oct;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % oct);
        const long p = i / oct;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        float acc[8];
        {
            const uint4 bv = *reinterpret_cast<const uint4*>(bias + o * 8);
            const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[2 * c] = bf2f((uint16_t)(bw[c] & 0xffff)); acc[2 * c + 1] = bf2f((uint16_t)(bw[c] >> 16)); }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const bf16_t* xp = x + ((long)iy * W + ix) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float xv = bf2f(xp[ci]);
                const uint4 wv = *reinterpret_cast<const uint4*>(wt + (long)(tap * CIN + ci) * Cout + o * 8);
                const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[2 * c] += xv * bf2f((uint16_t)(ww[c] & 0xffff));
                    acc[2 * c + 1] += xv * bf2f((uint16_t)(ww[c] >> 16));
                }
            }
        }
        *reinterpret_cast<uint4*>(y + p * Cout + o * 8) =
            make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
    }
}

static int grid_for(long work_items) {
    long g = (work_items + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t utx_group_norm_workspace_bytes_impl(void) { return (size_t)GN_MAXBLK * GN_GROUPS * 2 * sizeof(float); }

extern "C" int utx_launch_group_norm(const void* x, long npix, int C, const void* gamma, const void* beta, float eps, int silu,
                                     void* y, void* work, hipStream_t stream) {
    if (npix <= 0 || C < 128 || (C % 128) || C > 2048) return -2;   // cpg in {4, 8, 16, ...}: octet halves never straddle 3 groups
    long nb = (npix * (C >> 3) + 256 * 64 - 1) / (256 * 64);
    if (nb > GN_MAXBLK) nb = GN_MAXBLK;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((int)nb), dim3(256), 0, stream, (const bf16_t*)x, npix, C, (float*)work);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(npix * (C >> 3))), dim3(256), 0, stream, (const bf16_t*)x, npix, C,
                       (const float*)work, (int)nb, (const bf16_t*)gamma, (const bf16_t*)beta, eps, silu, (bf16_t*)y);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int utx_launch_softmax_rows(void* s, long nrow, long ld, int ncol, hipStream_t stream) {
    if (nrow <= 0 || ncol <= 0 || (ncol & 7) || (ld & 7)) return -2;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)nrow), dim3(256), 0, stream, (bf16_t*)s, ld, ncol);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int utx_launch_conv3x3_thin(const void* x, int H, int W, int Cin, const void* wt, const void* bias, int Cout, void* y,
                                       hipStream_t stream) {
    if (H <= 0 || W <= 0 || (Cout & 7)) return -2;
    const int g = grid_for((long)H * W * (Cout >> 3));
    if (Cin == 3)
        hipLaunchKernelGGL((conv3x3_thin_kernel<3>), dim3(g), dim3(256), 0, stream, (const bf16_t*)x, H, W, (const bf16_t*)wt, (const bf16_t*)bias, Cout, (bf16_t*)y);
    else if (Cin == 16)
        hipLaunchKernelGGL((conv3x3_thin_kernel<16>), dim3(g), dim3(256), 0, stream, (const bf16_t*)x, H, W, (const bf16_t*)wt, (const bf16_t*)bias, Cout, (bf16_t*)y);
    else
        return -2;
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
