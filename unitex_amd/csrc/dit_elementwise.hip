// HBM-bound elementwise / normalisation kernels of the FLUX DiT step (gfx950).
// Compiled with -ffp-contract=off so the fp32 arithmetic is the same sequence of roundings as the
// CPU oracle (oracle/dit_ref.py), which restates the bf16 tensor boundaries of the reference's
// diffusers modules [3p].
#include "common.h"
#include "kernels.h"

// ---------------------------------------------------------------------------------------------
// qkv_post: per-head RMSNorm(q), RMSNorm(k), RoPE(q), RoPE(k), head-major relayout, V transpose.
// Mirrors NativeFluxAttnProcessor2_0.__call__ (/root/reference/flux_piplines/texturing/
// attention_processor.py:42-87): view [B,S,H,128] -> norm_q/norm_k (RMSNorm eps 1e-6, weight) ->
// cat(text, image) along S -> apply_rotary_emb (interleaved pairs, fp32 math, cast back).
//   in : qkv [n_tok][ld] bf16, q at column q_col, k at k_col, v at v_col (each H*128 wide)
//   out: Qh, Kh [H][S_pad][128];  Vt [H][128][S_pad]  (row = tok_off + token)
// Block = 256 threads = 64 tokens x 1 head, 4 passes of 16 tokens; 16 lanes share a token row and each lane owns 8
// consecutive channels (4 rotation pairs): every global access is 16 bytes per lane (the one-pair-per-lane version moved
// 4 bytes per lane and ran at ~4 TB/s).

__global__ __launch_bounds__(256) void qkv_post_kernel(QkvPostParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t sv[64][136];  // V tile [token][d], 272-byte rows (16-byte aligned)
    const int tid = threadIdx.x;
    const int sub = tid & 15;            // 16-byte chunk of the 128-channel row
    const int trow = tid >> 4;           // token within a pass (0..15)
    const int head = blockIdx.y;
    const int t0 = blockIdx.x * 64;
    // head -> offset: plain [H][...] or grouped (sequence-parallel send layout: one group of heads per destination rank)
    // (a second level -- sub_heads: the heads of a destination rank cut into head groups whose exchanges are pipelined with attention -- puts head
    // hi of a rank at (hi / sub_heads) * gs2 + (hi % sub_heads) * hs)
    const int hg = p.heads_per_group > 0 ? head / p.heads_per_group : 0;
    int hi = p.heads_per_group > 0 ? head - hg * p.heads_per_group : head;
    long hoff_qk = (long)hg * p.gs_qk, hoff_v = (long)hg * p.gs_v;
    if (p.sub_heads > 0) { const int sg = hi / p.sub_heads; hi -= sg * p.sub_heads; hoff_qk += (long)sg * p.gs2_qk; hoff_v += (long)sg * p.gs2_v; }
    hoff_qk += (long)hi * p.hs_qk; hoff_v += (long)hi * p.hs_v;
    float wq[8], wk[8];
    {
        const uint4 a = *reinterpret_cast<const uint4*>((const bf16_t*)p.wq + 8 * sub);
        const uint4 b = *reinterpret_cast<const uint4*>((const bf16_t*)p.wk + 8 * sub);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            wq[2 * c] = bf2f((uint16_t)(aw[c] & 0xffff)); wq[2 * c + 1] = bf2f((uint16_t)(aw[c] >> 16));
            wk[2 * c] = bf2f((uint16_t)(bw[c] & 0xffff)); wk[2 * c + 1] = bf2f((uint16_t)(bw[c] >> 16));
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int tl = pass * 16 + trow;
        const int tok = t0 + tl;
        if (tok >= p.n_tok) {  // keep the LDS tile defined for the transpose
            *reinterpret_cast<uint4*>(&sv[tl][8 * sub]) = make_uint4(0, 0, 0, 0);
            continue;
        }
        const bf16_t* row = (const bf16_t*)p.qkv + (long)tok * p.ld + head * 128 + 8 * sub;
        const long srow = (long)(p.tok_off + tok);
        const uint4 vraw = *reinterpret_cast<const uint4*>(row + p.v_col);
        if (!p.skip_qk) {      // skip_qk: q / k came out of the GEMM's fused epilogue (utx_gemm_desc.qk_cols); only V is transposed here
        const float4 cs = *reinterpret_cast<const float4*>(p.cosb + srow * 64 + 4 * sub);
        const float4 sn = *reinterpret_cast<const float4*>(p.sinb + srow * 64 + 4 * sub);
        const float csv[4] = {cs.x, cs.y, cs.z, cs.w}, snv[4] = {sn.x, sn.y, sn.z, sn.w};
        const uint4 qraw = *reinterpret_cast<const uint4*>(row + p.q_col);
        const uint4 kraw = *reinterpret_cast<const uint4*>(row + p.k_col);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const uint4 raw = which ? kraw : qraw;
            const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
            float x[8];
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[2 * c] = bf2f((uint16_t)(rw[c] & 0xffff)); x[2 * c + 1] = bf2f((uint16_t)(rw[c] >> 16));
                ss += x[2 * c] * x[2 * c] + x[2 * c + 1] * x[2 * c + 1];
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);     // the 16 lanes of this token row
            const float rstd = 1.0f / sqrtf(ss / 128.0f + p.eps);
            const float qs = which ? 1.0f : p.q_scale;
            uint32_t ow[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float w0 = which ? wk[2 * c] : wq[2 * c], w1 = which ? wk[2 * c + 1] : wq[2 * c + 1];
                const float a0 = rbf(rbf(x[2 * c] * rstd) * w0);
                const float a1 = rbf(rbf(x[2 * c + 1] * rstd) * w1);
                const float r0 = (a0 * csv[c] + (-a1) * snv[c]) * qs;
                const float r1 = (a1 * csv[c] + a0 * snv[c]) * qs;
                ow[c] = pack2bf(r0, r1);
            }
            bf16_t* dst = (bf16_t*)(which ? p.Kh : p.Qh) + hoff_qk + srow * 128 + 8 * sub;
            *reinterpret_cast<uint4*>(dst) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        }
        *reinterpret_cast<uint4*>(&sv[tl][8 * sub]) = vraw;
    }
    __syncthreads();
    // transpose-store: thread -> (d = tid>>1, 32 tokens)
    const int d = tid >> 1, tb = (tid & 1) * 32;
    bf16_t* vrow = (bf16_t*)p.Vt + hoff_v + (long)d * p.S_pad + p.tok_off + t0 + tb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int tl = tb + 8 * c;
        const int valid = p.n_tok - (t0 + tl);  // tokens valid in this chunk of 8
        if (valid >= 8) {
            uint4 o;
            o.x = (uint32_t)sv[tl + 0][d] | ((uint32_t)sv[tl + 1][d] << 16);
            o.y = (uint32_t)sv[tl + 2][d] | ((uint32_t)sv[tl + 3][d] << 16);
            o.z = (uint32_t)sv[tl + 4][d] | ((uint32_t)sv[tl + 5][d] << 16);
            o.w = (uint32_t)sv[tl + 6][d] | ((uint32_t)sv[tl + 7][d] << 16);
            *reinterpret_cast<uint4*>(vrow + 8 * c) = o;
        } else {
            for (int e = 0; e < valid; ++e) vrow[8 * c + e] = sv[tl + e][d];
        }
    }
}

extern "C" int utx_launch_qkv_post(const QkvPostParams* hp, hipStream_t stream) {
    QkvPostParams p = *hp;
    if (p.n_tok <= 0 || p.H <= 0) return -1;
    if ((p.tok_off & 7) || (p.S_pad & 7) || (p.ld & 7) || (p.q_col & 7) || (p.k_col & 7) || (p.v_col & 7)) return -2;   // 16-byte lanes
    if (p.heads_per_group < 0 || (p.heads_per_group > 0 && ((p.H % p.heads_per_group) || (p.gs_qk & 7) || (p.gs_v & 7)))) return -2;
    if (p.sub_heads < 0 || (p.sub_heads > 0 && (p.heads_per_group <= 0 || (p.heads_per_group % p.sub_heads) || (p.gs2_qk & 7) || (p.gs2_v & 7)))) return -2;
    dim3 grid((p.n_tok + 63) / 64, p.H);
    hipLaunchKernelGGL(qkv_post_kernel, grid, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// OCP MX fp8 quantisation of the activation operand of an mx8 GEMM (unitex_hip.h: utx_quant_mx8).  One thread per block of 32
// elements: 64 B read (4 x 16 B), 32 B + 1 scale byte written.  HBM-bound: 3.03 B per element.
//   e = floor(log2(amax)) - 8  (exponent field of amax: bf16 inputs are exact in fp32), clamp [-127, 127]; amax == 0 -> -127
//   q = e4m3_rne(clamp(x * 2^-e, -448, 448))            (v_cvt_pk_fp8_f32: OCP e4m3fn on gfx950)
// PACKED (utx_quant_mx8_packed): the scale bytes in the tile-packed order the one-wave-per-SIMD GEMM reads (gemm_w4.hip, MX): dword
// [K-tile kt = blk / 4][row block rb = row / 128][l = row % 32][im = (row % 128) / 32] holds the four scale bytes blk % 4 = 0..3 of the K-tile; the four
// threads of a K-tile (consecutive lanes: nblk % 4 == 0) assemble the dword with two DPP-class shuffles and lane 0 of them stores it.  lds = row blocks
// per K-tile slab.  Same bytes, same q: only the scale addressing differs.
template <bool PACKED>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                        uint8_t* __restrict__ sc, long lds, int M, int nblk) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)M * nblk) return;      // PACKED: M * nblk is a multiple of 4, so the four lanes of a K-tile leave together
    const int row = (int)(t / nblk), blk = (int)(t - (long)row * nblk);
    const bf16_t* src = x + (long)row * ldx + blk * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 raw = *reinterpret_cast<const uint4*>(src + 8 * c);
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[8 * c + 2 * j] = bf2f((uint16_t)(w[j] & 0xffff)); v[8 * c + 2 * j + 1] = bf2f((uint16_t)(w[j] >> 16));
            amax = fmaxf(amax, fmaxf(fabsf(v[8 * c + 2 * j]), fabsf(v[8 * c + 2 * j + 1])));
        }
    }
    int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
    if (amax == 0.f || e < -127) e = -127;
    if (e > 127) e = 127;
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);     // 2^-e, exact (e in [-127, 127] -> exponent field 0..254; 0 only for e = 127)
    uint32_t out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = fminf(fmaxf(v[4 * j + c] * inv, -448.f), 448.f);
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], pk, true);
        out[j] = (uint32_t)pk;
    }
    uint4* dst = reinterpret_cast<uint4*>(q + (long)row * ldq + blk * 32);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    if constexpr (PACKED) {
        uint32_t w = (uint32_t)(e + 127) << (8 * (blk & 3));
        w |= (uint32_t)__shfl_xor((int)w, 1, 64);
        w |= (uint32_t)__shfl_xor((int)w, 2, 64);
        if ((blk & 3) == 0)
            reinterpret_cast<uint32_t*>(sc)[(((long)(blk >> 2) * lds + (row >> 7)) * 32 + (row & 31)) * 4 + ((row >> 5) & 3)] = w;
    } else sc[(long)row * lds + blk] = (uint8_t)(e + 127);
}

extern "C" int utx_launch_quant_mx8(const void* x, long ldx, void* q, long ldq, void* s, long lds, int M, int K, hipStream_t stream) {
    if (M <= 0 || K <= 0 || (K & 31) || (ldx & 7) || (ldq & 15) || lds < K / 32) return -2;
    const long total = (long)M * (K / 32);
    hipLaunchKernelGGL(quant_mx8_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)q, ldq,
                       (uint8_t*)s, lds, M, K / 32);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int utx_launch_quant_mx8_packed(const void* x, long ldx, void* q, long ldq, void* s, long row_blocks, int M, int K, hipStream_t stream) {
    if (M <= 0 || K <= 0 || (K & 127) || (ldx & 7) || (ldq & 15) || row_blocks < (M + 127) / 128 || ((uintptr_t)s & 15)) return -2;
    const long total = (long)M * (K / 32);
    hipLaunchKernelGGL(quant_mx8_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)q, ldq,
                       (uint8_t*)s, row_blocks, M, K / 32);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// Sequence-parallel exchange, receive side (unitex_hip.h: utx_sp_unpack_qkv / utx_sp_unpack_o): relayout of what the
// all-to-all delivered into the attention kernel's / the out-projection's operand layouts.  Pure copies: one 16-byte vector
// per thread and step, sources and destinations in runs of >= 128 B (S_loc % 64 == 0), grid-stride.
// text_rows > 0 ("key de-duplication across ranks", round 6): the caller guarantees that the first text_rows tokens of EVERY source rank's block are the same rows (the
// identical text tokens every rank carries, flux/transformer.py) -- K and V^T keep them once, from source rank 0, followed by the image tokens of rank 0 .. P-1:
// S_k = text_rows + P (S_loc - text_rows) keys in exactly the single-GPU order, so the rank's attention launch is the single-GPU launch (key multiplicity on tile 0
// only: the 4 x 64 kernel takes it) over H / P heads.  Q keeps every row (each rank wants its own rows back).
__global__ __launch_bounds__(256) void sp_unpack_qkv_kernel(const uint4* __restrict__ recv, int P, int Hp, int S_loc, int text_rows, uint4* __restrict__ q,
                                                            uint4* __restrict__ k, uint4* __restrict__ vt) {
    const long Ev = (long)S_loc * 16;                 // 16-byte vectors per (src, which, head) block: S_loc * 128 * 2 B / 16
    const long S = (long)P * S_loc;
    const long I = S_loc - text_rows;                 // tokens of a rank behind the shared text rows
    const long Sk = text_rows + (long)P * I;          // keys kept (== S when text_rows == 0)
    const long total = 3L * P * Hp * Ev;
    const long slv = S_loc / 8;                       // vectors per V^T row segment
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i % Ev;
        long b = i / Ev;
        const int hp = (int)(b % Hp); b /= Hp;
        const int which = (int)(b % 3);
        const int src = (int)(b / 3);
        if (which == 0) {
            q[((long)hp * S + (long)src * S_loc) * 16 + e] = recv[i];     // [hp][src*S_loc + tok][128]
        } else if (which == 1) {
            const long tok = e >> 4;
            if (tok < text_rows) { if (src == 0) k[((long)hp * Sk + tok) * 16 + (e & 15)] = recv[i]; }
            else k[((long)hp * Sk + text_rows + (long)src * I + (tok - text_rows)) * 16 + (e & 15)] = recv[i];      // [hp][text | img of src 0 | img of src 1 ...][128]
        } else {
            const long d = e / slv, t8 = e - d * slv;                       // eight tokens t8 * 8 .. of channel d
            if (t8 * 8 < text_rows) { if (src == 0) vt[((long)hp * 128 + d) * (Sk / 8) + t8] = recv[i]; }
            else vt[((long)hp * 128 + d) * (Sk / 8) + (text_rows + (long)src * I) / 8 + (t8 - text_rows / 8)] = recv[i];   // [hp][d][same key order]
        }
    }
}

__global__ __launch_bounds__(256) void sp_unpack_o_kernel(const uint4* __restrict__ recv, int P, int Hp, int S_loc, bf16_t* __restrict__ out, long ld, long src_cols) {
    const long Wv = (long)Hp * 16;                    // vectors per (src, tok) row: Hp * 128 * 2 B / 16
    const long total = (long)P * S_loc * Wv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long c = i % Wv;
        long b = i / Wv;
        const long tok = b % S_loc;
        const long src = b / S_loc;
        *reinterpret_cast<uint4*>(out + tok * ld + src * src_cols + c * 8) = recv[i];       // src_cols = Hp * 128 unless this is one head group of several
    }
}

extern "C" int utx_launch_sp_unpack_qkv(const void* recv, int P, int Hp, int S_loc, int text_rows, void* q, void* k, void* vt, hipStream_t stream) {
    if (P <= 0 || Hp <= 0 || S_loc <= 0 || (S_loc & 63) || text_rows < 0 || text_rows >= S_loc || (text_rows & 63)) return -2;
    if ((((uintptr_t)recv) | ((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)vt)) & 15) return -2;
    const long total = 3L * P * Hp * S_loc * 16;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sp_unpack_qkv_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint4*)recv, P, Hp, S_loc, text_rows, (uint4*)q, (uint4*)k, (uint4*)vt);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int utx_launch_sp_unpack_o(const void* recv, int P, int Hp, int S_loc, void* out, long ld, long src_cols, hipStream_t stream) {
    if (src_cols <= 0) src_cols = (long)Hp * 128;
    if (P <= 0 || Hp <= 0 || S_loc <= 0 || (ld & 7) || (src_cols & 7) || src_cols < (long)Hp * 128 || ld < (long)(P - 1) * src_cols + (long)Hp * 128) return -2;
    if ((((uintptr_t)recv) | ((uintptr_t)out)) & 15) return -2;
    const long total = (long)P * S_loc * Hp * 16;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sp_unpack_o_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint4*)recv, P, Hp, S_loc, (bf16_t*)out, ld, src_cols);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (no affine, eps) + AdaLN modulation:  y = bf16( bf16( bf16(LN(x)) * bf16(1+scale) ) + shift )
// AdaLayerNormZero / ZeroSingle / Continuous of the FLUX blocks [3p, SURVEY 3.4].

// One WAVE per token (four tokens per 256-thread block): a lane holds up to 8 chunks of 8 channels (D <= 4096) -- all its loads are in flight at
// once, both reductions are wave-level (no LDS, no block barrier).  (Round 1 used one 256-thread block per token with two __syncthreads
// reductions: 3.0 TB/s at D = 3072; the per-block latency chain, not HBM, set the time.)
// MXQ (utx_ln_mod_desc.q): the result leaves as OCP MX fp8 -- a block of 32 channels = the 4 consecutive chunks of lanes 4k .. 4k+3 (block maximum by
// two shuffles), a K-tile of 128 = 16 lanes (its four scale bytes assembled into one dword by two more shuffles, stored by the first of them at its
// tile-packed place); the bf16 rounding of the unfused path is kept in front of the quantisation, so the bytes equal utx_ln_mod -> utx_quant_mx8_packed.
template <bool MXQ>
__global__ __launch_bounds__(256) void ln_mod_kernel(LnModParams p) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= p.n_tok) return;
    const bf16_t* xr = (const bf16_t*)p.x + (long)tok * p.ldx;
    const int nchunk = p.D >> 3;
    uint4 raw[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = lane + 64 * it;
        raw[it] = (c < nchunk) ? *reinterpret_cast<const uint4*>(xr + c * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    float v[8][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const uint32_t w[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[it][2 * j] = bf2f((uint16_t)(w[j] & 0xffff));
            v[it][2 * j + 1] = bf2f((uint16_t)(w[j] >> 16));
            s += v[it][2 * j] + v[it][2 * j + 1];          // chunks beyond D are zeros
        }
    }
    const float mean = wave_sum(s) / (float)p.D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        if (lane + 64 * it < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = v[it][j] - mean; q += dlt * dlt; }
        }
    }
    const float var = wave_sum(q) / (float)p.D;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    bf16_t* yr = (bf16_t*)p.y + (long)tok * p.ldy;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            const uint4 sh = *reinterpret_cast<const uint4*>((const bf16_t*)p.shift + c * 8);
            const uint4 sc = *reinterpret_cast<const uint4*>((const bf16_t*)p.scale + c * 8);
            const uint32_t shw[4] = {sh.x, sh.y, sh.z, sh.w};
            const uint32_t scw[4] = {sc.x, sc.y, sc.z, sc.w};
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float shf = bf2f((uint16_t)(e ? (shw[j] >> 16) : (shw[j] & 0xffff)));
                    const float scf = bf2f((uint16_t)(e ? (scw[j] >> 16) : (scw[j] & 0xffff)));
                    const float n = rbf((v[it][2 * j + e] - mean) * rstd);
                    o[e] = rbf(n * rbf(1.0f + scf)) + shf;
                }
                ow[j] = pack2bf(o[0], o[1]);
            }
            if constexpr (!MXQ) *reinterpret_cast<uint4*>(yr + c * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            else {
                float f[8], amax = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[2 * j] = bf2f((uint16_t)(ow[j] & 0xffff)); f[2 * j + 1] = bf2f((uint16_t)(ow[j] >> 16));
                    amax = fmaxf(amax, fmaxf(fabsf(f[2 * j]), fabsf(f[2 * j + 1])));
                }
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
                if (amax == 0.f || e < -127) e = -127;
                if (e > 127) e = 127;
                const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
                float a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fminf(fmaxf(f[j] * inv, -448.f), 448.f);
                int p0 = 0, p1 = 0;
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], p0, false); p0 = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], p0, true);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(a[4], a[5], p1, false); p1 = __builtin_amdgcn_cvt_pk_fp8_f32(a[6], a[7], p1, true);
                *reinterpret_cast<uint2*>((uint8_t*)p.q + (long)tok * p.ldq + c * 8) = make_uint2((uint32_t)p0, (uint32_t)p1);
                uint32_t w = (uint32_t)(e + 127) << (8 * ((lane >> 2) & 3));
                w |= (uint32_t)__shfl_xor((int)w, 4, 64);
                w |= (uint32_t)__shfl_xor((int)w, 8, 64);
                if ((lane & 15) == 0)
                    reinterpret_cast<uint32_t*>(p.qs)[(((long)(c >> 4) * p.qs_row_blocks + (tok >> 7)) * 32 + (tok & 31)) * 4 + ((tok >> 5) & 3)] = w;
            }
        }
    }
}

extern "C" int utx_launch_ln_mod(const LnModParams* hp, hipStream_t stream) {
    LnModParams p = *hp;
    if (p.n_tok <= 0 || p.D <= 0 || (p.D & 7) || p.D > 4096 || (p.ldx & 7) || (p.ldy & 7)) return -2;
    if (p.q) {
        if (!p.qs || (p.D & 127) || (p.ldq & 7) || p.qs_row_blocks < (p.n_tok + 127) / 128 || ((uintptr_t)p.qs & 15) || ((uintptr_t)p.q & 7)) return -2;
        hipLaunchKernelGGL(ln_mod_kernel<true>, dim3((p.n_tok + 3) / 4), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    if (!p.y) return -2;
    hipLaunchKernelGGL(ln_mod_kernel<false>, dim3((p.n_tok + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// out = bf16(bf16(a + b) + c) (b may be null: out = bf16(a + c)): the sum of the timestep / guidance / pooled-text embeddings,
// conditioning = (timesteps_emb + guidance_emb) + pooled_projections in bf16 (CombinedTimestepGuidanceTextProjEmbeddings [3p]); n <= a few thousand.
__global__ __launch_bounds__(256) void add3_bf16_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* c, bf16_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = bf2f(a[i]);
    if (b) t = rbf(t + bf2f(b[i]));
    out[i] = f2bf(t + bf2f(c[i]));
}

extern "C" int utx_launch_add3_bf16(const void* a, const void* b, const void* c, void* out, int n, hipStream_t stream) {
    if (!a || !c || !out || n <= 0) return -2;
    hipLaunchKernelGGL(add3_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)out, n);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------
// Flow-match Euler step + condition re-pin, fused:
//   x[s] = bf16( float(x[s]) + dsigma * float(v[s]) )   for s <  n_noise   (scheduler.step, fp32 upcast)
//   x[s] = cond[s - n_noise]                           for s >= n_noise   (re-pin of clean condition tokens)
// /root/reference/flux_piplines/texturing/pipeline.py:644-645,660  (FlowMatchEulerDiscreteScheduler.step [3p])
// NB the reference steps ALL tokens and overwrites the tail at the top of the next iteration; the
// tail's stepped values are never observed, so writing the clean condition directly is equivalent.

__global__ __launch_bounds__(256) void sched_step_kernel(SchedParams p) {
    const long nchunk = p.n_total_elems >> 3;
    bf16_t* px = (bf16_t*)p.x; const bf16_t* pv = (const bf16_t*)p.v; const bf16_t* pcond = (const bf16_t*)p.cond;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (long)gridDim.x * blockDim.x) {
        const long e = c * 8;
        if (e < p.n_noise_elems) {
            const uint4 xr = *reinterpret_cast<const uint4*>(px + e);
            const uint4 vr = *reinterpret_cast<const uint4*>(pv + e);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
            const uint32_t vw[4] = {vr.x, vr.y, vr.z, vr.w};
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = bf2f((uint16_t)(xw[j] & 0xffff)) + p.dsigma * bf2f((uint16_t)(vw[j] & 0xffff));
                const float a1 = bf2f((uint16_t)(xw[j] >> 16)) + p.dsigma * bf2f((uint16_t)(vw[j] >> 16));
                ow[j] = pack2bf(a0, a1);
            }
            *reinterpret_cast<uint4*>(px + e) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        } else if (pcond) {
            *reinterpret_cast<uint4*>(px + e) = *reinterpret_cast<const uint4*>(pcond + (e - p.n_noise_elems));
        }
    }
}

extern "C" int utx_launch_sched_step(const SchedParams* hp, hipStream_t stream) {
    SchedParams p = *hp;
    if (p.n_total_elems <= 0 || (p.n_total_elems & 7) || (p.n_noise_elems & 7)) return -2;
    long nchunk = p.n_total_elems >> 3;
    int blocks = (int)((nchunk + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sched_step_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
