// Does utx_qkv_post, BUILT WITH packed fp32 instructions, go wrong beside one MFMA kernel of another stream -- outside the DiT step?
//
// Round 4 tied the two-stream nondeterminism of the double blocks to the packed fp32 instructions of dit_elementwise.hip (DESIGN 0, 9 b) and
// tools/pk_fp32_mfma_probe.hip (a generic v_pk_* chain beside generic neighbours) did not reproduce it.  This probe keeps the real instruction stream:
// this translation unit COMPILES csrc/dit_elementwise.hip itself, with hipcc's defaults (packed fp32 on) -- the victim is that qkv_post_kernel, the
// control is the library's own (built without packed fp32: csrc/build.py), the neighbours are the library's GEMM at the text half's shapes
// (utx_gemm_bf16, M = 512 / 64 / 4096, N = 9216, K = 3072) or a bare MFMA loop.  Every victim launch is compared dword for dword with the result of a
// launch that ran alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iunitex_amd/csrc -Iinclude tools/qkv_post_pk_probe.hip -Lunitex_amd/lib -lunitex_hip \
//         -Wl,-rpath,$PWD/unitex_amd/lib -o /tmp/qkv_pk_probe && /tmp/qkv_pk_probe [iterations] [tokens]
#include "../unitex_amd/csrc/dit_elementwise.hip"      // the victim: qkv_post_kernel / utx_launch_qkv_post of THIS build (packed fp32 on)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef float pf16v __attribute__((ext_vector_type(16)));
typedef __bf16 pbf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void bare_mfma(float* sink, int iters) {
    pf16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    pbf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t == 12345.678f) sink[0] = t;
}

__global__ __launch_bounds__(256) void cmp_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, long n, unsigned long long* cnt, unsigned long long* first) {
    unsigned long long bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (a[i] != b[i]) { ++bad; atomicMin(first, (unsigned long long)i); }
    if (bad) atomicAdd(cnt, bad);
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float unif(uint32_t& s) { return (float)(lcg(s) >> 8) * (1.0f / 16777216.0f); }
static float gauss(uint32_t& s) { float u1 = unif(s) + 1e-7f, u2 = unif(s); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    const int n_tok = argc > 2 ? atoi(argv[2]) : 16384;
    const int H = 24, D = 3072, ld = 3 * D;
    utx_ctx* ctx = nullptr;
    if (utx_init(0, &ctx) != 0) { printf("utx_init failed\n"); return 2; }
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

    // ---- the victim's operands
    uint32_t seed = 12345;
    std::vector<uint16_t> pool(1 << 20);      // one million normal deviates, tiled with a stride coprime to every row length (host time is GPU-box time)
    for (auto& v : pool) v = f2bf(gauss(seed));
    auto fill = [&](std::vector<uint16_t>& dst, float) { for (size_t i = 0; i < dst.size(); ++i) dst[i] = pool[(i * 7 + (i >> 20) * 13) & ((1 << 20) - 1)]; };
    std::vector<uint16_t> h_qkv((size_t)n_tok * ld), h_w(256);
    fill(h_qkv, 1.f);
    for (auto& v : h_w) v = f2bf(1.0f + 0.1f * gauss(seed));
    std::vector<float> h_cos((size_t)n_tok * 64), h_sin((size_t)n_tok * 64);
    for (size_t i = 0; i < h_cos.size(); ++i) { const float th = 6.2831853f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f; h_cos[i] = cosf(th); h_sin[i] = sinf(th); }
    void *qkv, *wq, *wk, *cosb, *sinb, *Qh, *Kh, *Vt, *Qr, *Kr, *Vr;
    const size_t hb = (size_t)H * n_tok * 128 * 2;
    CK(hipMalloc(&qkv, h_qkv.size() * 2)); CK(hipMalloc(&wq, 256)); CK(hipMalloc(&wk, 256)); CK(hipMalloc(&cosb, h_cos.size() * 4)); CK(hipMalloc(&sinb, h_sin.size() * 4));
    for (void** p : {&Qh, &Kh, &Vt, &Qr, &Kr, &Vr}) { CK(hipMalloc(p, hb)); CK(hipMemset(*p, 0, hb)); }
    CK(hipMemcpy(qkv, h_qkv.data(), h_qkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wq, h_w.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(wk, h_w.data() + 128, 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(cosb, h_cos.data(), h_cos.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sinb, h_sin.data(), h_sin.size() * 4, hipMemcpyHostToDevice));
    utx_qkv_post_desc d;
    memset(&d, 0, sizeof(d));
    d.qkv = qkv; d.ld = ld; d.q_col = 0; d.k_col = D; d.v_col = 2 * D; d.wq = wq; d.wk = wk; d.cosb = (const float*)cosb; d.sinb = (const float*)sinb;
    d.Qh = Qh; d.Kh = Kh; d.Vt = Vt; d.hs_qk = (long)n_tok * 128; d.hs_v = (long)128 * n_tok; d.S_pad = n_tok; d.n_tok = n_tok; d.tok_off = 0; d.H = H;
    d.eps = 1e-6f; d.q_scale = 0.08838834764831845f * 1.4426950408889634f;
    utx_qkv_post_desc dr = d;
    dr.Qh = Qr; dr.Kh = Kr; dr.Vt = Vr;

    // ---- the neighbours' operands: text-half GEMMs [M, 3072] x [9216, 3072]^T + bias
    const int Mmax = 4096, N = 9216, K = 3072;
    std::vector<uint16_t> h_a((size_t)Mmax * K), h_b((size_t)N * K);
    fill(h_a, 1.f);
    for (size_t i = 0; i < h_b.size(); ++i) h_b[i] = f2bf(0.02f * (float)((int)((i * 2654435761u) >> 24 & 15) - 8));
    void *A, *B, *Cc, *bias;
    CK(hipMalloc(&A, h_a.size() * 2)); CK(hipMalloc(&B, h_b.size() * 2)); CK(hipMalloc(&Cc, (size_t)Mmax * N * 2)); CK(hipMalloc(&bias, N * 2));
    CK(hipMemcpy(A, h_a.data(), h_a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h_b.data(), h_b.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 2));
    utx_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = K; g.B = B; g.ldb = K; g.N = N; g.K = K; g.K2 = 0; g.lora_seg_n = 128; g.alpha = 1.0f; g.bias = bias; g.gelu_from = N; g.C = Cc; g.ldc = N; g.n_split = N;
    float* sink; CK(hipMalloc(&sink, 16));
    unsigned long long* cnt; CK(hipMalloc(&cnt, 16));

    // reference: the packed-fp32 victim alone; the library's (no packed fp32) build must give the same bits
    if (utx_launch_qkv_post(&dr, s1) != 0) { printf("victim launch failed\n"); return 2; }
    CK(hipStreamSynchronize(s1));
    if (utx_qkv_post(ctx, &d, (utx_stream)s1) != 0) { printf("library qkv_post failed\n"); return 2; }
    const long ndw = (long)(hb / 4);
    const unsigned long long none = ~0ull;
    unsigned long long h_cnt[2];
    h_cnt[0] = 0; h_cnt[1] = none;
    CK(hipMemcpyAsync(cnt, h_cnt, 16, hipMemcpyHostToDevice, s1));
    hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, s1, (const uint32_t*)Qh, (const uint32_t*)Qr, ndw, cnt, cnt + 1);
    hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, s1, (const uint32_t*)Kh, (const uint32_t*)Kr, ndw, cnt, cnt + 1);
    hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, s1, (const uint32_t*)Vt, (const uint32_t*)Vr, ndw, cnt, cnt + 1);
    CK(hipMemcpyAsync(h_cnt, cnt, 16, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
    printf("{\"check\": \"packed-fp32 build of qkv_post (this binary) vs the library's build without packed fp32, both alone\", \"differing_dwords\": %llu, \"tokens\": %d, \"heads\": %d}\n", h_cnt[0], n_tok, H);

    struct Arm { const char* name; int victim_pk; int gemm_m; int bare; };
    const Arm arms[] = {
        {"packed victim alone", 1, 0, 0},
        {"packed victim beside utx_gemm_bf16 M=512 N=9216 K=3072 on a second stream", 1, 512, 0},
        {"packed victim beside utx_gemm_bf16 M=64", 1, 64, 0},
        {"packed victim beside utx_gemm_bf16 M=4096", 1, 4096, 0},
        {"packed victim beside a bare v_mfma_f32_32x32x16_bf16 loop", 1, 0, 1},
        {"library victim (no packed fp32) beside utx_gemm_bf16 M=512", 0, 512, 0},
        {"library victim (no packed fp32) beside utx_gemm_bf16 M=64", 0, 64, 0},
    };
    int ncu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
    for (const Arm& arm : arms) {
        unsigned long long bad_iters = 0, bad_dwords = 0, first_idx = none;
        int first_buf = -1, first_iter = -1;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        const int batch = 25;
        for (int it0 = 0; it0 < iters; it0 += batch) {
            // per iteration of the batch: neighbours on s2, victim on s1, compare Q and K on s1 (counts accumulate per batch: buffer 0 / 1 in cnt[0..1] / cnt[2..3] is overkill -- one counter, first index)
            h_cnt[0] = 0; h_cnt[1] = none;
            CK(hipMemcpyAsync(cnt, h_cnt, 16, hipMemcpyHostToDevice, s1));
            for (int it = it0; it < it0 + batch && it < iters; ++it) {
                if (arm.gemm_m) {
                    g.M = arm.gemm_m;
                    const int reps = arm.gemm_m >= 4096 ? 1 : 3;
                    for (int r = 0; r < reps; ++r)
                        if (utx_gemm_bf16(ctx, &g, (utx_stream)s2) != 0) { printf("gemm failed: %s\n", utx_last_error(ctx)); return 2; }
                }
                if (arm.bare) hipLaunchKernelGGL(bare_mfma, dim3(ncu), dim3(256), 0, s2, sink, 6000);
                CK(hipMemsetAsync(Qh, 0xff, 256, s1));      // a stale result must not pass for a fresh one
                const int rc = arm.victim_pk ? utx_launch_qkv_post(&d, s1) : utx_qkv_post(ctx, &d, (utx_stream)s1);
                if (rc != 0) { printf("victim failed\n"); return 2; }
                hipLaunchKernelGGL(cmp_kernel, dim3(1024), dim3(256), 0, s1, (const uint32_t*)Qh, (const uint32_t*)Qr, ndw, cnt, cnt + 1);
                hipLaunchKernelGGL(cmp_kernel, dim3(1024), dim3(256), 0, s1, (const uint32_t*)Kh, (const uint32_t*)Kr, ndw, cnt, cnt + 1);
            }
            CK(hipMemcpyAsync(h_cnt, cnt, 16, hipMemcpyDeviceToHost, s1));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            if (h_cnt[0]) {
                ++bad_iters; bad_dwords += h_cnt[0];
                if (first_idx == none) { first_idx = h_cnt[1]; first_iter = it0; first_buf = 0; }
            }
        }
        CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"arm\": \"%s\", \"victim_launches\": %d, \"batches_of_%d_with_a_difference\": %llu, \"differing_dwords\": %llu", arm.name, iters, batch, bad_iters, bad_dwords);
        if (first_idx != none) {
            const long i = (long)first_idx;      // dword index into [H][n_tok][64 dwords]
            printf(", \"first\": {\"batch_at\": %d, \"head\": %ld, \"token\": %ld, \"channel_pair\": %ld, \"lane_of_its_wave\": %ld}", first_iter, i / ((long)n_tok * 64), (i / 64) % n_tok,
                   i % 64, ((i / 64) % 4) * 16 + (i % 64) / 4);
        }
        printf(", \"ms\": %.1f}\n", ms);
        fflush(stdout);
        (void)first_buf;
    }
    utx_free(ctx);
    return 0;
}
