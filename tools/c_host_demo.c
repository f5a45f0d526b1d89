/* A NON-PYTHON host of the DiT step: plain C over include/unitex_hip.h.
 *   hipcc -O2 -x c tools/c_host_demo.c -Iinclude -Lunitex_amd/lib -lunitex_hip -Wl,-rpath,$PWD/unitex_amd/lib -o tools/bin/c_host_demo && tools/bin/c_host_demo
 * Allocates a small FLUX-shaped transformer (2 heads, 1 double + 2 single blocks, D = 256) with hipMalloc, fills weights and inputs with a deterministic pattern,
 * lets utx_dit_load assemble the step, runs it twice through utx_dit_step on a stream of its own and checks that the prediction is finite, non-trivial and
 * reproducible.  No Python, no torch: the library borrows raw device pointers only.  (Numerical parity of the plan itself is the job of the GPU tests, which
 * compare the C-built plan byte for byte with the one the Python host builds and both against the CPU oracle.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "unitex_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng = 12345u;
static float frand(void) { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }     /* [-1, 1) */

static void* dev_bf16(size_t n, float scale, float offset) {      /* n bf16 values scale * U(-1, 1) + offset on the device */
    uint16_t* h = (uint16_t*)malloc(n * 2);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(scale * frand() + offset);
    void* d; CHECK(hipMalloc(&d, n * 2)); CHECK(hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice)); free(h);
    return d;
}
static void* dev_zero(size_t bytes) { void* d; CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 0, bytes)); return d; }
static utx_dit_linear linear(int out_f, int in_f, float gain) {
    utx_dit_linear L; memset(&L, 0, sizeof(L));
    L.w = dev_bf16((size_t)out_f * in_f, gain / sqrtf((float)in_f), 0.f);
    L.b = dev_bf16((size_t)out_f, 0.02f, 0.f);
    return L;
}

int main(void) {
    enum { H = 2, D = 256, ND = 1, NS = 2, IN = 64, JOINT = 64, POOLED = 64, MR = 4, S_TXT = 64, S_IMG = 448, S = S_TXT + S_IMG, S_PAD = 512 };
    utx_ctx* ctx = NULL;
    int rc = utx_init(0, &ctx);
    if (rc) { fprintf(stderr, "utx_init -> %d (an MI355X / gfx950 device is required)\n", rc); return 1; }
    utx_dit_config cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.num_heads = H; cfg.num_double = ND; cfg.num_single = NS; cfg.in_channels = IN; cfg.joint_dim = JOINT; cfg.pooled_dim = POOLED; cfg.mlp_ratio = MR;
    cfg.guidance_embeds = 1; cfg.S_txt = S_TXT; cfg.S_img = S_IMG; cfg.n_out = 192; cfg.two_streams = 1; cfg.n_cus = 256;
    /* weights, packed as include/unitex_hip.h describes */
    utx_dit_double_block dbl[ND]; utx_dit_single_block sgl[NS];
    utx_dit_weights wt; memset(&wt, 0, sizeof(wt));
    wt.x_embedder = linear(D, IN, 1.f); wt.context_embedder = linear(D, JOINT, 1.f); wt.proj_out = linear(IN, D, 1.f);
    wt.t_lin1 = linear(D, 256, 1.f); wt.t_lin2 = linear(D, D, 1.f); wt.g_lin1 = linear(D, 256, 1.f); wt.g_lin2 = linear(D, D, 1.f);
    wt.p_lin1 = linear(D, POOLED, 1.f); wt.p_lin2 = linear(D, D, 1.f);
    int off = 0;
    for (int i = 0; i < ND; ++i) {
        memset(&dbl[i], 0, sizeof(dbl[i]));
        dbl[i].qkv_x = linear(3 * D, D, 1.f); dbl[i].qkv_c = linear(3 * D, D, 1.f); dbl[i].out_x = linear(D, D, 1.f); dbl[i].out_c = linear(D, D, 1.f);
        dbl[i].ff1_x = linear(MR * D, D, 1.f); dbl[i].ff2_x = linear(D, MR * D, 1.f); dbl[i].ff1_c = linear(MR * D, D, 1.f); dbl[i].ff2_c = linear(D, MR * D, 1.f);
        dbl[i].nq = dev_bf16(128, 0.1f, 1.f); dbl[i].nk = dev_bf16(128, 0.1f, 1.f); dbl[i].naq = dev_bf16(128, 0.1f, 1.f); dbl[i].nak = dev_bf16(128, 0.1f, 1.f);
        dbl[i].mod_x = off; off += 6 * D; dbl[i].mod_c = off; off += 6 * D;
    }
    for (int i = 0; i < NS; ++i) {
        memset(&sgl[i], 0, sizeof(sgl[i]));
        sgl[i].qkvm = linear((3 + MR) * D, D, 1.f); sgl[i].out = linear(D, (1 + MR) * D, 1.f);
        sgl[i].nq = dev_bf16(128, 0.1f, 1.f); sgl[i].nk = dev_bf16(128, 0.1f, 1.f);
        sgl[i].mod = off; off += 3 * D;
    }
    wt.mod_out = off; off += 2 * D; wt.n_mod = off;
    wt.mod = linear(off, D, 0.5f);
    wt.dbl = dbl; wt.sgl = sgl;
    /* workspaces */
    utx_dit_workspace ws; memset(&ws, 0, sizeof(ws));
    ws.lat = dev_bf16((size_t)S_IMG * IN, 1.f, 0.f); ws.enc = dev_bf16((size_t)S_TXT * JOINT, 0.5f, 0.f); ws.pooled = dev_bf16(POOLED, 0.5f, 0.f);
    {   /* sinusoidal projections of t = 500 and guidance = 3500: cos | sin halves (diffusers Timesteps(256, flip_sin_to_cos=True)) */
        uint16_t tp[256], gp[256];
        for (int i = 0; i < 128; ++i) {
            const float f = expf(-logf(10000.f) * (float)i / 128.f);
            tp[i] = f2bf(cosf(500.f * f)); tp[128 + i] = f2bf(sinf(500.f * f)); gp[i] = f2bf(cosf(3500.f * f)); gp[128 + i] = f2bf(sinf(3500.f * f));
        }
        CHECK(hipMalloc(&ws.tproj, 512)); CHECK(hipMemcpy(ws.tproj, tp, 512, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&ws.gproj, 512)); CHECK(hipMemcpy(ws.gproj, gp, 512, hipMemcpyHostToDevice));
    }
    ws.e1 = dev_zero(D * 2); ws.e_t = dev_zero(D * 2); ws.e_g = dev_zero(D * 2); ws.e_p = dev_zero(D * 2); ws.temb = dev_zero(D * 2); ws.mod = dev_zero((size_t)off * 2);
    ws.h = dev_zero((size_t)S * D * 2); ws.xn = dev_zero((size_t)S * D * 2); ws.qkv = dev_zero((size_t)S * 3 * D * 2); ws.cat = dev_zero((size_t)S * (1 + MR) * D * 2);
    ws.attn = dev_zero((size_t)S * D * 2); ws.out = dev_zero((size_t)S_IMG * IN * 2);
    {   /* RoPE tables: position (0, y, x) of a 16 x 28 token grid for the image rows, zeros for the text rows; axes (16, 56, 56), theta 10000 */
        float* c = (float*)malloc((size_t)S * 64 * 4); float* s = (float*)malloc((size_t)S * 64 * 4);
        const int axes[3] = {16, 56, 56};
        for (int t = 0; t < S; ++t) {
            const double pos[3] = {0.0, t < S_TXT ? 0.0 : (double)((t - S_TXT) / 28), t < S_TXT ? 0.0 : (double)((t - S_TXT) % 28)};
            int col = 0;
            for (int a = 0; a < 3; ++a)
                for (int j = 0; j < axes[a] / 2; ++j, ++col) {
                    const double ang = pos[a] / pow(10000.0, (2.0 * j) / axes[a]);
                    c[(size_t)t * 64 + col] = (float)cos(ang); s[(size_t)t * 64 + col] = (float)sin(ang);
                }
        }
        CHECK(hipMalloc((void**)&ws.cos, (size_t)S * 64 * 4)); CHECK(hipMemcpy(ws.cos, c, (size_t)S * 64 * 4, hipMemcpyHostToDevice));
        CHECK(hipMalloc((void**)&ws.sin, (size_t)S * 64 * 4)); CHECK(hipMemcpy(ws.sin, s, (size_t)S * 64 * 4, hipMemcpyHostToDevice));
        free(c); free(s);
    }
    ws.Qh = dev_zero((size_t)H * S_PAD * 128 * 2); ws.Kh = dev_zero((size_t)H * S_PAD * 128 * 2); ws.Vt = dev_zero((size_t)H * 128 * S_PAD * 2);
    ws.attn_work_bytes = utx_attn_workspace_bytes(ctx, H, S, S);
    if (ws.attn_work_bytes) ws.attn_work = dev_zero(ws.attn_work_bytes);
    utx_plan* plan = NULL;
    rc = utx_dit_load(ctx, &cfg, &wt, &ws, &plan);
    if (rc) { fprintf(stderr, "utx_dit_load -> %d: %s\n", rc, utx_last_error(ctx)); return 1; }
    hipStream_t st; CHECK(hipStreamCreate(&st));
    uint16_t* out[2];
    for (int rep = 0; rep < 2; ++rep) {
        int bad = -1;
        rc = utx_dit_step(plan, (utx_stream)st, &bad);
        if (rc) { fprintf(stderr, "utx_dit_step -> %d at entry %d\n", rc, bad); return 1; }
        CHECK(hipStreamSynchronize(st));
        out[rep] = (uint16_t*)malloc((size_t)cfg.n_out * IN * 2);
        CHECK(hipMemcpy(out[rep], ws.out, (size_t)cfg.n_out * IN * 2, hipMemcpyDeviceToHost));
    }
    double sum = 0, asum = 0; int finite = 1;
    for (int i = 0; i < cfg.n_out * IN; ++i) { const float v = bf2f(out[0][i]); if (!isfinite(v)) finite = 0; sum += v; asum += fabs(v); }
    const int same = memcmp(out[0], out[1], (size_t)cfg.n_out * IN * 2) == 0;
    printf("c_host_demo: %d plan entries, noise prediction [%d x %d]: sum %.4f mean|v| %.4f finite %d reproducible %d\n", utx_plan_size(plan), cfg.n_out, IN, sum,
           asum / (cfg.n_out * IN), finite, same);
    utx_plan_free(plan); utx_free(ctx);
    return (finite && same && asum > 1.0) ? 0 : 1;
}
