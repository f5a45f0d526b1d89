// utx_plan: a replayable list of stream-ordered launches -- the C side of FluxDiT's per-step plan (`utx_dit_step`).
//
// The Python host builds one denoise step as a flat list of (entry point, descriptor) pairs over fixed workspaces (unitex_amd/flux/transformer.py).
// Replaying that list is ~700 ctypes calls per step; here the descriptors are COPIED once into a utx_plan and utx_plan_run replays them with plain C
// calls of the same launchers -- one C call per step, the same kernels in the same order on the same streams (bit-identical by construction), usable
// from any language that can fill the descriptors, and capturable into a HIP graph (nothing synchronises, nothing allocates).
// Two-stream sections (the text half of a double block beside its image half): FORK records an event on the caller's stream and makes the plan's side
// stream wait for it; entries tagged `side` launch there; JOIN records on the side stream and makes the caller's stream wait.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "kernels.h"

namespace {
enum Kind { K_GEMM, K_GEMV, K_LN_MOD, K_QKV_POST, K_ATTN, K_QUANT, K_ADD3, K_FORK, K_JOIN, K_QUANT_VT, K_ATTN8 };
struct AttnArgs { const void *q, *k, *vt; void* o; long q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss; int H, S_q, S_kv; float scale, kb; int period; void* work; size_t work_bytes; };
struct QuantArgs { const void* x; long ldx; void* q; long ldq; void* s; long lds; int M, K, packed; };
struct Add3Args { const void *a, *b, *c; void* out; int n; };
struct QuantVtArgs { const void* vt; void *v8, *vs; int H, S_pad; };
struct Entry {
    Kind kind; int side;
    union { utx_gemm_desc gemm; utx_gemv_desc gemv; utx_ln_mod_desc ln; utx_qkv_post_desc qkv; AttnArgs attn; QuantArgs quant; Add3Args add3; QuantVtArgs qvt; Attn8Params attn8; };
    Entry() { memset(this, 0, sizeof(*this)); }
};
}  // namespace

struct utx_plan {
    std::vector<Entry> entries;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> events;     // one (fork, join) pair per two-stream section
    int cur_side = 0, open_sections = 0;
    int device = 0;
};

extern "C" int utx_launch_add3_bf16(const void* a, const void* b, const void* c, void* out, int n, hipStream_t stream);   // dit_elementwise.hip

extern "C" {

int utx_plan_create(utx_ctx* ctx, utx_plan** out) {
    (void)ctx;
    if (!out) return -2;
    utx_plan* p = new utx_plan();
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; return -5; }
    *out = p;
    return 0;
}

void utx_plan_free(utx_plan* p) {
    if (!p) return;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
}

int utx_plan_size(const utx_plan* p) { return p ? (int)p->entries.size() : -2; }

static Entry& push(utx_plan* p, Kind k) { p->entries.emplace_back(); Entry& e = p->entries.back(); e.kind = k; e.side = p->cur_side; return e; }

int utx_plan_add_gemm(utx_plan* p, const utx_gemm_desc* d) { if (!p || !d) return -2; push(p, K_GEMM).gemm = *d; return 0; }
int utx_plan_add_gemv(utx_plan* p, const utx_gemv_desc* d) { if (!p || !d) return -2; push(p, K_GEMV).gemv = *d; return 0; }
int utx_plan_add_ln_mod(utx_plan* p, const utx_ln_mod_desc* d) { if (!p || !d) return -2; push(p, K_LN_MOD).ln = *d; return 0; }
int utx_plan_add_qkv_post(utx_plan* p, const utx_qkv_post_desc* d) { if (!p || !d) return -2; push(p, K_QKV_POST).qkv = *d; return 0; }
int utx_plan_add_attn(utx_plan* p, const void* q, const void* k, const void* vt, void* o, long q_hs, long q_ss, long k_hs, long k_ss, long vt_hs,
                      long vt_ds, long o_ss, int H, int S_q, int S_kv, float softmax_scale, float key_bias_log2, int key_bias_period, void* work,
                      size_t work_bytes) {
    if (!p || !q || !k || !vt || !o) return -2;
    AttnArgs a = {q, k, vt, o, q_hs, q_ss, k_hs, k_ss, vt_hs, vt_ds, o_ss, H, S_q, S_kv, softmax_scale, key_bias_log2, key_bias_period, work, work_bytes};
    push(p, K_ATTN).attn = a;
    return 0;
}
int utx_plan_add_quant_mx8(utx_plan* p, const void* x, long ldx, void* q, long ldq, void* s, long lds_or_row_blocks, int M, int K, int packed) {
    if (!p || !x || !q || !s) return -2;
    QuantArgs a = {x, ldx, q, ldq, s, lds_or_row_blocks, M, K, packed};
    push(p, K_QUANT).quant = a;
    return 0;
}
int utx_plan_add_add3(utx_plan* p, const void* a, const void* b, const void* c, void* out, int n) {
    if (!p || !a || !c || !out || n <= 0) return -2;      // b may be NULL: out = a + c
    Add3Args g = {a, b, c, out, n};
    push(p, K_ADD3).add3 = g;
    return 0;
}
// the opt-in MX fp8 attention (attention_fp8.hip): V^T quantised along the keys, and the attention over utx_quant_mx8's Q8 / K8 and that V8^T
int utx_plan_add_quant_vt_mx8(utx_plan* p, const void* vt, void* v8, void* vs, int H, int S_pad) {
    if (!p || !vt || !v8 || !vs) return -2;
    QuantVtArgs a = {vt, v8, vs, H, S_pad};
    push(p, K_QUANT_VT).qvt = a;
    return 0;
}
int utx_plan_add_attn_fp8(utx_plan* p, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                          int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period) {
    if (!p || !q8 || !qs || !k8 || !ks || !v8t || !vs || !o) return -2;
    Attn8Params a;
    memset(&a, 0, sizeof(a));
    a.q8 = (const uint8_t*)q8; a.k8 = (const uint8_t*)k8; a.v8t = (const uint8_t*)v8t;
    a.qs = (const uint32_t*)qs; a.ks = (const uint32_t*)ks; a.vs = (const uint32_t*)vs;
    a.o = (bf16_t*)o; a.o_ss = o_ss; a.H = H; a.S = S_kv; a.Sq = S_q; a.S_pad = S_pad;
    a.key_bias_log2 = key_bias_log2; a.key_bias_period = key_bias_period;
    push(p, K_ATTN8).attn8 = a;
    return 0;
}
// the same with caller-owned scratch for the key-split tail round (utx_attn_workspace_bytes; the plan holds the pointer: the buffer must outlive the plan)
int utx_plan_add_attn_fp8_ws(utx_plan* p, const void* q8, const void* qs, const void* k8, const void* ks, const void* v8t, const void* vs, void* o, long o_ss,
                             int H, int S_q, int S_kv, int S_pad, float key_bias_log2, int key_bias_period, void* work, size_t work_bytes) {
    const int rc = utx_plan_add_attn_fp8(p, q8, qs, k8, ks, v8t, vs, o, o_ss, H, S_q, S_kv, S_pad, key_bias_log2, key_bias_period);
    if (rc) return rc;
    p->entries.back().attn8.work = work;
    p->entries.back().attn8.work_bytes = work ? work_bytes : 0;
    return 0;
}
// two-stream section: fork; [side entries]; utx_plan_main; [main entries]; join
int utx_plan_fork(utx_plan* p) {
    if (!p || p->cur_side || p->open_sections) return -2;
    if (!p->side && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) return -5;
    hipEvent_t a, b;
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess) return -5;
    if (hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(a); return -5; }
    p->events.push_back(a); p->events.push_back(b);
    push(p, K_FORK);
    p->cur_side = 1; p->open_sections = 1;
    return 0;
}
int utx_plan_main(utx_plan* p) { if (!p || !p->open_sections) return -2; p->cur_side = 0; return 0; }
int utx_plan_join(utx_plan* p) {
    if (!p || !p->open_sections) return -2;
    p->cur_side = 0; p->open_sections = 0;
    push(p, K_JOIN);
    return 0;
}

// the split-tail scratch of the large-M GEMM (utx_gemm_desc.sk_work) for the GEMMs of the caller's stream -- they are ordered among themselves; the side
// stream's GEMMs (the text half: far below the size where a tail is split) get none.  Only when some launch has more 256 x 256 tiles than CUs
// (FluxDiT._assign_streamk).  Returns the number of descriptors that received the scratch.
int utx_plan_assign_sk(utx_plan* p, void* sk_work, size_t sk_work_bytes, int n_cus) {
    if (!p || !sk_work || n_cus <= 0) return 0;
    bool any = false;
    for (const Entry& e : p->entries)
        if (e.kind == K_GEMM && !e.side && (long)((e.gemm.M + 255) / 256) * (e.gemm.N / 256) > n_cus) any = true;
    if (!any) return 0;
    int n = 0;
    for (Entry& e : p->entries)
        if (e.kind == K_GEMM && !e.side) { e.gemm.sk_work = sk_work; e.gemm.sk_work_bytes = sk_work_bytes; ++n; }
    return n;
}

// Read an entry back (tests, debuggers): kind 0 gemm, 1 gemv, 2 ln_mod, 3 qkv_post, 4 attention, 5 quant_mx8, 6 add3, 7 fork, 8 join, 9 quant_vt_mx8, 10 fp8
// attention (its argument block = {q8, k8, v8t, qs, ks, vs, o, ...} as kernels.h's Attn8Params lays them out); `side` = launched on the
// plan's side stream; the descriptor (kinds 0-3: the public structs) or the entry's argument block (kinds 4-6) is copied into buf.  Returns the bytes copied,
// or a negative code.
int utx_plan_entry(const utx_plan* p, int i, int* kind, int* side, void* buf, size_t cap) {
    if (!p || i < 0 || i >= (int)p->entries.size() || !kind || !side) return -2;
    const Entry& e = p->entries[i];
    *kind = (int)e.kind; *side = e.side;
    const void* src = nullptr; size_t n = 0;
    switch (e.kind) {
        case K_GEMM: src = &e.gemm; n = sizeof(e.gemm); break;
        case K_GEMV: src = &e.gemv; n = sizeof(e.gemv); break;
        case K_LN_MOD: src = &e.ln; n = sizeof(e.ln); break;
        case K_QKV_POST: src = &e.qkv; n = sizeof(e.qkv); break;
        case K_ATTN: src = &e.attn; n = sizeof(e.attn); break;
        case K_QUANT: src = &e.quant; n = sizeof(e.quant); break;
        case K_ADD3: src = &e.add3; n = sizeof(e.add3); break;
        case K_QUANT_VT: src = &e.qvt; n = sizeof(e.qvt); break;
        case K_ATTN8: src = &e.attn8; n = sizeof(e.attn8); break;
        default: break;
    }
    if (n > cap) return -2;
    if (n && buf) memcpy(buf, src, n);
    return (int)n;
}

// Replay.  Returns 0, or the first failing launcher's code with its entry index in *failed_entry (may be NULL).
// entries [begin, end) of the plan (utx_plan_run = all of them).  A range must hold whole two-stream sections; sections are numbered by the JOINs in
// front of `begin`.  A launcher error inside a forked section still JOINS before returning: the caller's stream must not be left without its wait on the
// side stream (work already queued there would otherwise run unordered against whatever the caller launches next).
int utx_plan_run_range(utx_plan* p, int begin, int end, utx_stream stream_, int* failed_entry) {
    if (!p || p->open_sections || begin < 0 || end > (int)p->entries.size() || begin > end) return -2;
    hipStream_t main_s = (hipStream_t)stream_;
    size_t section = 0;
    int open_at_begin = 0;      // FORKs in front of `begin` that have not met their JOIN: `begin` may not lie inside a section, on either stream's half
    for (int i = 0; i < begin; ++i) {
        if (p->entries[i].kind == K_FORK) ++open_at_begin;
        if (p->entries[i].kind == K_JOIN) { ++section; --open_at_begin; }
    }
    if (open_at_begin != 0) { if (failed_entry) *failed_entry = begin; return -2; }
    bool forked = false;
    int rc = 0, bad = -1;
    for (int i = begin; i < end; ++i) {
        const Entry& e = p->entries[i];
        hipStream_t st = e.side ? p->side : main_s;
        if ((e.side || e.kind == K_JOIN) && !forked) { rc = -2; bad = i; break; }      // a side entry or a JOIN without its FORK in this call
        switch (e.kind) {
            case K_GEMM: rc = utx_launch_gemm_bf16(&e.gemm, st); break;
            case K_GEMV: rc = utx_launch_gemv_bf16(&e.gemv, st); break;
            case K_LN_MOD: rc = utx_launch_ln_mod(&e.ln, st); break;
            case K_QKV_POST: rc = utx_launch_qkv_post(&e.qkv, st); break;
            case K_ATTN: {
                const AttnArgs& a = e.attn;
                rc = utx_launch_attn_fwd(a.q, a.k, a.vt, a.o, a.q_hs, a.q_ss, a.k_hs, a.k_ss, a.vt_hs, a.vt_ds, a.o_ss, a.H, a.S_kv, a.S_q, a.scale, a.kb,
                                         a.period, a.work, a.work_bytes, st);
                break;
            }
            case K_QUANT: {
                const QuantArgs& a = e.quant;
                rc = a.packed ? utx_launch_quant_mx8_packed(a.x, a.ldx, a.q, a.ldq, a.s, a.lds, a.M, a.K, st)
                              : utx_launch_quant_mx8(a.x, a.ldx, a.q, a.ldq, a.s, a.lds, a.M, a.K, st);
                break;
            }
            case K_ADD3: rc = utx_launch_add3_bf16(e.add3.a, e.add3.b, e.add3.c, e.add3.out, e.add3.n, st); break;
            case K_QUANT_VT: rc = utx_launch_quant_vt_mx8(e.qvt.vt, e.qvt.v8, e.qvt.vs, e.qvt.H, e.qvt.S_pad, st); break;
            case K_ATTN8: rc = utx_launch_attn_fwd_fp8(&e.attn8, st); break;
            case K_FORK:
                if (hipEventRecord(p->events[2 * section], main_s) != hipSuccess || hipStreamWaitEvent(p->side, p->events[2 * section], 0) != hipSuccess) rc = -4;
                else forked = true;
                break;
            case K_JOIN:
                if (hipEventRecord(p->events[2 * section + 1], p->side) != hipSuccess || hipStreamWaitEvent(main_s, p->events[2 * section + 1], 0) != hipSuccess) rc = -4;
                forked = false;
                ++section;
                break;
        }
        if (rc != 0) { bad = i; break; }
    }
    if (rc == 0 && forked) { rc = -2; bad = end; }      // a range that ends inside a section
    if (forked) {      // error (or a cut section): join what was forked
        if (hipEventRecord(p->events[2 * section + 1], p->side) == hipSuccess) (void)hipStreamWaitEvent(main_s, p->events[2 * section + 1], 0);
    }
    if (rc != 0 && failed_entry) *failed_entry = bad;
    return rc;
}

int utx_plan_run(utx_plan* p, utx_stream stream_, int* failed_entry) {
    if (!p) return -2;
    return utx_plan_run_range(p, 0, (int)p->entries.size(), stream_, failed_entry);
}

}  // extern "C"
