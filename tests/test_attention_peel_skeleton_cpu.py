"""The control flow of the attention kernels' fast loops (attention_glds.hip FAST, the default of the pre-scaled launch since round 5; attention_fp8.hip VAR 1) against
their general loops, on the CPU.

The fast loop runs the SAME tile arithmetic (macros) -- what was written by hand is WHICH tile goes through which copy, into which ring slot the next tile is
requested and when, where the barrier sits, and which ring slot the next tile's first K fragments are read from.  This test lifts exactly those source lines
out of the kernel (from the `if` that selects the fast loop to the end of the general loop), compiles them with g++ around stubs that record the events, and
checks for every tile count and raggedness
  * that both loops process tiles 0 .. nt - 1 once each, in order, with the general body exactly on the first tile and on a ragged last tile (and the general loop
    itself whenever key-multiplicity tiles recur);
  * the ring discipline of a workgroup whose waves are only ordered by its barriers: a slot is read only when the tile expected there was requested into it and a
    barrier (which carries the vmcnt(0) that retires the DMA) lies between request and read; a slot is requested into only when a barrier lies between the last read
    of its old content and the request, and that content has been consumed;
  * that the K fragments a fast tile computes on were read from that tile's slot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "unitex_amd", "csrc", "attention_glds.hip")

HARNESS = r"""
#include <cstdio>
#include <vector>
#include <tuple>
struct P { float key_bias_log2; int key_bias_period; };
typedef int bf16x8;
typedef std::vector<std::tuple<int, int, int, int>> Ev;
// kinds: 0 request(tile, slot) | 1 old-style body(tile, slot, general?) reads K and V of its slot | 2 barrier | 3 fast A(tile, slot) computes on the prefetched fragments,
//        reads the rest of K and V of its slot | 4 fast B: prefetch K fragments of (tile, slot) | 5 prefetch K fragments from (slot) in front of the loop
#define AG_KVB 64
constexpr int ABL = 0;      // the kernel's timing-ablation bits (ablation build only)
#define AG_STAGE(t_, slot_) ev.push_back({0, (int)(t_), (int)(slot_), 0})
#define AG_TILE_BODY { ev.push_back({1, t, gs + sub, 1}); }
#define __syncthreads() ev.push_back({2, 0, 0, 0})
#define AG_BARRIER() __syncthreads()
#define AG_FAST_A(KB_) ev.push_back({3, u, gs, (int)(KB_)});
#define AG_FAST_B(PF_) ev.push_back({4, u + 1, gs ^ 1, (PF_)});
#define AG_LOAD_KFA(slot_) ev.push_back({5, -1, (int)(slot_), 0});
template <bool FAST, int TPB, bool KBP = false>
static Ev run(int Sk, P p) {
    const int tb = 0;      // first tile of this workgroup's key range (a key-split tail workgroup starts further in)
    Ev ev;
    const int nt = (Sk + AG_KVB - 1) / AG_KVB;
    const int ngrp = (nt + TPB - 1) / TPB;
    for (int i = 0; i < TPB; ++i)
        if (i < nt) AG_STAGE(i, i);
    __syncthreads();
%s
    return ev;
}
static int bad = 0;
#define FAIL(...) do { ++bad; printf(__VA_ARGS__); printf("\n"); } while (0)
static void discipline(const Ev& ev, int Sk, int var, bool general_everywhere, int kb_period = 0) {
    const int nt = (Sk + 63) / 64;
    const bool rag = Sk %% 64;
    int content[2] = {-1, -1};
    bool requested[2] = {false, false}, read[2] = {false, false};
    std::vector<int> done(nt, 0);
    int next = 0, kfa_tile = -1;
    for (auto [k, t, s, sp] : ev) {
        if (k == 0) {
            if (t < 0 || t >= nt || s < 0 || s > 1) { FAIL("var %%d Sk %%d: request of tile %%d into slot %%d", var, Sk, t, s); continue; }
            if (read[s]) FAIL("var %%d Sk %%d: tile %%d requested into slot %%d while waves may still read it (no barrier since the last read)", var, Sk, t, s);
            if (content[s] >= 0 && !done[content[s]]) FAIL("var %%d Sk %%d: tile %%d overwrites tile %%d, which has not been processed", var, Sk, t, content[s]);
            content[s] = t; requested[s] = true;
        } else if (k == 2) {
            requested[0] = requested[1] = read[0] = read[1] = false;
        } else if (k == 1 || k == 3) {
            if (t != next) FAIL("var %%d Sk %%d: tile %%d processed, expected %%d", var, Sk, t, next);
            if (content[s] != t || requested[s]) FAIL("var %%d Sk %%d: tile %%d reads slot %%d (holds %%d, requested since the last barrier: %%d)", var, Sk, t, s, content[s], (int)requested[s]);
            if (k == 3 && (sp != 0) != (kb_period > 0 && t %% kb_period == 0)) FAIL("var %%d Sk %%d: tile %%d takes the %%s copy of the fast tile", var, Sk, t, sp ? "key-multiplicity" : "plain");
            if (k == 3 && kfa_tile != t) FAIL("var %%d Sk %%d: fast tile %%d computes on the K fragments of tile %%d", var, Sk, t, kfa_tile);
            if (k == 1) {
                const bool want = general_everywhere || t == 0 || (rag && t == nt - 1);
                if ((sp == 1) != want) FAIL("var %%d Sk %%d: tile %%d takes the %%s body", var, Sk, t, sp ? "general" : "fast");
            } else if (general_everywhere || t == 0 || (rag && t == nt - 1)) FAIL("var %%d Sk %%d: tile %%d takes the fast form", var, Sk, t);
            read[s] = true; if (t >= 0 && t < nt) done[t] = 1; ++next;
        } else {      // 4 / 5: K fragment prefetch from slot s
            if (requested[s]) FAIL("var %%d Sk %%d: K fragments read from slot %%d behind a request without a barrier", var, Sk, s);
            read[s] = true;
            kfa_tile = content[s];
        }
    }
    if (next != nt) FAIL("var %%d Sk %%d: %%d of %%d tiles processed", var, Sk, next, nt);
}
int main() {
    int checked = 0;
    for (int per : {0, 4, 3, 1, 7})      // key-multiplicity period in tiles (0: none): even and odd periods enter a run of ordinary tiles on odd and on even tiles
    for (int Sk = 1; Sk <= 64 * 9; Sk += (Sk %% 64 == 0 ? 1 : 21)) {
        P p = {per ? 3.0f : (Sk %% 2 ? 0.0f : 3.0f), per};
        const bool general_everywhere = per && p.key_bias_log2 != 0.f;
        Ev g = run<false, 1>(Sk, p);
        discipline(g, Sk, 0, true);
        // the launcher: periodic key multiplicity -> the KBP instance of the fast kernel (key-multiplicity tiles through their own copy of the fast tile), else the plain one
        Ev v = general_everywhere ? run<true, 1, true>(Sk, p) : run<true, 1, false>(Sk, p);
        ++checked;
        discipline(v, Sk, 1, false, general_everywhere ? p.key_bias_period : 0);
    }
    printf("checked %%d bad %%d\n", checked, bad);
    return bad ? 1 : 0;
}
"""


def _loop_source():
    lines = open(SRC).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.lstrip().startswith("if (FAST && TPB == 1) {"))
    end = next(i for i, l in enumerate(lines) if "this group fully read by every wave" in l)
    assert 0 < start < end and lines[end + 1].strip() == "}", "the loop block of attention_glds.hip moved: update this test's markers"
    block = [l for l in lines[start:end + 2] if not l.lstrip().startswith("#pragma")]
    assert any("AG_FAST_A" in l for l in block) and any("AG_FAST_B(1)" in l for l in block)
    return "\n".join(block)


def _run(tmp_path, code, name="skeleton"):
    src = tmp_path / (name + ".cpp")
    src.write_text(HARNESS % code)
    exe = tmp_path / name
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)


def test_fast_attention_loop_keeps_the_tile_order_and_the_ring_discipline(tmp_path):
    r = _run(tmp_path, _loop_source())
    assert r.returncode == 0 and "bad 0" in r.stdout and "checked 0" not in r.stdout, r.stdout[-3000:]


def test_the_skeleton_check_sees_a_broken_loop(tmp_path):
    """mutations of the lifted source that each describe a real bug (wrong slot, request in front of the barrier, prefetch from the slot being overwritten, a ragged
    last tile on the fast path, a dropped barrier) must all be reported"""
    src = _loop_source()
    mutations = [("const int fast_end_ = rag_ ? nt - 1 : nt;                 // tiles [1, fast_end_) take the fast form", "const int fast_end_ = nt;"),
                 ("            if (1 < nt) AG_STAGE(1, 1);\n", "            if (1 < nt) AG_STAGE(1, 0);\n"),
                 ("if (u + 2 < nt) AG_STAGE(u + 2, gs); ", "if (u + 2 < nt) AG_STAGE(u + 2, gs ^ 1);"),
                 ("            AG_BARRIER();                                      /* tile u + 1 has landed", "            /* tile u + 1 has landed"),
                 ("                AG_FAST_TILE_(uu + 1, 0, false)\n", "                AG_FAST_TILE_(uu + 1, 1, false)\n"),
                 ("            if (uu < fast_end_) AG_FAST_TILE_(uu, 1, false)\n", ""),
                 ("                if (uu < fast_end_) { AG_FAST_TILE_(uu, uu & 1, true) ++uu; }\n", "                if (uu < fast_end_) { AG_FAST_TILE_(uu, uu & 1, false) ++uu; }\n"),
                 ("                if (uu < nb && !(uu & 1)) { AG_FAST_TILE_(uu, 0, false) ++uu; }", "                if (uu < nb && !(uu & 1)) { AG_FAST_TILE_(uu, 1, false) ++uu; }"),
                 ("                if (uu < nb) { AG_FAST_TILE_(uu, 1, false) ++uu; }\n", "                if (uu < nb) { AG_FAST_TILE_(uu, 0, false) ++uu; }\n"),
                 ("        if (2 < nt) AG_STAGE(2, 0);", "        if (2 < nt) AG_STAGE(2, 1);"),
                 ("        if (1 < fast_end_) AG_LOAD_KFA(1)", "        if (1 < fast_end_) AG_LOAD_KFA(0)")]
    for i, (a, b) in enumerate(mutations):
        assert src.count(a) >= 1, a
        r = _run(tmp_path, src.replace(a, b), "mut%d" % i)
        assert r.returncode == 1 and "bad 0" not in r.stdout, "mutation %d (%s -> %s) went unnoticed" % (i, a, b)


FP8_SRC = os.path.join(os.path.dirname(HERE), "unitex_amd", "csrc", "attention_fp8.hip")
FP8_HARNESS = r"""
#include <cstdio>
#include <vector>
#include <utility>
struct P { float key_bias_log2; int key_bias_period; };
#define A8_KVB 64
#define A8_TILE_BODY(SP_) { ev.push_back({t, 1}); }
#define A8_FAST_BODY { ev.push_back({t, 0}); }
template <int VAR>
static std::vector<std::pair<int, int>> run(int S, P p) {
    std::vector<std::pair<int, int>> ev;
    const int nt = (S + A8_KVB - 1) / A8_KVB;
%s
    return ev;
}
int main() {
    int bad = 0, checked = 0;
    for (int per = 0; per < 2; ++per)
    for (int S = 1; S <= 64 * 7; S += (S %% 64 == 0 ? 1 : 21)) {
        P p = {per ? 3.0f : (S %% 2 ? 0.0f : 3.0f), per ? 4 : 0};
        const int nt = (S + 63) / 64;
        const bool rag = S %% 64, general_everywhere = per && p.key_bias_log2 != 0.f;
        auto g = run<0>(S, p), v = run<1>(S, p);
        ++checked;
        if ((int)g.size() != nt || (int)v.size() != nt) { ++bad; printf("S %%d: %%zu / %%zu tiles of %%d\n", S, g.size(), v.size(), nt); continue; }
        for (int t = 0; t < nt; ++t) {
            const bool want_general = general_everywhere || t == 0 || (rag && t == nt - 1);
            if (g[t].first != t || g[t].second != 1 || v[t].first != t || (v[t].second == 1) != want_general) { ++bad; printf("S %%d per %%d tile %%d: general loop (%%d, %%d), variant (%%d, %%d)\n", S, per, t, g[t].first, g[t].second, v[t].first, v[t].second); break; }
        }
    }
    printf("checked %%d bad %%d\n", checked, bad);
    return bad ? 1 : 0;
}
"""


def test_peeled_fp8_attention_loop_takes_every_tile_once_through_the_right_body(tmp_path):
    """attention_fp8.hip, VAR 1 (UTX_ATTN8_PEEL): the loop selection lifted from the source -- tiles 0 .. nt - 1 in order, the general body exactly on tile 0 and on a ragged
    last tile, the general loop whenever key-multiplicity tiles recur.  (Requests, ring slots and the barrier sit INSIDE both bodies, copied line for line.)"""
    lines = open(FP8_SRC).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.lstrip().startswith("if (VAR == 1 && !(p.key_bias_period > 0"))
    end = next(i for i, l in enumerate(lines) if l.strip() == "for (int t = 0; t < nt; ++t) A8_TILE_BODY(1)")
    assert 0 < start < end < start + 12
    src = tmp_path / "fp8_skeleton.cpp"
    src.write_text(FP8_HARNESS % "\n".join(lines[start:end + 1]))
    exe = tmp_path / "fp8_skeleton"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "bad 0" in r.stdout and "checked 0" not in r.stdout, r.stdout[-2000:]
    # the fast body's request / fence lines are the general body's, character for character
    text = open(FP8_SRC).read()
    gen = text[text.index("#define A8_TILE_BODY(SP_)"):text.index("#define A8_EXPQ")]
    fast = text[text.index("#define A8_FAST_BODY"):text.index("// VAR 1 (the default since round 5")]
    norm = lambda s: " ".join(s.replace("\\", " ").split())
    for must in ("A8_STAGE(t + 1, slot ^ 1);", "ksc0n = ksb[kr]; ksc1n = ksb[kr + 32];", "vscn = vsb[(long)(2 * (t + 1) + lh) * 32 + lq];", "const int slot = t & 1;",
                 'asm volatile("" : "+v"(ksc0n), "+v"(ksc1n), "+v"(vscn));', "ksc0 = (int)(ksc0n >> (8 * lh)); ksc1 = (int)(ksc1n >> (8 * lh)); vsc = (int)vscn;", "A8_BARRIER();",
                 "vsc_pv = vsc;", "l_run += ps0 + ps1;", "pb = a8_i32x8{pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7};"):
        assert norm(must) in norm(gen) and norm(must) in norm(fast), must


EXP_HARNESS = r"""
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <random>
static float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
// stand-in for the packing instruction: any injective-looking function of (a, b, old, which half) pins WHICH values go WHERE in which order
static int __builtin_amdgcn_cvt_pk_fp8_f32(float a, float b, int old, bool hi) {
    uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    const uint32_t h = ((ua * 2654435761u) ^ (ub * 40503u + 0x9e3779b9u)) & 0xffffu;
    return hi ? (int)(((uint32_t)old & 0xffffu) | (h << 16)) : (int)(((uint32_t)old & 0xffff0000u) | h);
}
#define _Pragma(x)
%s
int main() {
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-12.f, 4.f);
    int bad = 0;
    for (int it = 0; it < 2000; ++it) {
        float sa[16];
        for (int r = 0; r < 16; ++r) sa[r] = d(g);
        int a0, a1, a2, a3, b0, b1, b2, b3; float psa, psb;
        A8_EXPB(sa, a0, a1, a2, a3, psa)
        { float sc0_ = 0.f, sc1_ = 0.f; A8_EXPQ(sa, 0, b0, b1) A8_EXPQ(sa, 8, b2, b3) psb = sc0_ + sc1_; }
        uint32_t x, y; memcpy(&x, &psa, 4); memcpy(&y, &psb, 4);
        if (x != y || a0 != b0 || a1 != b1 || a2 != b2 || a3 != b3) ++bad;
    }
    printf("bad %%d\n", bad);
    return bad ? 1 : 0;
}
"""


def test_exponential_quarters_of_the_fp8_fast_body_equal_the_whole_block(tmp_path):
    """A8_EXPQ twice (running sums carried across) against A8_EXPB, both macro texts lifted from attention_fp8.hip and compiled on the host: the same row-sum bits
    (same summation order) and the same values in the same packing slots."""
    text = open(FP8_SRC).read()

    def macro(name):
        i = text.index("#define " + name + "(")
        out = []
        for l in text[i:].split("\n"):
            out.append(l)
            if not l.rstrip().endswith("\\"):
                break
        return "\n".join(out).replace('_Pragma("unroll")', "")
    src = tmp_path / "expq.cpp"
    src.write_text(EXP_HARNESS % (macro("A8_EXPB") + "\n" + macro("A8_EXPQ")))
    exe = tmp_path / "expq"
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-ffp-contract=off", "-w", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-500:]
