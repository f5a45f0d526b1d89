"""The control flow of the opt-in peeled attention loop (attention_glds.hip, VAR 12 ... 15: UTX_ATTN_PEEL) against the general loop, on the CPU.

The peeled variants run the SAME tile body (a macro) -- what was written by hand is only WHICH tile goes through which copy, into which ring slot the next
tile is staged, and where the barriers sit.  This test lifts exactly those source lines out of the kernel (from the `if ((VAR >= 12 ...` that selects the
peeled loop to the end of the general loop), compiles them with g++ around stubs that record the events (stage(tile, slot) / body(tile, slot, special) /
barrier), and checks for every tile count and raggedness that the peeled loop issues the general loop's event sequence -- with the general body exactly on
the first tile and a ragged last tile, and with the general loop itself whenever key-multiplicity tiles recur."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "unitex_amd", "csrc", "attention_glds.hip")

HARNESS = r"""
#include <cstdio>
#include <vector>
#include <tuple>
struct P { float key_bias_log2; int key_bias_period; };
typedef std::vector<std::tuple<int, int, int, int>> Ev;      // (kind 0 stage / 1 body / 2 barrier, tile, slot, special)
#define AG_KVB 64
#define AG_STAGE(t_, slot_) ev.push_back({0, (int)(t_), (int)(slot_), 0})
#define AG_TILE_BODY(SP_) { ev.push_back({1, t, gs + sub, (SP_)}); }
#define __syncthreads() ev.push_back({2, 0, 0, 0})
template <int VAR, int TPB>
static Ev run(int Sk, P p) {
    Ev ev;
    const int nt = (Sk + AG_KVB - 1) / AG_KVB;
    const int ngrp = (nt + TPB - 1) / TPB;
    for (int i = 0; i < TPB; ++i)
        if (i < nt) AG_STAGE(i, i);
    __syncthreads();
%s
    return ev;
}
int main() {
    int bad = 0, checked = 0;
    for (int per = 0; per < 2; ++per)
    for (int Sk = 1; Sk <= 64 * 9; Sk += (Sk %% 64 == 0 ? 1 : 21)) {
        P p = {per ? 3.0f : (Sk %% 2 ? 0.0f : 3.0f), per ? 4 : 0};
        const int nt = (Sk + 63) / 64;
        const bool rag = Sk %% 64;
        Ev g = run<0, 1>(Sk, p);
        Ev vs[4] = {run<12, 1>(Sk, p), run<13, 1>(Sk, p), run<14, 1>(Sk, p), run<15, 1>(Sk, p)};
        for (auto& v : vs) {
            ++checked;
            if (v.size() != g.size()) { ++bad; printf("Sk %%d per %%d: %%zu vs %%zu events\n", Sk, per, v.size(), g.size()); continue; }
            for (size_t i = 0; i < g.size(); ++i) {
                auto [k0, t0, s0, sp0] = g[i];
                auto [k1, t1, s1, sp1] = v[i];
                bool ok = k0 == k1 && t0 == t1 && s0 == s1;
                if (k0 == 1) {      // which copy of the body: general on the first tile and on a ragged last tile -- everywhere when the key-multiplicity tiles recur
                    const bool want_special = (per && p.key_bias_log2 != 0.f) ? true : (t0 == 0 || (rag && t0 == nt - 1));
                    ok = ok && sp0 == 1 && (sp1 == 1) == want_special;
                }
                if (!ok) { ++bad; printf("Sk %%d per %%d event %%zu: (%%d %%d %%d %%d) vs (%%d %%d %%d %%d)\n", Sk, per, i, k0, t0, s0, sp0, k1, t1, s1, sp1); break; }
            }
        }
        // the staging discipline itself: a tile is staged exactly once, into the slot its body later reads, before that body, and never into the slot of the tile being read
        std::vector<int> slot(nt, -1);
        int reading = -1;
        for (auto [k, t, s, sp] : vs[0]) {
            if (k == 0) { if (slot[t] != -1 || (reading >= 0 && slot[reading] == s)) { ++bad; printf("Sk %%d: bad staging of tile %%d\n", Sk, t); } slot[t] = s; }
            if (k == 1) { if (slot[t] != s) { ++bad; printf("Sk %%d: body of tile %%d reads slot %%d, staged into %%d\n", Sk, t, s, slot[t]); } reading = t; }
            if (k == 2) reading = -1;
        }
    }
    printf("checked %%d bad %%d\n", checked, bad);
    return bad ? 1 : 0;
}
"""


def _loop_source():
    lines = open(SRC).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.lstrip().startswith("if ((VAR >= 12 && VAR <= 15) && TPB == 1"))
    end = next(i for i, l in enumerate(lines) if "this group fully read by every wave" in l)
    assert 0 < start < end and lines[end + 1].strip() == "}", "the loop block of attention_glds.hip moved: update this test's markers"
    block = [l for l in lines[start:end + 2] if not l.lstrip().startswith("#pragma")]
    return "\n".join(block)


def test_peeled_attention_loop_issues_the_general_loops_events(tmp_path):
    src = tmp_path / "skeleton.cpp"
    src.write_text(HARNESS % _loop_source())
    exe = tmp_path / "skeleton"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "bad 0" in r.stdout and "checked 0" not in r.stdout, r.stdout[-500:]
