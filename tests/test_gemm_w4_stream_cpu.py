"""The generated steady-state K loop of gemm256_w4_kernel (unitex_amd/csrc/gemm_w4_loop_asm.inc, tools/gen_gemm_w4_loop.py) audited on the CPU from the EMITTED text:

  * the committed .inc is the generator's output;
  * one trip = two K-tiles (stage slot 0, then 1): per K-tile and accumulator (jn, im) exactly four MFMAs, K-steps 0..3 in order, A operand = the B-fragment jn / B operand = the A-fragment
    im of the fragment set the K-step computes on (F0, F1, F0, F1) -- the C++ loop's MFMAs in its order (bit identity on the GPU: tests/test_fullsize_gpu.py, tools/gemm_fastk_check.py);
  * a fragment set is read by MFMAs only after a `s_waitcnt lgkmcnt(0)` that follows the eight ds_reads which filled it, and never overwritten while the current K-step still computes on it;
    the reads walk k-chunks 1, 2, 3 of the computing stage and chunk 0 of the NEXT stage, fragment i at +4096 i;
  * LDS-DMA: each of the 16 pieces of a K-tile (operand A / B x 8) exactly once per K-tile, M0 = wave base + stage * 65536 + operand * 32768 + piece * 4096 written in front of the MFMA before
    it, pieces 6..15 into the OTHER stage (K-steps 0, 1), pieces 0..5 into the computing stage only BEHIND the barrier that follows K-step 2's reads of it and its vmcnt(0); both operand
    pointers step 128 bytes between piece 15 and piece 0."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gemm_w4_loop as G  # noqa: E402


def test_committed_stream_is_the_generators_output():
    text = open(os.path.join(ROOT, "unitex_amd", "csrc", "gemm_w4_loop_asm.inc")).read()
    emitted = [m.group(1) for m in re.finditer(r'^    "(.*)\\n\\t" \\$', text, re.M)]
    assert emitted == G.gen(), "gemm_w4_loop_asm.inc is stale: run python tools/gen_gemm_w4_loop.py"


def _fragset(tok):
    """'%[fb2]' -> ('F0', 'b', 2); 'v[228:231]' -> ('F1', 'a', 1)"""
    m = re.fullmatch(r"%\[f([ab])(\d)\]", tok)
    if m:
        return "F0", m.group(1), int(m.group(2))
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    base = int(m.group(1))
    if G.F1A <= base < G.F1A + 16:
        return "F1", "a", (base - G.F1A) // 4
    assert G.F1B <= base < G.F1B + 16
    return "F1", "b", (base - G.F1B) // 4


def test_trip_structure_mfma_order_fragment_discipline_and_dma_placement():
    lines = G.gen()
    i0 = lines.index("W4F_LOOP_%=:") + 1
    body = lines[i0:]
    assert body[-3:] == ["s_sub_u32 s%d, s%d, 1" % (G.S_CNT, G.S_CNT), "s_cmp_lg_u32 s%d, 0" % G.S_CNT, "s_cbranch_scc1 W4F_LOOP_%="]
    body = body[:-3]
    mf = [k for k, l in enumerate(body) if l.startswith("v_mfma")]
    assert len(mf) == 128
    # state of the walk
    filled = {"F0": True, "F1": False}          # F0 arrives filled (the C++ loop's K-step 3 / the previous trip's)
    pending = {}                                 # set -> number of reads issued since it was last waited for
    ptr_steps = 0
    barrier_seen_in_tile = False
    cut = [j for j in range(mf[63], len(body)) if body[j] == "s_waitcnt lgkmcnt(0)"][0] + 1      # a K-tile ends with the wait behind its 64th MFMA's trailing read / DMA
    tiles = [body[:cut], body[cut:]]
    for kt in range(2):
        c = kt
        seg = tiles[kt]
        pieces, m0, n_mfma = [], None, 0
        barrier_seen_in_tile = False
        acc_steps = {}
        reads_this_step = []
        for l in seg:
            op, _, rest = l.partition(" ")
            a = [x.strip() for x in rest.split(",")]
            ks = min(n_mfma // 16, 3) if op != "v_mfma_f32_32x32x16_bf16" else n_mfma // 16
            if op == "v_mfma_f32_32x32x16_bf16":
                i = n_mfma % 16
                jn, im = i >> 2, i & 3
                assert a[0] == a[3] == "%%[acc%d%d]" % (jn, im)
                cur = "F0" if ks % 2 == 0 else "F1"
                assert _fragset(a[1]) == (cur, "b", jn) and _fragset(a[2]) == (cur, "a", im), (kt, ks, i, l)
                assert filled[cur], "K-tile %d K-step %d computes on a fragment set no wait has covered" % (kt, ks)
                acc_steps.setdefault((jn, im), []).append(ks)
                n_mfma += 1
            elif op == "ds_read_b128":
                ks = (n_mfma - 1) // 16
                dst = _fragset(a[0])
                cur = "F0" if ks % 2 == 0 else "F1"
                assert dst[0] != cur, "a read lands in the set the K-step computes on"
                m = re.fullmatch(r"%\[r([ab])(\d)s(\d)\] offset:(\d+)", a[1])
                w, kk, slot, off = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
                assert w == dst[1] and off == 4096 * dst[2]
                assert (kk, slot) == ((ks + 1) % 4, c if ks < 3 else c ^ 1), "K-step %d reads chunk %d of stage %d" % (ks, kk, slot)
                if ks == 2:
                    assert not barrier_seen_in_tile
                filled[dst[0]] = False
                pending[dst[0]] = pending.get(dst[0], 0) + 1
                reads_this_step.append((dst[1], dst[2]))
            elif l == "s_waitcnt lgkmcnt(0)":
                for st, n in list(pending.items()):
                    assert n == 8, "a K-step fills its next set with %d reads" % n
                    filled[st] = True
                assert sorted(reads_this_step) == sorted([(w_, i_) for w_ in "ab" for i_ in range(4)]) and reads_this_step == G.READ_ORDER
                pending, reads_this_step = {}, []
            elif op == "s_add_u32" and a[0] == "m0":
                assert a[1] == "%[ldsdma]"
                m0 = int(a[2])
            elif op == "global_load_lds_dwordx4":
                ks = (n_mfma - 1) // 16
                m = re.fullmatch(r"%\[vo([AB])(\d)\]", a[0])
                isb, d = int(m.group(1) == "B"), int(m.group(2))
                p = 8 * isb + d
                assert a[1] == "s[%d:%d]" % ((G.S_PB, G.S_PB + 1) if isb else (G.S_PA, G.S_PA + 1))
                assert G.piece_of(ks, (n_mfma - 1) % 16) == p
                stage = (c ^ 1) if ks < 3 else c
                assert m0 == stage * G.STAGE + isb * 32768 + d * 4096, "piece %d of K-step %d lands at %s" % (p, ks, m0)
                if ks == 3:
                    assert barrier_seen_in_tile, "the computing stage is overwritten before the barrier behind its last reads"
                    assert ptr_steps == 2 * (kt + 1), "pieces 0..5 belong to the NEXT K-tile of the cursor: the pointers must have stepped"
                else:
                    assert ptr_steps == 2 * kt
                pieces.append(p)
                m0 = None                     # one M0 write per piece
            elif op in ("s_add_u32", "s_addc_u32"):
                if op == "s_add_u32":
                    assert a == ["s%d" % r for r in ((G.S_PA,) * 2 if a[0] == "s%d" % G.S_PA else (G.S_PB,) * 2)] + ["0x80"]
                    ptr_steps += 1
                    assert (n_mfma - 1) // 16 == 1 and set(pieces) >= set(range(6, 16)), "the cursor advances behind its K-tile's last piece"
            elif l == "s_waitcnt vmcnt(0)":
                assert (n_mfma - 1) // 16 == 2
            elif l == "s_barrier":
                assert (n_mfma - 1) // 16 == 2 and not pending, "the barrier stands behind K-step 2's reads AND their wait"
                barrier_seen_in_tile = True
            else:
                raise AssertionError("unexpected instruction in the stream: " + l)
        assert n_mfma == 64 and sorted(pieces) == list(range(16)) and pieces[:5] == [6, 7, 8, 9, 10] and pieces[10:] == [0, 1, 2, 3, 4, 5]
        assert all(v == [0, 1, 2, 3] for v in acc_steps.values()) and len(acc_steps) == 16
    assert filled["F0"], "the trip hands F0 = K-step 0 of the next stage back to the loop"
