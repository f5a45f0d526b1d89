#!/usr/bin/env python3
"""Generator of the steady-state K loop of gemm256_w4_kernel (one wave per SIMD, 256 x 256 tile):  python tools/gen_gemm_w4_loop.py  ->  unitex_amd/csrc/gemm_w4_loop_asm.inc

Why: the kernel is power-bound like the attention kernel (time = energy / cap: DESIGN 8 Round 6), and hipcc's K-tile costs 250 instructions beside its 64 MFMAs -- 32 fragment
reads and 16 LDS-DMA pieces that have to be there, and ~200 that do not: 28 waits, 18 v_readlane (SGPR spills of the cursor state), 18 v_add_u32 (LDS addresses re-formed from a
run-time stage base), the if-converted tile-origin divisions of the staging cursor, 16 s_mov m0 ...  (listing census: profiles/r06_gemm_w4_ktile_census.log).  In the steady state
of a K segment none of that state changes: two K-tiles per trip (stage slots 0 / 1 as literals), every LDS address = loop-invariant register + immediate, the two operand
pointers step by 128 bytes -- 74 instructions beside the 64 MFMAs.  The C++ loop keeps every boundary (tile / segment / LoRA switch / split-tail range / parking): the stream runs only
while BOTH cursors stay inside their segments (attention of gemm_w4.hip: W4_FAST_TRIPS).

The schedule is the C++ K-tile's, slot for slot (gemm_w4.hip header): K-step ks = 16 MFMAs (jn, im) = (i >> 2, i & 3) on the fragment set F0 / F1; behind MFMA i < 8 one fragment
read of the NEXT K-step (order B0 A0 A1 B1 A2 A3 B2 B3); LDS-DMA piece p of the staging cursor's K-tile behind MFMAs 0, 3, .. 15 of K-step 3 (p 0..5), 2, 5, .. 14 of K-step 0
(p 6..10), 1, 4, .. 13 of K-step 1 (p 11..15), M0 written in front of the MFMA before it; lgkmcnt(0) behind every K-step, vmcnt(0) + barrier behind K-step 2.  Same MFMAs on the same
operands in the same order per accumulator: bit-identical to the C++ loop.
"""
import os

F1A, F1B = 224, 240            # private fragment set F1: A fragments v[224:239], B fragments v[240:255]
S_PA, S_PB, S_CNT = 88, 90, 92
STAGE = 65536


def piece_of(ks, i):
    if ks == 3:
        return i // 3 if i % 3 == 0 else -1
    if ks == 0:
        return 6 + i // 3 if i % 3 == 2 else -1
    if ks == 1:
        return 11 + i // 3 if (i % 3 == 1 and i < 15) else -1
    return -1


def frag(which, idx, f1):
    """operand text of fragment idx (0..3) of operand `which` ('a' | 'b') in set F1 (private registers) or F0 (asm operands)"""
    if f1:
        base = (F1A if which == "a" else F1B) + 4 * idx
        return "v[%d:%d]" % (base, base + 3)
    return "%%[f%s%d]" % (which, idx)


READ_ORDER = [("b", 0), ("a", 0), ("a", 1), ("b", 1), ("a", 2), ("a", 3), ("b", 2), ("b", 3)]


def kstep(out, ks, c, cur_f1, read_slot, read_kk, dma_slot):
    """ks 0..3 of the K-tile in stage slot c; MFMAs on set cur_f1; reads -> the other set from stage read_slot at k-chunk read_kk; DMA pieces into stage dma_slot"""
    for i in range(16):
        jn, im = i >> 2, i & 3
        p = piece_of(ks, i)
        if p >= 0:
            isb, d = (1, p - 8) if p >= 8 else (0, p)
            out.append("s_add_u32 m0, %%[ldsdma], %d" % (dma_slot * STAGE + isb * 32768 + d * 4096))
        out.append("v_mfma_f32_32x32x16_bf16 %%[acc%d%d], %s, %s, %%[acc%d%d]" % (jn, im, frag("b", jn, cur_f1), frag("a", im, cur_f1), jn, im))
        if i < 8:
            w, idx = READ_ORDER[i]
            out.append("ds_read_b128 %s, %%[r%s%ds%d] offset:%d" % (frag(w, idx, not cur_f1), w, read_kk, read_slot, 4096 * idx))
        if p >= 0:
            out.append("global_load_lds_dwordx4 %%[vo%s%d], s[%d:%d]" % ("B" if isb else "A", d, S_PB if isb else S_PA, (S_PB if isb else S_PA) + 1))
    if ks == 1:      # the staging cursor advances one K-tile: both operand pointers + 128 bytes
        out.append("s_add_u32 s%d, s%d, 0x80" % (S_PA, S_PA))
        out.append("s_addc_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1))
        out.append("s_add_u32 s%d, s%d, 0x80" % (S_PB, S_PB))
        out.append("s_addc_u32 s%d, s%d, 0" % (S_PB + 1, S_PB + 1))
    out.append("s_waitcnt lgkmcnt(0)")
    if ks == 2:
        out.append("s_waitcnt vmcnt(0)")
        out.append("s_barrier")


def ktile(out, c):
    kstep(out, 0, c, False, c, 1, c ^ 1)
    kstep(out, 1, c, True, c, 2, c ^ 1)
    kstep(out, 2, c, False, c, 3, c ^ 1)
    kstep(out, 3, c, True, c ^ 1, 0, c)      # behind the barrier: pieces 0..5 of K-tile + 2 into THIS stage; F0 <- K-step 0 of the next stage


def gen():
    out = []
    out.append("s_mov_b32 s%d, %%[pa_lo]" % S_PA)
    out.append("s_mov_b32 s%d, %%[pa_hi]" % (S_PA + 1))
    out.append("s_mov_b32 s%d, %%[pb_lo]" % S_PB)
    out.append("s_mov_b32 s%d, %%[pb_hi]" % (S_PB + 1))
    out.append("s_mov_b32 s%d, %%[trips]" % S_CNT)
    out.append("W4F_LOOP_%=:")
    ktile(out, 0)
    ktile(out, 1)
    out.append("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
    out.append("s_cmp_lg_u32 s%d, 0" % S_CNT)
    out.append("s_cbranch_scc1 W4F_LOOP_%=")
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "unitex_amd", "csrc", "gemm_w4_loop_asm.inc")
    lines = gen()
    body = [l for l in lines if not l.endswith(":")]
    n_mfma = sum(1 for l in body if l.startswith("v_mfma"))
    outs = ['[acc%d%d] "+a"(acc[%d][%d])' % (j, i, j, i) for j in range(4) for i in range(4)] + \
           ['[fa%d] "+v"(fa0[%d])' % (i, i) for i in range(4)] + ['[fb%d] "+v"(fb0[%d])' % (i, i) for i in range(4)]
    ins = ['[r%s%ds%d] "v"(w4f_r%s[%d][%d])' % (w, k, sl, w, sl, k) for w in "ab" for sl in range(2) for k in range(4)] + \
          ['[voA%d] "v"(voA%d)' % (d, d) for d in range(8)] + ['[voB%d] "v"(voB%d)' % (d, d) for d in range(8)] + \
          ['[pa_lo] "s"(w4f_pa_lo)', '[pa_hi] "s"(w4f_pa_hi)', '[pb_lo] "s"(w4f_pb_lo)', '[pb_hi] "s"(w4f_pb_hi)', '[trips] "s"(w4f_trips)', '[ldsdma] "s"(w4f_ldsdma)']
    clob = ['"memory"', '"scc"'] + ['"v%d"' % r for r in range(F1A, F1B + 16)] + ['"s%d"' % r for r in range(S_PA, S_CNT + 1)]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4_loop.py -- do not edit; the generator's header explains the stream.\n")
        f.write("// one trip = two K-tiles: %d MFMAs + %d other instructions (%.2f per MFMA gap)\n" % (n_mfma, len(body) - n_mfma - 8, (len(body) - n_mfma - 8) / n_mfma))
        f.write("#define W4F_ASM_TEXT \\\n")
        for l in lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define W4F_ASM_OUTPUTS " + ", ".join(outs) + "\n")
        f.write("#define W4F_ASM_INPUTS " + ", ".join(ins) + "\n")
        f.write("#define W4F_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    print("wrote %s: trip of two K-tiles = %d MFMAs + %d others" % (path, n_mfma, len(body) - n_mfma - 8))


if __name__ == "__main__":
    main()
