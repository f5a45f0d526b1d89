#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of the 4 x 64 attention kernel:  python tools/gen_attn_q64.py  ->  unitex_amd/csrc/attention_q64_asm.inc

Why a generator: one wave per SIMD means program order IS issue order, and a 32x32x16 MFMA hides at most five single-issue instructions
(MI355X_MICROARCH.md, "one wave per SIMD").  hipcc would not keep a 512-register stream in the written order (round 5's kernel: all of a stage's
v_exp_f32 in front of its third MFMA, packed fp32 row sums, 5.9 issued instructions per MFMA gap -- profiles/r06_attn_pmc_arms_q64.log), so the whole
key loop is ONE asm statement whose private registers are named literally (and clobbered) and whose `s_waitcnt lgkmcnt(n)` are COUNTED here by
simulating the in-order LGKM queue.  The C++ around it (attention_q64.hip) keeps index arithmetic, the epilogue and the launch plan.

Pipeline (b = 32-key block, two per 64-key tile; h = 32-query half of the wave's 64 queries):
    stage b :  QK^T(b+1)  ||  softmax(b)  ||  PV(b-1)          32 MFMAs in eight groups  [QK h0, QK h1, PV h0, PV h1]
    fillers per group:  gap0  exp exp (+ LDS-DMA piece / scalar bookkeeping)     gap1  exp exp, K fragment read (two groups ahead)
                        gap2  add add cvt_pk, V^T fragment read (two groups ahead)    gap3  add add cvt_pk, the next group's counted lgkmcnt
One loop trip = one tile = stage 2t+1, the tile's barrier (+ vmcnt(0): the DMA issued a stage earlier), stage 2t+2 with the DMA of K(t+3) / V(t+2).
Three-slot LDS rings (K and V^T tiles of 16 KB); fragment addresses live in registers and are stepped once per tile right behind their last use.
The softmax is the sum-checked one of attention_glds.hip: scores leave the MFMA as s - m (the C operand is the -m block), m is the exact maximum of the
first block, a row sum beyond 2^40 marks the 64-query group for the repair pass.  Arithmetic order per element = the 8 x 32 kernel's.
"""
import os
import sys

# ---------------------------------------------------------------- private registers (named literally in the stream, listed as clobbers)
SA = {(0, 0): 192, (0, 1): 208, (1, 0): 224, (1, 1): 240}     # scores [block parity][half], 16 VGPRs each
NEGM = {0: 160, 1: 176}                                        # -m blocks [half], 16 VGPRs each
PB = {(0, 0): 128, (0, 1): 136, (1, 0): 144, (1, 1): 152}     # bf16 probabilities [block parity][half], 8 VGPRs (word i = elements 2i, 2i+1)
KF = [112, 116, 120, 124]                                      # K fragment window (4 VGPRs each)
VF = [96, 100, 104, 108]                                       # V^T fragment window
E = [88, 89, 90, 91]                                           # exponentials of the current group: h0 even, h0 odd, h1 even, h1 odd
SUM = {0: (84, 85, 86, 87), 1: (80, 81, 82, 83)}               # row-sum accumulators [parity]: h0 even, h0 odd, h1 even, h1 odd
PS0, PS1, T0, T1 = 78, 79, 76, 77
V_LO, V_HI = 76, 255
Q0, Q1 = 128, 160                                              # AGPR bases of the Q fragments of half 0 / 1 (8 x 4 each)
A_LO, A_HI = 128, 191
# scalars
S_KPTR, S_VPTR = 40, 42                                        # 64-bit: next K / V^T tile to request
S_T3, S_KD, S_VD, S_KSTEP, S_VSTEP, S_TMP, S_CNT = 44, 45, 46, 47, 48, 49, 50
S_POS, S_NEG, S_KD1, S_KD2, S_KDEND, S_VDEND = 51, 52, 53, 54, 55, 56
S_LO, S_HI = 40, 56
TILE = 16384


def v(n, w=1):
    return "v%d" % n if w == 1 else "v[%d:%d]" % (n, n + w - 1)


def a(n, w=1):
    return "a%d" % n if w == 1 else "a[%d:%d]" % (n, n + w - 1)


def s(n, w=1):
    return "s%d" % n if w == 1 else "s[%d:%d]" % (n, n + w - 1)


class Stream:
    def __init__(self):
        self.lines = []
        self.lgkm = []          # in-order queue of outstanding LDS reads (tags)
        self.n_mfma = 0
        self.gap = None         # instructions issued since the last MFMA (statistics)
        self.gaps = []

    def raw(self, text):
        self.lines.append(text)

    def ins(self, text):
        self.lines.append(text)
        if self.gap is not None:
            self.gap += 1

    def comment(self, text):
        self.lines.append("; " + text)

    def mfma(self, dst, a_, b_, c_):
        if self.gap is not None:
            self.gaps.append(self.gap)
        self.gap = 0
        self.n_mfma += 1
        self.lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, a_, b_, c_))

    def ds_read(self, tag, dst, addr, off=0):
        self.ins("ds_read_b128 %s, %s%s" % (dst, addr, (" offset:%d" % off) if off else ""))
        self.lgkm.append(tag)

    def wait_lgkm(self, tags):
        """every read in `tags` has returned: lgkmcnt(n) with n = the reads issued behind the youngest of them"""
        idx = [i for i, t in enumerate(self.lgkm) if t in tags]
        if not idx:
            return
        last = max(idx)
        n = len(self.lgkm) - 1 - last
        assert n <= 15
        self.ins("s_waitcnt lgkmcnt(%d)" % n)
        self.lgkm = self.lgkm[last + 1:]

    def wait_lgkm_all(self):
        self.ins("s_waitcnt lgkmcnt(0)")
        self.lgkm = []

    def text(self):
        return self.lines


def dma(st, voff_op, sbase, m0_base, m0_imm):
    """one 1 KB piece: M0 = LDS destination of the wave-instruction, then global_load_lds (saddr + per-lane 32-bit offset)"""
    st.ins("s_add_u32 m0, %s, %d" % (m0_base, m0_imm))
    st.ins("s_nop 0")
    st.ins("global_load_lds_dwordx4 %s, %s" % (voff_op, s(sbase, 2)))


def advance_ptr(st, ptr, step, cond_lhs, cond_rhs):
    """ptr += (cond_lhs < cond_rhs) ? step : 0   (unsigned; the last tile is requested again past the end: branch-free stages, no garbage in LDS)"""
    st.ins("s_cmp_lt_u32 %s, %s" % (cond_lhs, cond_rhs))
    st.ins("s_cselect_b32 %s, %s, 0" % (s(S_TMP), step))
    st.ins("s_add_u32 %s, %s, %s" % (s(ptr), s(ptr), s(S_TMP)))
    st.ins("s_addc_u32 %s, %s, 0" % (s(ptr + 1), s(ptr + 1)))


def gen_prologue(st):
    st.comment("---- scalars")
    st.ins("s_mov_b32 %s, %%[kptr_lo]" % s(S_KPTR))
    st.ins("s_mov_b32 %s, %%[kptr_hi]" % s(S_KPTR + 1))
    st.ins("s_mov_b32 %s, %%[vptr_lo]" % s(S_VPTR))
    st.ins("s_mov_b32 %s, %%[vptr_hi]" % s(S_VPTR + 1))
    st.ins("s_mov_b32 %s, 0x4000" % s(S_POS))
    st.ins("s_mov_b32 %s, 0xffff8000" % s(S_NEG))
    st.ins("s_add_u32 %s, %%[ldsk], 0x4000" % s(S_KD1))
    st.ins("s_add_u32 %s, %%[ldsk], 0x8000" % s(S_KD2))
    st.ins("s_add_u32 %s, %%[ldsk], 0xc000" % s(S_KDEND))
    st.ins("s_add_u32 %s, %%[ldsv], 0xc000" % s(S_VDEND))
    st.comment("---- Q fragments of both 32-query halves, straight into AGPRs (MFMA B operands only)")
    for h, (qp, base) in enumerate((("%[qp0]", Q0), ("%[qp1]", Q1))):
        for kk in range(8):
            st.ins("global_load_dwordx4 %s, %s, off offset:%d" % (a(base + 4 * kk, 4), qp, 32 * kk))
    st.comment("---- ring fill: K(0) V(0) K(1) V(1) K(2); tile indices clamp at nt - 1")
    for j in range(4):
        dma(st, "%%[ko%d]" % j, S_KPTR, "%[ldsk]", 1024 * j)
    for j in range(4):
        dma(st, "%%[vo%d]" % j, S_VPTR, "%[ldsv]", 1024 * j)
    st.ins("s_mov_b32 %s, 1" % s(S_T3))
    advance_ptr(st, S_KPTR, "%[kstride]", s(S_T3), "%[nt]")
    advance_ptr(st, S_VPTR, "0x80", s(S_T3), "%[nt]")
    for j in range(4):
        dma(st, "%%[ko%d]" % j, S_KPTR, "%[ldsk]", TILE + 1024 * j)
    for j in range(4):
        dma(st, "%%[vo%d]" % j, S_VPTR, "%[ldsv]", TILE + 1024 * j)
    st.ins("s_mov_b32 %s, 2" % s(S_T3))
    advance_ptr(st, S_KPTR, "%[kstride]", s(S_T3), "%[nt]")
    for j in range(4):
        dma(st, "%%[ko%d]" % j, S_KPTR, "%[ldsk]", 2 * TILE + 1024 * j)
    st.comment("state of 'trip -1': K dest slot 2, V dest slot 1, K pointer at tile min(2, nt-1), V pointer at tile min(1, nt-1), T3 = 2")
    st.ins("s_mov_b32 %s, %s" % (s(S_KD), s(S_KD2)))
    st.ins("s_add_u32 %s, %%[ldsv], 0x4000" % s(S_VD))
    st.ins("s_waitcnt vmcnt(0)")
    st.ins("s_barrier")
    st.comment("---- raw scores of blocks 0 and 1 (C = 0), K fragments four at a time")
    for blk in range(2):
        for quad in range(2):
            if blk or quad:
                st.ins("s_nop 7")
            for i in range(4):
                st.ds_read(("pk", blk, quad, i), v(KF[i], 4), "%%[kx%d]" % (4 * quad + i), 8192 * blk)
            st.wait_lgkm_all()
            for i in range(4):
                g = 4 * quad + i
                for h, qb in ((0, Q0), (1, Q1)):
                    st.mfma(v(SA[(blk, h)], 16), v(KF[i], 4), a(qb + 4 * g, 4), "0" if g == 0 else v(SA[(blk, h)], 16))
    st.gap = None
    for i in range(8):
        st.ins("v_add_u32 %%[kx%d], %s, %%[kx%d]" % (i, s(S_POS), i))      # slot 0 -> slot 1: every K fragment address now points at tile 1
    st.ins("s_nop 15")
    st.ins("s_nop 15")
    st.comment("---- key multiplicity of the first tile: scores += log2(multiplicity)")
    st.ins("s_cmp_eq_u32 %[kbv], 0")
    st.ins("s_cbranch_scc1 AQ2_NOKB0_%=")
    for h in range(2):
        for r in range(16):
            st.ins("v_add_f32 %s, %%[kbv], %s" % (v(SA[(0, h)] + r), v(SA[(0, h)] + r)))
    st.raw("AQ2_NOKB0_%=:")
    st.comment("---- exact step on block 0: m = row maximum, P(0) = 2^(s - m), -m block, block 1's scores shifted")
    for h in range(2):
        sa0, sa1 = SA[(0, h)], SA[(1, h)]
        st.ins("v_max3_f32 %s, %s, %s, %s" % (v(T0), v(sa0), v(sa0 + 1), v(sa0 + 2)))
        for r in range(3, 15, 2):
            st.ins("v_max3_f32 %s, %s, %s, %s" % (v(T0), v(T0), v(sa0 + r), v(sa0 + r + 1)))
        st.ins("v_max_f32 %s, %s, %s" % (v(T0), v(T0), v(sa0 + 15)))
        st.ins("ds_bpermute_b32 %s, %%[bperm], %s" % (v(T1), v(T0)))
        st.ins("s_waitcnt lgkmcnt(0)")
        st.ins("v_max_f32 %s, %s, %s" % (v(T0), v(T0), v(T1)))
        st.ins("v_mov_b32 %%[m%dr], %s" % (h, v(T0)))
        for r in range(16):
            st.ins("v_sub_f32 %s, %s, %s" % (v(sa0 + r), v(sa0 + r), v(T0)))
        for r in range(16):
            st.ins("v_exp_f32 %s, %s" % (v(sa0 + r), v(sa0 + r)))
        ev, od = SUM[0][2 * h], SUM[0][2 * h + 1]
        st.ins("v_mov_b32 %s, %s" % (v(ev), v(sa0)))
        st.ins("v_mov_b32 %s, %s" % (v(od), v(sa0 + 1)))
        for r in range(2, 16, 2):
            st.ins("v_add_f32 %s, %s, %s" % (v(ev), v(ev), v(sa0 + r)))
            st.ins("v_add_f32 %s, %s, %s" % (v(od), v(od), v(sa0 + r + 1)))
        for i in range(8):
            st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(0, h)] + i), v(sa0 + 2 * i), v(sa0 + 2 * i + 1)))
        for r in range(16):
            st.ins("v_xor_b32 %s, 0x80000000, %s" % (v(NEGM[h] + r), v(T0)))
        for r in range(16):
            st.ins("v_sub_f32 %s, %s, %s" % (v(sa1 + r), v(sa1 + r), v(T0)))
    st.ins("s_cmp_eq_u32 %[kbv], 0")
    st.ins("s_cbranch_scc1 AQ2_NOKB1_%=")
    for h in range(2):
        for r in range(16):
            st.ins("v_add_f32 %s, %%[kbv], %s" % (v(SA[(1, h)] + r), v(SA[(1, h)] + r)))
    st.raw("AQ2_NOKB1_%=:")
    st.ins("v_mov_b32 %[l0], 0")
    st.ins("v_mov_b32 %[l1], 0")
    st.ins("v_mov_b32 %[psmax], 0")
    st.comment("---- fragments of the first two groups of stage 1: K(tile 1, block 0)[0, 1], V(tile 0, slab 0)[db 0, 1]")
    st.ds_read(("K", 0), v(KF[0], 4), "%[kx0]", 0)
    st.ds_read(("V", 0), v(VF[0], 4), "%[vx0]", 0)
    st.ds_read(("K", 1), v(KF[1], 4), "%[kx1]", 0)
    st.ds_read(("V", 1), v(VF[1], 4), "%[vx0]", 4096)
    st.ins("s_mov_b32 %s, %%[nt]" % s(S_CNT))
    st.ins("s_nop 3")


def gen_stage(st, G0, odd, stats):
    """one stage of the loop; G0 = running index of its first group (read tags); odd: stage 2t+1 (softmax of a tile's block 1), else stage 2t+2"""
    p = 1 if odd else 0            # parity of the block whose softmax runs
    q = 1 - p                      # parity of the block QK^T writes / PV reads
    sums_p, sums_q = SUM[p], SUM[q]
    for g in range(8):
        G = G0 + g
        st.wait_lgkm([("K", G), ("V", G)])
        # ---- QK^T, half 0
        st.mfma(v(SA[(q, 0)], 16), v(KF[G % 4], 4), a(Q0 + 4 * g, 4), v(NEGM[0], 16) if g == 0 else v(SA[(q, 0)], 16))
        if odd:      # scalar bookkeeping of the trip, a few per gap (state of trip t from that of trip t - 1)
            if g == 0:
                st.ins("s_add_u32 %s, %s, 0x4000" % (s(S_KD), s(S_KD)))
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KDEND)))
                st.ins("s_cselect_b32 %s, %%[ldsk], %s" % (s(S_KD), s(S_KD)))
            elif g == 1:
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KD2)))
                st.ins("s_cselect_b32 %s, %s, %s" % (s(S_VSTEP), s(S_NEG), s(S_POS)))
            elif g == 2:
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KD1)))
                st.ins("s_cselect_b32 %s, %s, %s" % (s(S_KSTEP), s(S_NEG), s(S_POS)))
            elif g == 3:
                st.ins("s_add_u32 %s, %s, 0x4000" % (s(S_VD), s(S_VD)))
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_VD), s(S_VDEND)))
                st.ins("s_cselect_b32 %s, %%[ldsv], %s" % (s(S_VD), s(S_VD)))
            elif g == 4:
                st.ins("s_add_u32 %s, %s, 1" % (s(S_T3), s(S_T3)))
                st.ins("s_cmp_lt_u32 %s, %%[nt]" % s(S_T3))
                st.ins("s_cselect_b32 %s, %%[kstride], 0" % s(S_TMP))
            elif g == 5:
                st.ins("s_add_u32 %s, %s, %s" % (s(S_KPTR), s(S_KPTR), s(S_TMP)))
                st.ins("s_addc_u32 %s, %s, 0" % (s(S_KPTR + 1), s(S_KPTR + 1)))
            elif g == 6:
                st.ins("s_sub_u32 %s, %s, 1" % (s(S_TMP), s(S_T3)))
                st.ins("s_cmp_lt_u32 %s, %%[nt]" % s(S_TMP))
                st.ins("s_cselect_b32 %s, 0x80, 0" % s(S_TMP))
            else:
                st.ins("s_add_u32 %s, %s, %s" % (s(S_VPTR), s(S_VPTR), s(S_TMP)))
                st.ins("s_addc_u32 %s, %s, 0" % (s(S_VPTR + 1), s(S_VPTR + 1)))
        else:        # LDS-DMA piece g of the tile boundary: K(t+3) pieces 0-3, V(t+2) pieces 0-3; M0 first, the load behind the exponentials
            if g < 4:
                st.ins("s_add_u32 m0, %s, %d" % (s(S_KD), 1024 * g))
            else:
                st.ins("s_add_u32 m0, %s, %d" % (s(S_VD), 1024 * (g - 4)))
        ev0 = sums_p[0] if g == 0 else E[0]
        od0 = sums_p[1] if g == 0 else E[1]
        st.ins("v_exp_f32 %s, %s" % (v(ev0), v(SA[(p, 0)] + 2 * g)))
        st.ins("v_exp_f32 %s, %s" % (v(od0), v(SA[(p, 0)] + 2 * g + 1)))
        if not odd:
            if g < 4:
                st.ins("global_load_lds_dwordx4 %%[ko%d], %s" % (g, s(S_KPTR, 2)))
            else:
                st.ins("global_load_lds_dwordx4 %%[vo%d], %s" % (g - 4, s(S_VPTR, 2)))
        # ---- QK^T, half 1
        st.mfma(v(SA[(q, 1)], 16), v(KF[G % 4], 4), a(Q1 + 4 * g, 4), v(NEGM[1], 16) if g == 0 else v(SA[(q, 1)], 16))
        ev1 = sums_p[2] if g == 0 else E[2]
        od1 = sums_p[3] if g == 0 else E[3]
        st.ins("v_exp_f32 %s, %s" % (v(ev1), v(SA[(p, 1)] + 2 * g)))
        st.ins("v_exp_f32 %s, %s" % (v(od1), v(SA[(p, 1)] + 2 * g + 1)))
        # K fragment of group G + 2
        gg = g + 2
        if odd:      # QK^T(2t+2) = (tile t+1, block 0); groups 6, 7 fetch fragments 0, 1 of (tile t+1, block 1) and step those addresses to tile t+2
            if gg < 8:
                st.ds_read(("K", G + 2), v(KF[(G + 2) % 4], 4), "%%[kx%d]" % gg, 0)
            else:
                st.ds_read(("K", G + 2), v(KF[(G + 2) % 4], 4), "%%[kx%d]" % (gg - 8), 8192)
                st.ins("v_add_u32 %%[kx%d], %s, %%[kx%d]" % (gg - 8, s(S_KSTEP), gg - 8))
        else:        # QK^T(2t+3) = (tile t+1, block 1): fragments 2..7 then step; groups 6, 7 fetch fragments 0, 1 of (tile t+2, block 0)
            if gg < 8:
                st.ds_read(("K", G + 2), v(KF[(G + 2) % 4], 4), "%%[kx%d]" % gg, 8192)
                st.ins("v_add_u32 %%[kx%d], %s, %%[kx%d]" % (gg, s(S_KSTEP), gg))
            else:
                st.ds_read(("K", G + 2), v(KF[(G + 2) % 4], 4), "%%[kx%d]" % (gg - 8), 0)
        # ---- PV, half 0
        st.mfma("%%[o0%d]" % (g & 3), v(VF[G % 4], 4), v(PB[(q, 0)] + 4 * (g >> 2), 4), "%%[o0%d]" % (g & 3))
        if g == 0:   # the row sums of the previous stage's block are final: l += ps (block order), headroom record
            st.ins("v_add_f32 %s, %s, %s" % (v(PS0), v(sums_q[0]), v(sums_q[1])))
            st.ins("v_add_f32 %[l0], %[l0], " + v(PS0))
        else:
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[0]), v(sums_p[0]), v(E[0])))
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[1]), v(sums_p[1]), v(E[1])))
        st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(p, 0)] + g), v(ev0), v(od0)))
        # V^T fragment of group G + 2: (slab, db) of the stage it belongs to
        if odd:      # PV(2t) = tile t slabs 0, 1; groups 6, 7: slab 2 of the next stage
            if gg < 4:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx0]", 4096 * gg)
                if gg == 3:
                    st.ins("v_add_u32 %%[vx0], %s, %%[vx0]" % s(S_VSTEP))
            elif gg < 8:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx1]", 4096 * (gg - 4))
                if gg == 7:
                    st.ins("v_add_u32 %%[vx1], %s, %%[vx1]" % s(S_VSTEP))
            else:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx2]", 4096 * (gg - 8))
        else:        # PV(2t+1) = tile t slabs 2, 3; groups 6, 7: slab 0 of tile t+1
            if gg < 4:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx2]", 4096 * gg)
                if gg == 3:
                    st.ins("v_add_u32 %%[vx2], %s, %%[vx2]" % s(S_VSTEP))
            elif gg < 8:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx3]", 4096 * (gg - 4))
                if gg == 7:
                    st.ins("v_add_u32 %%[vx3], %s, %%[vx3]" % s(S_VSTEP))
            else:
                st.ds_read(("V", G + 2), v(VF[(G + 2) % 4], 4), "%[vx0]", 4096 * (gg - 8))
        # ---- PV, half 1
        st.mfma("%%[o1%d]" % (g & 3), v(VF[G % 4], 4), v(PB[(q, 1)] + 4 * (g >> 2), 4), "%%[o1%d]" % (g & 3))
        if g == 0:
            st.ins("v_add_f32 %s, %s, %s" % (v(PS1), v(sums_q[2]), v(sums_q[3])))
            st.ins("v_add_f32 %[l1], %[l1], " + v(PS1))
            st.ins("v_max3_f32 %%[psmax], %%[psmax], %s, %s" % (v(PS0), v(PS1)))
        else:
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[2]), v(sums_p[2]), v(E[2])))
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[3]), v(sums_p[3]), v(E[3])))
        st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(p, 1)] + g), v(ev1), v(od1)))


def gen():
    st = Stream()
    gen_prologue(st)
    entry = list(st.lgkm)
    assert entry == [("K", 0), ("V", 0), ("K", 1), ("V", 1)]
    st.raw("AQ2_LOOP_%=:")
    n0 = len(st.lines)
    st.gap = None
    gen_stage(st, 0, True, None)
    st.comment("---- tile boundary: this wave's pieces of K(t+2) / V(t+1) have landed (requested one stage ago); behind the barrier they are visible and the slots of K(t) / V(t-1) are free")
    st.ins("s_waitcnt vmcnt(0)")
    st.ins("s_barrier")
    gen_stage(st, 8, False, None)
    st.ins("s_sub_u32 %s, %s, 1" % (s(S_CNT), s(S_CNT)))
    st.ins("s_cmp_lg_u32 %s, 0" % s(S_CNT))
    st.ins("s_cbranch_scc1 AQ2_LOOP_%=")
    back = [(k, G - 16) for k, G in st.lgkm]
    assert back == entry, (back, entry)       # the LGKM queue at the back edge is the queue at the loop's entry: the counted waits hold on every trip
    loop_lines = [l for l in st.lines[n0:] if not l.startswith(";")]
    st.comment("---- drain: the last PV MFMAs, the prefetched fragments and the clamped re-requests")
    st.ins("s_nop 15")
    st.ins("s_nop 15")
    st.ins("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return st, loop_lines


OPERANDS_OUT = [("o%d%d" % (h, d), "+a", "oacc%d[%d]" % (h, d)) for h in range(2) for d in range(4)] + \
    [("l0", "=&v", "l_run0"), ("l1", "=&v", "l_run1"), ("m0r", "=&v", "m_run0"), ("m1r", "=&v", "m_run1"), ("psmax", "=&v", "ps_max")] + \
    [("kx%d" % i, "+v", "kx[%d]" % i) for i in range(8)] + [("vx%d" % i, "+v", "vx[%d]" % i) for i in range(4)]
OPERANDS_IN = [("ko%d" % i, "v", "ko[%d]" % i) for i in range(4)] + [("vo%d" % i, "v", "vo[%d]" % i) for i in range(4)] + \
    [("qp0", "v", "qp0"), ("qp1", "v", "qp1"), ("bperm", "v", "bperm"),
     ("kptr_lo", "s", "kptr_lo"), ("kptr_hi", "s", "kptr_hi"), ("vptr_lo", "s", "vptr_lo"), ("vptr_hi", "s", "vptr_hi"),
     ("kstride", "s", "kstride"), ("nt", "s", "nt"), ("ldsk", "s", "ldsk"), ("ldsv", "s", "ldsv"), ("kbv", "s", "kbv")]


def main():
    st, loop_lines = gen()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unitex_amd", "csrc", "attention_q64_asm.inc")
    n_mfma = sum(1 for l in loop_lines if l.startswith("v_mfma"))
    n_other = len(loop_lines) - n_mfma
    gaps = st.gaps
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_q64.py -- do not edit; the generator's header explains the stream.\n")
        f.write("// loop trip (one 64-key tile, 64 queries per wave): %d MFMAs, %d other instructions (%.2f per MFMA gap)\n" % (n_mfma, n_other, n_other / n_mfma))
        f.write("#define AQ2_ASM_TEXT \\\n")
        for l in st.text():
            f.write('    "%s\\n\\t" \\\n' % l.replace("\\", "\\\\").replace('"', '\\"'))
        f.write('    ""\n')
        f.write("#define AQ2_ASM_OUTPUTS " + ", ".join('[%s] "%s"(%s)' % o for o in OPERANDS_OUT) + "\n")
        f.write("#define AQ2_ASM_INPUTS " + ", ".join('[%s] "%s"(%s)' % o for o in OPERANDS_IN) + "\n")
        clob = ['"memory"', '"vcc"', '"scc"'] + ['"v%d"' % i for i in range(V_LO, V_HI + 1)] + ['"a%d"' % i for i in range(A_LO, A_HI + 1)] + ['"s%d"' % i for i in range(S_LO, S_HI + 1)]
        f.write("#define AQ2_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    # per-gap census of the loop (what the header of attention_q64.hip quotes)
    st2 = Stream()
    st2.lgkm = [("K", 0), ("V", 0), ("K", 1), ("V", 1)]
    gen_stage(st2, 0, True, None)
    st2.ins("s_waitcnt vmcnt(0)"); st2.ins("s_barrier")
    gen_stage(st2, 8, False, None)
    print("wrote %s: loop %d MFMAs + %d others = %.2f per gap; gap histogram %s" % (out, n_mfma, n_other, n_other / n_mfma,
          {k: st2.gaps.count(k) for k in sorted(set(st2.gaps))}))


if __name__ == "__main__":
    main()
