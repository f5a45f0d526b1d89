#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of the 4 x 64 attention kernel:  python tools/gen_attn_q64.py  ->  unitex_amd/csrc/attention_q64_asm.inc

Why a generator: one wave per SIMD means program order IS issue order, and a 32x32x16 MFMA hides at most five single-issue instructions
(MI355X_MICROARCH.md, "one wave per SIMD").  hipcc would not keep a 512-register stream in the written order (round 5's kernel: all of a stage's
v_exp_f32 in front of its third MFMA, packed fp32 row sums, 5.9 issued instructions per MFMA gap -- profiles/r06_attn_pmc_arms_q64.log), so the whole
key loop is ONE asm statement whose private registers are named literally (and clobbered) and whose `s_waitcnt lgkmcnt(n)` are COUNTED here by
simulating the in-order LGKM queue.  The C++ around it (attention_q64.hip) keeps index arithmetic, the epilogue and the launch plan.

Pipeline (b = 32-key block, two per 64-key tile; h = 32-query half of the wave's 64 queries):
    stage b :  QK^T(b+1)  ||  softmax(b)  ||  PV(b-1)          32 MFMAs in eight groups  [QK h0, QK h1, PV h0, PV h1]
    fillers per group:  gap0  exp exp (+ LDS-DMA piece / scalar bookkeeping)     gap1  exp exp, K fragment read (AHEAD groups ahead)
                        gap2  add add cvt_pk, V^T fragment read                   gap3  add add cvt_pk, the next group's counted lgkmcnt
One loop trip = one tile = stage 2t+1, the tile's barrier (+ the counted vmcnt that retires the DMA batch due now), stage 2t+2 with the DMA of
K(t+N) / V(t+N-1) into the N-slot LDS rings (K and V^T tiles of 16 KB).  Fragment addresses live in registers and are stepped once per tile right
behind their last use.  The softmax is the sum-checked one of attention_glds.hip: scores leave the MFMA as s - m (the C operand is the -m block), m is the
exact maximum of the first block, a row sum beyond 2^40 marks the 64-query group for the repair pass.  Arithmetic order per element = the 8 x 32 kernel's.

Variants (Config): the product stream is `default`; `python tools/gen_attn_q64.py --variants` also writes attention_q64_asm_var.inc with the A/B arms that
attention_q64.hip compiles in the UTX_ABLATION build only (tools/attn_q64_check.py times them; arms named abl_* give WRONG results by design).
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
S_TK, S_KD, S_VD, S_KSTEP, S_VSTEP, S_TMP, S_CNT = 44, 45, 46, 47, 48, 49, 50
S_POS, S_NEG, S_KWK, S_KWV, S_KDEND, S_VDEND = 51, 52, 53, 54, 55, 56
S_ONES = S_KSTEP                                               # dot2 arm: the bf16 pair (1.0, 1.0); three tiles per trip step no address, so the register is free
S_LO, S_HI = 40, 56
TILE = 16384


class Config:
    def __init__(self, name, nslot=3, ahead=2, novm=False, nobar=False, noexp=False, nodma=False, unroll3=False, wait2=False, pkadd=False, m0once=False, dot2=False, nosum=False, dot2c=False, mfma16qk=False):
        self.name, self.nslot, self.ahead = name, nslot, ahead
        # unroll3: three tiles per loop trip -- ring slots are literals, every LDS address is a loop-invariant register + an immediate, no address steps, no slot bookkeeping;
        # nt % 3 tiles run through a second copy of the first two tile bodies behind the loop.  wait2: one counted lgkmcnt per TWO groups (needs read-ahead 3).
        self.unroll3, self.wait2 = unroll3, wait2
        # m0once: ONE M0 write per four pieces -- an LDS-DMA's instruction offset moves its LDS address AND its global address (tools/glds_offset_probe.hip on the
        # hardware), so piece j carries offset:1024 j and the kernel hands over per-lane source offsets lowered by 1024 j (attention_q64.hip, AQ2_M0_ONCE)
        self.m0once = m0once
        # dot2: the row sum of a lane's two keys of a group as ONE v_dot2_f32_bf16 over the packed pair PV consumes (P . (1, 1) + sum): one instruction instead of two adds per
        # gap, one sum register per half, and l sums exactly the bf16 probabilities the PV MFMAs multiply.  NOT the 8 x 32 kernel's rounding (that one sums the fp32 P)
        self.dot2 = dot2
        self.dot2c = dot2c      # the dot2 arm with the 4-byte VOP2 form v_dot2c_f32_bf16 (sum zeroed by a v_mov at the head of the stage): encoding or execution unit?
        # timing ablation (wrong results): every QK^T MFMA of the loop as TWO v_mfma_f32_16x16x32_bf16 on the same operand registers (same FLOPs, fragment reads, DMA and VALU work): the
        # bound of what the vendor GEMM's MFMA shape could buy this kernel, for half of its MFMAs
        self.mfma16qk = mfma16qk
        self.nosum = nosum      # timing ablation: no row-sum instruction at all (wrong results): the bound of anything done about the row sums
        self.pkadd = pkadd      # the two row-sum adds of a gap as ONE v_pk_add_f32 (same two IEEE additions, one instruction to fetch and issue)
        assert not unroll3 or nslot == 3
        assert not wait2 or ahead == 3
        assert not dot2 or unroll3
        self.novm, self.nobar, self.noexp, self.nodma = novm, nobar, noexp, nodma      # timing ablations (wrong results)
        assert nslot in (3, 4) and ahead in (2, 3)
        self.dk, self.dv = nslot, nslot - 1          # tile look-ahead of the DMA: K(t + dk), V(t + dv) are requested in trip t
        # DMA batches that may still be in flight at the tile's barrier: the 3-slot ring needs the batch requested one stage ago NOW (vmcnt(0)); with 4 slots the batch
        # due now was requested a whole tile earlier and the latest one (8 pieces per wave) stays in flight
        self.vm_at_barrier = 0 if nslot == 3 else 8


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
        self.gap = None         # instructions issued since the last MFMA (statistics)
        self.gaps = []

    def raw(self, text):
        self.lines.append(text)

    def ins(self, text):
        self.lines.append(text)
        if self.gap is not None:
            self.gap += 1

    def comment(self, text):
        assert "%" not in text      # a percent sign is an operand escape in an asm string, also inside a comment
        self.lines.append("; " + text)

    def mfma(self, dst, a_, b_, c_):
        if self.gap is not None:
            self.gaps.append(self.gap)
        self.gap = 0
        self.lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, a_, b_, c_))

    def mfma16x2(self, dst0, a_, b_, c0):
        """two 16x16x32 MFMAs in place of one 32x32x16: 4-register accumulators dst0 .. dst0+3 and dst0+4 .. dst0+7 (VGPR numbers), the same A / B registers"""
        if self.gap is not None:
            self.gaps.append(self.gap)
        self.gap = 0
        for k in (0, 4):
            self.lines.append("v_mfma_f32_16x16x32_bf16 v[%d:%d], %s, %s, v[%d:%d]" % (dst0 + k, dst0 + k + 3, a_, b_, c0 + k, c0 + k + 3))

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


def dma(st, voff_op, sbase, m0_base, m0_imm, cfg=None, j=0):
    """one 1 KB piece: M0 = LDS destination of the wave-instruction, then global_load_lds (saddr + per-lane 32-bit offset)"""
    if cfg is not None and cfg.m0once:
        if j == 0:
            st.ins("s_add_u32 m0, %s, %d" % (m0_base, m0_imm))
            st.ins("s_nop 0")
        st.ins("global_load_lds_dwordx4 %s, %s%s" % (voff_op, s(sbase, 2), (" offset:%d" % (1024 * j)) if j else ""))
        return
    st.ins("s_add_u32 m0, %s, %d" % (m0_base, m0_imm))
    st.ins("s_nop 0")
    st.ins("global_load_lds_dwordx4 %s, %s" % (voff_op, s(sbase, 2)))


def advance_ptr(st, ptr, step, cond_lhs, cond_rhs):
    """ptr += (cond_lhs < cond_rhs) ? step : 0   (unsigned; the last tile is requested again past the end: branch-free stages, no garbage in LDS)"""
    st.ins("s_cmp_lt_u32 %s, %s" % (cond_lhs, cond_rhs))
    st.ins("s_cselect_b32 %s, %s, 0" % (s(S_TMP), step))
    st.ins("s_add_u32 %s, %s, %s" % (s(ptr), s(ptr), s(S_TMP)))
    st.ins("s_addc_u32 %s, %s, 0" % (s(ptr + 1), s(ptr + 1)))


def k_read(st, cfg, tag, buf, odd_target, gt, jt=0):
    """K fragment gt of the block the TARGET stage's QK^T multiplies: odd stage 2t+1 -> (tile t+1, block 0); even stage 2t+2 -> (tile t+1, block 1), after which the
    address steps to tile t+2.  unroll3: t = 3 trip + jt, the slot of tile t+1 is the literal (jt + 1) % 3 and nothing steps"""
    if cfg.unroll3:
        st.ds_read(tag, v(KF[buf], 4), "%%[kx%d]" % gt, ((jt + 1) % 3) * TILE + (0 if odd_target else 8192))
    elif odd_target:
        st.ds_read(tag, v(KF[buf], 4), "%%[kx%d]" % gt, 0)
    else:
        st.ds_read(tag, v(KF[buf], 4), "%%[kx%d]" % gt, 8192)
        st.ins("v_add_u32 %%[kx%d], %s, %%[kx%d]" % (gt, s(S_KSTEP), gt))


def v_read(st, cfg, tag, buf, odd_target, gt, jt=0):
    """V^T fragment (key slab, d-block gt & 3) of the block the TARGET stage's PV consumes: odd stage -> tile t slabs 0, 1; even stage -> tile t slabs 2, 3; an address steps
    to tile t+1 behind its fourth d-block (unroll3: the slot of tile t is the literal jt % 3)"""
    slab = (gt >> 2) + (0 if odd_target else 2)
    if cfg.unroll3:
        st.ds_read(tag, v(VF[buf], 4), "%%[vx%d]" % slab, (jt % 3) * TILE + 4096 * (gt & 3))
        return
    st.ds_read(tag, v(VF[buf], 4), "%%[vx%d]" % slab, 4096 * (gt & 3))
    if (gt & 3) == 3:
        st.ins("v_add_u32 %%[vx%d], %s, %%[vx%d]" % (slab, s(S_VSTEP), slab))


def gen_prologue(st, cfg):
    N = cfg.nslot
    st.comment("---- scalars")
    st.ins("s_mov_b32 %s, %%[kptr_lo]" % s(S_KPTR))
    st.ins("s_mov_b32 %s, %%[kptr_hi]" % s(S_KPTR + 1))
    st.ins("s_mov_b32 %s, %%[vptr_lo]" % s(S_VPTR))
    st.ins("s_mov_b32 %s, %%[vptr_hi]" % s(S_VPTR + 1))
    st.ins("s_mov_b32 %s, 0x%x" % (s(S_POS), TILE))
    st.ins("s_mov_b32 %s, 0x%x" % (s(S_NEG), (-(N - 1) * TILE) & 0xffffffff))
    # the K dest slot of trip t is t % N: the read address of K wraps in trip t + 1 iff t % N == (N - 3) % N, that of V^T iff t % N == (N - 2) % N
    st.ins("s_add_u32 %s, %%[ldsk], 0x%x" % (s(S_KWK), ((N - 3) % N) * TILE))
    st.ins("s_add_u32 %s, %%[ldsk], 0x%x" % (s(S_KWV), ((N - 2) % N) * TILE))
    st.ins("s_add_u32 %s, %%[ldsk], 0x%x" % (s(S_KDEND), N * TILE))
    st.ins("s_add_u32 %s, %%[ldsv], 0x%x" % (s(S_VDEND), N * TILE))
    st.ins("s_mov_b32 %s, %s" % (s(S_KSTEP), s(S_POS)))
    st.ins("s_mov_b32 %s, %s" % (s(S_VSTEP), s(S_POS)))
    if cfg.dot2:
        st.ins("s_mov_b32 %s, 0x3f803f80" % s(S_ONES))
    st.comment("---- Q fragments of both 32-query halves, straight into AGPRs (MFMA B operands only)")
    for h, (qp, base) in enumerate((("%[qp0]", Q0), ("%[qp1]", Q1))):
        for kk in range(8):
            st.ins("global_load_dwordx4 %s, %s, off offset:%d" % (a(base + 4 * kk, 4), qp, 32 * kk))
    st.comment("---- ring fill: K(0 .. %d), V(0 .. %d); tile indices clamp at nt - 1" % (cfg.dk - 1, cfg.dv - 1))
    for u in range(cfg.dk):
        if u > 0:
            st.ins("s_mov_b32 %s, %d" % (s(S_TK), u))
            advance_ptr(st, S_KPTR, "%[kstride]", s(S_TK), "%[nt]")
            if u < cfg.dv:
                advance_ptr(st, S_VPTR, "0x80", s(S_TK), "%[nt]")
        for j in range(4):
            dma(st, "%%[ko%d]" % j, S_KPTR, "%[ldsk]", u * TILE + (0 if cfg.m0once else 1024 * j), cfg, j)
        if u < cfg.dv:
            for j in range(4):
                dma(st, "%%[vo%d]" % j, S_VPTR, "%[ldsv]", u * TILE + (0 if cfg.m0once else 1024 * j), cfg, j)
    st.comment("state of 'trip -1': K dest slot N-1, V dest slot N-2, K pointer at tile min(N-1, nt-1), V pointer at tile min(N-2, nt-1), TK = N - 1")
    st.ins("s_add_u32 %s, %%[ldsk], 0x%x" % (s(S_KD), (N - 1) * TILE))
    st.ins("s_add_u32 %s, %%[ldsv], 0x%x" % (s(S_VD), (N - 2) * TILE))
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
        if not cfg.unroll3:
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
        if not cfg.dot2:
            st.ins("v_mov_b32 %s, %s" % (v(ev), v(sa0)))
            st.ins("v_mov_b32 %s, %s" % (v(od), v(sa0 + 1)))
            for r in range(2, 16, 2):
                st.ins("v_add_f32 %s, %s, %s" % (v(ev), v(ev), v(sa0 + r)))
                st.ins("v_add_f32 %s, %s, %s" % (v(od), v(od), v(sa0 + r + 1)))
        for i in range(8):
            st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(0, h)] + i), v(sa0 + 2 * i), v(sa0 + 2 * i + 1)))
        if cfg.dot2:
            for i in range(8):
                st.ins("v_dot2_f32_bf16 %s, %s, %s, %s" % (v(ev), v(PB[(0, h)] + i), s(S_ONES), "0" if i == 0 else v(ev)))
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
    st.comment("---- fragments of the first AHEAD groups of stage 1: K(tile 1, block 0), V(tile 0, slab 0)")
    for G in range(cfg.ahead):
        k_read(st, cfg, ("K", G), G % 4, True, G, 0)
        v_read(st, cfg, ("V", G), G % 4, True, G, 0)
    st.ins("s_mov_b32 %s, %s" % (s(S_CNT), "%[trips]" if cfg.unroll3 else "%[nt]"))
    st.ins("s_nop 3")


def gen_stage(st, cfg, G0, odd, j=0):
    """one stage of the loop; G0 = running index of its first group (read tags); odd: stage 2t+1 (softmax of a tile's block 1), else stage 2t+2"""
    p = 1 if odd else 0            # parity of the block whose softmax runs
    q = 1 - p                      # parity of the block QK^T writes / PV reads
    sums_p, sums_q = SUM[p], SUM[q]
    A = cfg.ahead
    EXP = "v_mov_b32" if cfg.noexp else "v_exp_f32"
    for g in range(8):
        G = G0 + g
        if not cfg.wait2:
            st.wait_lgkm([("K", G), ("V", G)])
        elif g % 2 == 0:
            st.wait_lgkm([("K", G), ("V", G), ("K", G + 1), ("V", G + 1)])
        # ---- QK^T, half 0
        if cfg.mfma16qk:
            st.mfma16x2(SA[(q, 0)], v(KF[G % 4], 4), a(Q0 + 4 * g, 4), NEGM[0] if g == 0 else SA[(q, 0)])
        else:
            st.mfma(v(SA[(q, 0)], 16), v(KF[G % 4], 4), a(Q0 + 4 * g, 4), v(NEGM[0], 16) if g == 0 else v(SA[(q, 0)], 16))
        if odd:      # scalar bookkeeping of the trip, two or three per gap (state of trip t from that of trip t - 1)
            if g == 0 and cfg.unroll3:
                pass
            elif g == 1 and cfg.unroll3:
                pass
            elif g == 0:
                st.ins("s_add_u32 %s, %s, 0x%x" % (s(S_KD), s(S_KD), TILE))
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KDEND)))
                st.ins("s_cselect_b32 %s, %%[ldsk], %s" % (s(S_KD), s(S_KD)))
            elif g == 1:
                st.ins("s_add_u32 %s, %s, 0x%x" % (s(S_VD), s(S_VD), TILE))
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_VD), s(S_VDEND)))
                st.ins("s_cselect_b32 %s, %%[ldsv], %s" % (s(S_VD), s(S_VD)))
            elif g == 2:
                st.ins("s_add_u32 %s, %s, 1" % (s(S_TK), s(S_TK)))
                st.ins("s_cmp_lt_u32 %s, %%[nt]" % s(S_TK))
                st.ins("s_cselect_b32 %s, %%[kstride], 0" % s(S_TMP))
            elif g == 3:
                st.ins("s_add_u32 %s, %s, %s" % (s(S_KPTR), s(S_KPTR), s(S_TMP)))
                st.ins("s_addc_u32 %s, %s, 0" % (s(S_KPTR + 1), s(S_KPTR + 1)))
            elif g == 4:
                st.ins("s_sub_u32 %s, %s, 1" % (s(S_TMP), s(S_TK)))
                st.ins("s_cmp_lt_u32 %s, %%[nt]" % s(S_TMP))
                st.ins("s_cselect_b32 %s, 0x80, 0" % s(S_TMP))
            elif g == 5:
                st.ins("s_add_u32 %s, %s, %s" % (s(S_VPTR), s(S_VPTR), s(S_TMP)))
                st.ins("s_addc_u32 %s, %s, 0" % (s(S_VPTR + 1), s(S_VPTR + 1)))
        elif not cfg.nodma:        # LDS-DMA piece g of the tile boundary: K(t+N) pieces 0-3, V(t+N-1) pieces 0-3; M0 first, the load behind the exponentials
            if cfg.unroll3 and cfg.m0once:
                if g == 0:
                    st.ins("s_add_u32 m0, %%[ldsk], %d" % ((j % 3) * TILE))
                elif g == 4:
                    st.ins("s_add_u32 m0, %%[ldsv], %d" % (((j + 2) % 3) * TILE))
            elif cfg.unroll3:      # K(t+3) -> slot t % 3 = j, V(t+2) -> slot (j + 2) % 3
                st.ins("s_add_u32 m0, %s, %d" % (("%[ldsk]", (j % 3) * TILE + 1024 * g) if g < 4 else ("%[ldsv]", ((j + 2) % 3) * TILE + 1024 * (g - 4))))
            elif g < 4:
                st.ins("s_add_u32 m0, %s, %d" % (s(S_KD), 1024 * g))
            else:
                st.ins("s_add_u32 m0, %s, %d" % (s(S_VD), 1024 * (g - 4)))
        if cfg.dot2c and g == 0:
            st.ins("v_mov_b32 %s, 0" % v(sums_p[0]))
            st.ins("v_mov_b32 %s, 0" % v(sums_p[2]))
        ev0 = sums_p[0] if (g == 0 and not cfg.dot2) else E[0]
        od0 = sums_p[1] if (g == 0 and not cfg.dot2) else E[1]
        st.ins("%s %s, %s" % (EXP, v(ev0), v(SA[(p, 0)] + 2 * g)))
        st.ins("%s %s, %s" % (EXP, v(od0), v(SA[(p, 0)] + 2 * g + 1)))
        if not odd and not cfg.nodma:
            off = (" offset:%d" % (1024 * (g & 3))) if (cfg.m0once and (g & 3)) else ""
            if g < 4:
                st.ins("global_load_lds_dwordx4 %%[ko%d], %s%s" % (g, s(S_KPTR, 2), off))
            else:
                st.ins("global_load_lds_dwordx4 %%[vo%d], %s%s" % (g - 4, s(S_VPTR, 2), off))
        # ---- QK^T, half 1
        if cfg.mfma16qk:
            st.mfma16x2(SA[(q, 1)], v(KF[G % 4], 4), a(Q1 + 4 * g, 4), NEGM[1] if g == 0 else SA[(q, 1)])
        else:
            st.mfma(v(SA[(q, 1)], 16), v(KF[G % 4], 4), a(Q1 + 4 * g, 4), v(NEGM[1], 16) if g == 0 else v(SA[(q, 1)], 16))
        ev1 = sums_p[2] if (g == 0 and not cfg.dot2) else E[2]
        od1 = sums_p[3] if (g == 0 and not cfg.dot2) else E[3]
        st.ins("%s %s, %s" % (EXP, v(ev1), v(SA[(p, 1)] + 2 * g)))
        st.ins("%s %s, %s" % (EXP, v(od1), v(SA[(p, 1)] + 2 * g + 1)))
        gt = g + A                 # the group AHEAD groups on, in this stage or the next
        tgt_odd = odd if gt < 8 else (not odd)
        jt = j + 1 if (gt >= 8 and not odd) else j      # the trip-local tile index of the TARGET stage
        k_read(st, cfg, ("K", G + A), (G + A) % 4, tgt_odd, gt % 8, jt)
        if not odd and g >= 6 and not cfg.unroll3:     # the address steps of the NEXT trip, behind the last step of this one (even-stage targets are read through group 7 - AHEAD)
            if g == 6:             # V^T read address wraps in trip t + 1 iff t % N == (N - 2) % N
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KWV)))
                st.ins("s_cselect_b32 %s, %s, %s" % (s(S_VSTEP), s(S_NEG), s(S_POS)))
            else:
                st.ins("s_cmp_eq_u32 %s, %s" % (s(S_KD), s(S_KWK)))
                st.ins("s_cselect_b32 %s, %s, %s" % (s(S_KSTEP), s(S_NEG), s(S_POS)))
        # ---- PV, half 0
        st.mfma("%%[o0%d]" % (g & 3), v(VF[G % 4], 4), v(PB[(q, 0)] + 4 * (g >> 2), 4), "%%[o0%d]" % (g & 3))
        if cfg.dot2:
            if g == 0:
                st.ins("v_add_f32 %[l0], %[l0], " + v(sums_q[0]))
        elif g == 0:   # the row sums of the previous stage's block are final: l += ps (block order), headroom record
            st.ins("v_add_f32 %s, %s, %s" % (v(PS0), v(sums_q[0]), v(sums_q[1])))
            st.ins("v_add_f32 %[l0], %[l0], " + v(PS0))
        elif cfg.nosum:
            pass
        elif cfg.pkadd:
            st.ins("v_pk_add_f32 %s, %s, %s" % (v(sums_p[0], 2), v(sums_p[0], 2), v(E[0], 2)))
        else:
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[0]), v(sums_p[0]), v(E[0])))
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[1]), v(sums_p[1]), v(E[1])))
        st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(p, 0)] + g), v(ev0), v(od0)))
        v_read(st, cfg, ("V", G + A), (G + A) % 4, tgt_odd, gt % 8, jt)
        if cfg.dot2:
            if cfg.dot2c:
                st.ins("v_dot2c_f32_bf16 %s, %s, %s" % (v(sums_p[0]), s(S_ONES), v(PB[(p, 0)] + g)))
            else:
                st.ins("v_dot2_f32_bf16 %s, %s, %s, %s" % (v(sums_p[0]), v(PB[(p, 0)] + g), s(S_ONES), "0" if g == 0 else v(sums_p[0])))
        # ---- PV, half 1
        st.mfma("%%[o1%d]" % (g & 3), v(VF[G % 4], 4), v(PB[(q, 1)] + 4 * (g >> 2), 4), "%%[o1%d]" % (g & 3))
        if cfg.dot2:
            if g == 0:
                st.ins("v_add_f32 %[l1], %[l1], " + v(sums_q[2]))
                st.ins("v_max3_f32 %%[psmax], %%[psmax], %s, %s" % (v(sums_q[0]), v(sums_q[2])))
        elif g == 0:
            st.ins("v_add_f32 %s, %s, %s" % (v(PS1), v(sums_q[2]), v(sums_q[3])))
            st.ins("v_add_f32 %[l1], %[l1], " + v(PS1))
            st.ins("v_max3_f32 %%[psmax], %%[psmax], %s, %s" % (v(PS0), v(PS1)))
        elif cfg.nosum:
            pass
        elif cfg.pkadd:
            st.ins("v_pk_add_f32 %s, %s, %s" % (v(sums_p[2], 2), v(sums_p[2], 2), v(E[2], 2)))
        else:
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[2]), v(sums_p[2]), v(E[2])))
            st.ins("v_add_f32 %s, %s, %s" % (v(sums_p[3]), v(sums_p[3]), v(E[3])))
        st.ins("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(PB[(p, 1)] + g), v(ev1), v(od1)))
        if cfg.dot2:
            if cfg.dot2c:
                st.ins("v_dot2c_f32_bf16 %s, %s, %s" % (v(sums_p[2]), s(S_ONES), v(PB[(p, 1)] + g)))
            else:
                st.ins("v_dot2_f32_bf16 %s, %s, %s, %s" % (v(sums_p[2]), v(PB[(p, 1)] + g), s(S_ONES), "0" if g == 0 else v(sums_p[2])))


def gen_tile(st, cfg, G0, j):
    """one tile: stage 2t+1, the boundary, stage 2t+2 (t = 3 trip + j under unroll3)"""
    gen_stage(st, cfg, G0, True, j)
    st.comment("---- tile boundary: the DMA batch due now has landed (this wave's pieces: the counted vmcnt); behind the barrier it is visible and the slots of K(t) / V(t-1) are free")
    if not cfg.novm:
        st.ins("s_waitcnt vmcnt(%d)" % cfg.vm_at_barrier)
    if not cfg.nobar:
        st.ins("s_barrier")
    gen_stage(st, cfg, G0 + 8, False, j)


def gen(cfg):
    st = Stream()
    gen_prologue(st, cfg)
    entry = list(st.lgkm)
    assert entry == [(k, G) for G in range(cfg.ahead) for k in ("K", "V")], entry
    ntile = 3 if cfg.unroll3 else 1
    if cfg.unroll3:
        st.ins("s_cmp_eq_u32 %[trips], 0")
        st.ins("s_cbranch_scc1 AQ2_TAIL_%=")
    st.raw("AQ2_LOOP_%=:")
    n0 = len(st.lines)
    st.gap = None
    for j in range(ntile):
        gen_tile(st, cfg, 16 * j, j)
    st.ins("s_sub_u32 %s, %s, 1" % (s(S_CNT), s(S_CNT)))
    st.ins("s_cmp_lg_u32 %s, 0" % s(S_CNT))
    st.ins("s_cbranch_scc1 AQ2_LOOP_%=")
    back = [(k, G - 16 * ntile) for k, G in st.lgkm]
    assert back == entry, (back, entry)       # the LGKM queue at the back edge is the queue at the loop's entry: the counted waits hold on every trip
    loop_lines = [l for l in st.lines[n0:] if not l.startswith(";")]
    gaps = list(st.gaps)
    if cfg.unroll3:
        st.comment("---- the nt mod 3 tiles behind the loop: the first two tile bodies once more (the loop always leaves the ring at slot phase 0)")
        st.raw("AQ2_TAIL_%=:")
        st.lgkm = list(entry)
        st.ins("s_cmp_eq_u32 %[rem], 0")
        st.ins("s_cbranch_scc1 AQ2_END_%=")
        gen_tile(st, cfg, 0, 0)
        assert [(k, G - 16) for k, G in st.lgkm] == entry
        st.ins("s_cmp_eq_u32 %[rem], 1")
        st.ins("s_cbranch_scc1 AQ2_END_%=")
        gen_tile(st, cfg, 16, 1)
        assert [(k, G - 32) for k, G in st.lgkm] == entry
        st.raw("AQ2_END_%=:")
    st.comment("---- drain: the last PV MFMAs, the prefetched fragments and the clamped re-requests")
    st.ins("s_nop 15")
    st.ins("s_nop 15")
    st.ins("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return st, loop_lines, gaps


OPERANDS_OUT = [("o%d%d" % (h, d), "+a", "oacc%d[%d]" % (h, d)) for h in range(2) for d in range(4)] + \
    [("l0", "=&v", "l_run0"), ("l1", "=&v", "l_run1"), ("m0r", "=&v", "m_run0"), ("m1r", "=&v", "m_run1"), ("psmax", "=&v", "ps_max")] + \
    [("kx%d" % i, "+v", "kx[%d]" % i) for i in range(8)] + [("vx%d" % i, "+v", "vx[%d]" % i) for i in range(4)]
OPERANDS_IN = [("ko%d" % i, "v", "ko[%d]" % i) for i in range(4)] + [("vo%d" % i, "v", "vo[%d]" % i) for i in range(4)] + \
    [("qp0", "v", "qp0"), ("qp1", "v", "qp1"), ("bperm", "v", "bperm"),
     ("kptr_lo", "s", "kptr_lo"), ("kptr_hi", "s", "kptr_hi"), ("vptr_lo", "s", "vptr_lo"), ("vptr_hi", "s", "vptr_hi"),
     ("kstride", "s", "kstride"), ("nt", "s", "nt"), ("ldsk", "s", "ldsk"), ("ldsv", "s", "ldsv"), ("kbv", "s", "kbv"), ("trips", "s", "trips"), ("rem", "s", "rem")]

DEFAULT = Config("default", unroll3=True, ahead=3, wait2=True)
# A/B arms (UTX_ABLATION build; UTX_ATTN_VAR = index + 1 selects one).  Correct results unless named abl_*.  Measured, profiles/r06_attn_q64_arms_v*.log (TF/s at S = 13 376 / 50 240,
# same process, interleaved; every correct arm bit-identical to the 8 x 32 kernel on the sweep):
#   tile1 (the first stream of round 6: one tile per trip, addresses stepped in registers, read-ahead 2)   1380 / 1413      default (three tiles per trip, literal slots)  1417 / 1443
#   ring4 / read-ahead 3 alone / no vmcnt / no barrier: equal to tile1 within 0.4 %  -- the loop is not waiting for anything; it is POWER-bound (time = energy / cap)
#   m0 (one M0 write per four pieces: -6 SALU per tile) = default      pk (row sums as v_pk_add_f32: -32 instructions per tile) 1294 / 1321: packed fp32 beside MFMAs LOSES 10 %
#   abl_noexp +5.7 %, abl_nodma +4.6 % over tile1: what the exponentials and the L2 -> LDS traffic cost
#   dot2 / dot2c (row sums as ONE v_dot2_f32_bf16 / v_dot2c_f32_bf16 over the packed pair PV consumes: -32 instructions per tile, accuracy against fp64 equal or better)  1297 / 1332 and 1292 / 1320
#   against 1391 / 1424 for the default in the same process: -7 %, and in CYCLES (3.25 -> 4.06e7 per CU, matrix pipe 89 -> 71 % busy at a clock that rises 1.51 -> 1.74 GHz): a VOP3P / dot
#   instruction does not issue in an MFMA's shadow on gfx950 (~17 cycles each), like v_pk_add_f32.  abl_nosum (no row-sum instruction at all) +2.2 %: the bound of anything done about the sums
#   (profiles/r06_attn_q64_arms_v5.log, r06_attn_pmc_arms_rowsum.log, r06_attn_q64_dot2_accuracy.log)
#   abl_mfma16qk (every QK^T MFMA as two 16x16x32 on the same operand registers: the vendor GEMM's shape, the one the bare-MFMA probe sustains 10 % more of) 1398 / 1454 against 1406 / 1453:
#   EQUAL in time -- 19 % more cycles (3.33 -> 3.97e7 per CU, matrix pipe 87 -> 73 % busy) at a clock 18 % higher (1.55 -> 1.83 GHz): the same energy for the same work, so the shape buys
#   this kernel nothing (profiles/r06_attn_q64_arms_v6.log); a correct 16x16x32 stream was therefore not written
VARIANTS = [Config("tile1"), Config("ring4", nslot=4), Config("m0", unroll3=True, ahead=3, wait2=True, m0once=True), Config("pk", unroll3=True, ahead=3, wait2=True, pkadd=True),
            Config("abl_novm", novm=True), Config("abl_nobar", novm=True, nobar=True), Config("abl_noexp", noexp=True), Config("abl_nodma", nodma=True, novm=True),
            Config("dot2", unroll3=True, ahead=3, wait2=True, dot2=True), Config("abl_nosum", unroll3=True, ahead=3, wait2=True, nosum=True),
            Config("dot2c", unroll3=True, ahead=3, wait2=True, dot2=True, dot2c=True), Config("abl_mfma16qk", unroll3=True, ahead=3, wait2=True, mfma16qk=True)]


def write_text(f, macro, st):
    f.write("#define %s \\\n" % macro)
    for l in st.lines:
        f.write('    "%s\\n\\t" \\\n' % l.replace("\\", "\\\\").replace('"', '\\"'))
    f.write('    ""\n')


def census(cfg, loop_lines, gaps):
    n_mfma = sum(1 for l in loop_lines if l.startswith("v_mfma"))
    n_other = len(loop_lines) - n_mfma
    loop_gaps = gaps[-n_mfma:]
    if cfg.unroll3:      # per tile
        n_mfma //= 3; n_other = round(n_other / 3.0, 1)
    return n_mfma, n_other, {k: loop_gaps.count(k) for k in sorted(set(loop_gaps))}


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "unitex_amd", "csrc", "attention_q64_asm.inc")
    st, loop_lines, gaps = gen(DEFAULT)
    n_mfma, n_other, hist = census(DEFAULT, loop_lines, gaps)
    with open(os.devnull if "--keep-default" in sys.argv else out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_q64.py -- do not edit; the generator's header explains the stream.\n")
        f.write("// loop trip (one 64-key tile, 64 queries per wave): %d MFMAs, %d other instructions (%.2f per MFMA gap; histogram of gap sizes %s)\n" % (n_mfma, n_other, n_other / n_mfma, hist))
        f.write("#define AQ2_NSLOT %d\n#define AQ2_M0_ONCE %d\n" % (DEFAULT.nslot, int(DEFAULT.m0once)))
        write_text(f, "AQ2_ASM_TEXT", st)
        f.write("#define AQ2_ASM_OUTPUTS " + ", ".join('[%s] "%s"(%s)' % o for o in OPERANDS_OUT) + "\n")
        f.write("#define AQ2_ASM_INPUTS " + ", ".join('[%s] "%s"(%s)' % o for o in OPERANDS_IN) + "\n")
        clob = ['"memory"', '"vcc"', '"scc"'] + ['"v%d"' % i for i in range(V_LO, V_HI + 1)] + ['"a%d"' % i for i in range(A_LO, A_HI + 1)] + ['"s%d"' % i for i in range(S_LO, S_HI + 1)]
        f.write("#define AQ2_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    print("wrote %s: loop %d MFMAs + %d others = %.2f per gap; gap histogram %s" % (out, n_mfma, n_other, n_other / n_mfma, hist))
    if "--variants" in sys.argv:
        outv = os.path.join(root, "unitex_amd", "csrc", "attention_q64_asm_var.inc")
        with open(outv, "w") as f:
            f.write("// GENERATED by tools/gen_attn_q64.py --variants -- A/B arms of the 4 x 64 stream for the UTX_ABLATION build (git-ignored; never part of the product library).\n")
            f.write("#define AQ2_NVAR %d\n" % len(VARIANTS))
            for i, cfg in enumerate(VARIANTS):
                stv, ll, gg = gen(cfg)
                n_mfma, n_other, hist = census(cfg, ll, gg)
                f.write("// arm %d = %s: %.2f per gap %s\n#define AQ2_NSLOT_V%d %d\n#define AQ2_M0_ONCE_V%d %d\n" % (i + 1, cfg.name, n_other / n_mfma, hist, i + 1, cfg.nslot, i + 1, int(cfg.m0once)))
                write_text(f, "AQ2_ASM_TEXT_V%d" % (i + 1), stv)
                print("  arm %d = %-14s %.2f per gap %s" % (i + 1, cfg.name, n_other / n_mfma, hist))


if __name__ == "__main__":
    main()
