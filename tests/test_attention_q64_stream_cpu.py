"""The generated instruction stream of the 4 x 64 attention kernel (unitex_amd/csrc/attention_q64_asm.inc, tools/gen_attn_q64.py), checked on the CPU by an
INDEPENDENT interpreter of the emitted text -- not by the generator's own bookkeeping:

  * the committed .inc is what the generator writes (nobody edited one without the other);
  * every MFMA operand that comes from LDS was filled by a ds_read that the stream has WAITED for (in-order lgkmcnt model), and no ds_read lands in a
    fragment buffer between its fill and its last consumer;
  * ring discipline over whole launches of 1 .. 11 tiles: every K / V^T fragment read hits the slot that holds the tile its consumer needs, that tile was
    requested, retired (counted vmcnt) and published (barrier) before the read, and no LDS-DMA overwrites a slot that any wave may still read (a request is
    legal only behind the barrier that follows the slot's last read);
  * register choreography: the n-th QK^T chain multiplies K block n with the right Q fragments into the score block the (n+1)-th softmax reads, the
    exponentials read scores that were complete at least two MFMAs earlier (the distance the MFMA -> VALU hazard needs; nothing pads inside asm), the
    probability words the PV MFMAs take are the ones the previous stage packed, in key-slab order;
  * the pointers of the clamped re-requests never leave the sequence, M0 is written before every LDS-DMA and not in the instruction directly in front of it.
The same checks run on the A/B arms of the generator (ring of 4, read-ahead 3), so an arm that is promoted to the product has already passed them."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_attn_q64 as G  # noqa: E402

TILE = 16384
KSTRIDE = 1 << 20          # bytes between K tiles in the model's address space (anything > 0 and distinct from V's 128)
KBASE, VBASE = 0x10000000, 0x70000000


def _stream_lines(cfg):
    st, _loop, _gaps = G.gen(cfg)
    return [l for l in st.lines if not l.startswith(";")]


def test_committed_stream_is_the_generators_output():
    st, _, _ = G.gen(G.DEFAULT)
    text = open(os.path.join(ROOT, "unitex_amd", "csrc", "attention_q64_asm.inc")).read()
    emitted = [m.group(1) for m in re.finditer(r'^    "(.*)\\n\\t" \\$', text, re.M)]
    assert emitted == [l for l in st.lines], "attention_q64_asm.inc is stale: run python tools/gen_attn_q64.py"
    assert "#define AQ2_NSLOT %d" % G.DEFAULT.nslot in text


class Machine:
    """one wave of the stream: scalar / address arithmetic interpreted exactly, VGPR contents carried as symbolic tags"""

    def __init__(self, lines, cfg, nt, kbv=0):
        self.lines, self.cfg, self.nt = lines, cfg, nt
        self.N = cfg.nslot
        self.vring = self.N * TILE
        self.reg = {}                     # scalar registers and named operands -> int
        self.tag = {}                     # VGPR / AGPR number -> symbolic content
        self.ready_at = {}                # VGPR number (score registers) -> index of the MFMA that completed it (count of MFMAs issued so far)
        self.labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        wave = 1
        self.dma_off = 4 * wave * 1024
        op = self.reg
        op["%[kptr_lo]"], op["%[kptr_hi]"] = KBASE & 0xffffffff, KBASE >> 32
        op["%[vptr_lo]"], op["%[vptr_hi]"] = VBASE & 0xffffffff, VBASE >> 32
        op["%[kstride]"], op["%[nt]"], op["%[kbv]"] = KSTRIDE, nt, kbv
        op["%[trips]"], op["%[rem]"] = nt // 3, nt % 3
        op["%[ldsk]"], op["%[ldsv]"] = self.dma_off, self.vring + self.dma_off
        for i in range(8):
            op["%%[kx%d]" % i] = 32 * i                      # distinct per fragment, < 8192
        for i in range(4):
            op["%%[vx%d]" % i] = self.vring + 64 * i         # distinct per key slab, < 4096
        for i in range(4):
            op["%%[ko%d]" % i], op["%%[vo%d]" % i] = 1000 + i, 2000 + i
        self.scc = 0
        self.m0 = None
        self.m0_age = 99
        self.vm = []                      # outstanding vector-memory operations (in order)
        self.lgkm = []                    # outstanding LDS operations (in order): ("frag", buffer, address) | ("bperm",)
        self.epoch = 0                    # barriers passed
        self.dmas = {}                    # LDS piece (byte address) -> list of dicts(tile, kind, issue_epoch, retired_epoch, pub_epoch)
        self.frag = {}                    # fragment buffer base register -> dict(addr, landed, used)
        self.n_mfma = 0
        self.qk_count = {0: 0, 1: 0}      # QK^T MFMAs seen per half
        self.pv_count = {0: 0, 1: 0}
        self.exp_seen = {}
        self.errors = []
        self.reads = []                   # (kind, address, epoch) of every consumed fragment, for the overwrite rule

    # ---------------------------------------------------------------- helpers
    def val(self, tok):
        tok = tok.strip()
        if tok in self.reg:
            return self.reg[tok]
        if tok == "m0":
            return self.m0
        if tok.startswith("0x"):
            return int(tok, 16)
        if re.fullmatch(r"-?\d+", tok):
            return int(tok) & 0xffffffff
        raise KeyError(tok)

    def err(self, i, msg):
        self.errors.append("line %d `%s`: %s" % (i, self.lines[i], msg))

    @staticmethod
    def regs(tok):
        """'v[112:115]' / 'v80' / 'a[128:131]' -> (file, first, count)"""
        m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok.strip())
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.fullmatch(r"([va])(\d+)", tok.strip())
        if m:
            return m.group(1), int(m.group(2)), 1
        return None

    def retire_lgkm(self, n):
        while len(self.lgkm) > n:
            e = self.lgkm.pop(0)
            if e[0] == "frag":
                self.frag[e[1]]["landed"] = True

    def retire_vm(self, n):
        while len(self.vm) > n:
            e = self.vm.pop(0)
            if e is not None:
                e["retired_epoch"] = self.epoch

    # ---------------------------------------------------------------- expectations (the pipeline's definition, independent of the generator's code)
    def expect_k(self, n):
        """the n-th QK^T MFMA of a half consumes K fragment kk = n % 8 of block n // 8"""
        blk, kk = n // 8, n % 8
        return blk // 2, (blk & 1) * 8192 + 32 * kk

    def expect_v(self, n):
        """the n-th PV MFMA of a half: block n // 8 (the loop's first PV is block 0), key slab (n % 8) >> 2 of the block, d-block n & 3"""
        blk, g = n // 8, n % 8
        slab = 2 * (blk & 1) + (g >> 2)
        return blk // 2, 64 * slab + 4096 * (g & 3), blk, g

    def check_fragment(self, i, kind, buf, tile, inner):
        f = self.frag.get(buf)
        if f is None:
            return self.err(i, "MFMA reads fragment buffer v%d that no ds_read filled" % buf)
        if not f["landed"]:
            self.err(i, "MFMA reads fragment buffer v%d before a wait covered its ds_read" % buf)
        f["used"] = True
        base = 0 if kind == "K" else self.vring
        off = f["addr"] - base
        slot, within = off // TILE, off % TILE
        if not (0 <= slot < self.N):
            return self.err(i, "%s fragment address 0x%x outside its ring" % (kind, f["addr"]))
        if tile > self.nt - 1:
            return                       # the pipeline's tail: scores of tiles that do not exist are computed and never used; any in-ring address will do
        if within != inner:
            self.err(i, "%s fragment of tile %d read at in-tile offset 0x%x, expected 0x%x" % (kind, tile, within, inner))
        # the slot must hold `tile`, published, and must not be re-requested before a barrier behind this read
        piece = base + slot * TILE + self.dma_off          # this wave's own first piece stands for the tile (all waves run the same stream)
        hist = self.dmas.get(piece, [])
        cur = [d for d in hist if d["issue_idx"] < f["issue_idx"]]
        if not cur or cur[-1]["tile"] != tile or cur[-1]["kind"] != kind:
            return self.err(i, "%s tile %d expected in slot %d, the slot's last request before the read was %s" % (kind, tile, slot, cur[-1] if cur else None))
        d = cur[-1]
        if d["pub_epoch"] is None or d["pub_epoch"] > f["epoch"]:
            self.err(i, "%s tile %d in slot %d read in barrier epoch %d but published in %s (retired %s)" % (kind, tile, slot, f["epoch"], d["pub_epoch"], d["retired_epoch"]))
        f["piece"], f["dma"] = piece, d

    # ---------------------------------------------------------------- execution
    def run(self, max_steps=2_000_000):
        pc, steps = 0, 0
        lines = self.lines
        while pc < len(lines):
            steps += 1
            assert steps < max_steps, "runaway loop"
            l = lines[pc]
            i = pc
            pc += 1
            if l.endswith(":"):
                continue
            op, _, rest = l.partition(" ")
            a = [x.strip() for x in rest.split(",")] if rest else []
            self.m0_age += 1
            if op == "s_mov_b32":
                if a[0] == "m0":
                    self.m0, self.m0_age = self.val(a[1]), 0
                else:
                    self.reg[a[0]] = self.val(a[1])
            elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
                x, y = self.val(a[1]), self.val(a[2])
                r = x + y + (self.scc if op == "s_addc_u32" else 0) if op != "s_sub_u32" else x - y
                self.scc = 1 if (r >> 32) != 0 or r < 0 else 0
                r &= 0xffffffff
                if a[0] == "m0":
                    self.m0, self.m0_age = r, 0
                else:
                    self.reg[a[0]] = r
            elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_lt_u32"):
                x, y = self.val(a[0]), self.val(a[1])
                self.scc = int({"s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_lt_u32": x < y}[op])
            elif op == "s_cselect_b32":
                self.reg[a[0]] = self.val(a[1]) if self.scc else self.val(a[2])
            elif op == "s_cbranch_scc1":
                if self.scc:
                    pc = self.labels[a[0]]
            elif op == "s_nop":
                pass
            elif op == "s_barrier":
                self.epoch += 1
                for hist in self.dmas.values():
                    for d in hist:
                        if d["pub_epoch"] is None and d["retired_epoch"] is not None:
                            d["pub_epoch"] = self.epoch
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    self.retire_vm(int(m.group(1)))
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m:
                    self.retire_lgkm(int(m.group(1)))
            elif op == "global_load_dwordx4":
                self.vm.append(None)
                f_, first, n = self.regs(a[0])
                for r in range(n):
                    self.tag[(f_, first + r)] = ("Q", 0 if a[1] == "%[qp0]" else 1, (int(re.search(r"offset:(\d+)", rest).group(1)) // 32), r)
            elif op == "global_load_lds_dwordx4":
                if self.m0 is None or self.m0_age < 2:
                    self.err(i, "LDS-DMA directly behind (or without) its M0 write")
                m = re.fullmatch(r"s\[(\d+):(\d+)\]", a[1].split()[0])
                ptr = self.reg["s" + m.group(1)] | (self.reg["s" + m.group(2)] << 32)
                kind = "K" if a[0].startswith("%[ko") else "V"
                j = int(a[0][4])
                if kind == "K":
                    tile, rem = divmod(ptr - KBASE, KSTRIDE)
                else:
                    tile, rem = divmod(ptr - VBASE, 128)
                if rem or not (0 <= tile < self.nt):
                    self.err(i, "%s request through pointer 0x%x: not a tile of the sequence (nt = %d)" % (kind, ptr, self.nt))
                base = 0 if kind == "K" else self.vring
                mo = re.search(r"offset:(\d+)", rest)
                dest = self.m0 + (int(mo.group(1)) if mo else 0)      # the instruction offset moves the LDS address too (tools/glds_offset_probe.hip, measured)
                off = dest - base - self.dma_off - 1024 * j
                if off % TILE or not (0 <= off // TILE < self.N):
                    self.err(i, "%s piece %d lands at LDS 0x%x: not piece %d of this wave in a ring slot" % (kind, j, dest, j))
                d = dict(tile=tile, kind=kind, issue_epoch=self.epoch, issue_idx=i + steps * 0, retired_epoch=None, pub_epoch=None, step=steps)
                d["issue_idx"] = steps
                self.vm.append(d)
                if j == 0:
                    piece = dest
                    # overwrite rule: every consumed read of this slot's previous content lies in an EARLIER barrier epoch (another wave may lag up to the last barrier)
                    for f in self.frag_history:
                        if f.get("piece") == piece and f["epoch"] >= self.epoch and f["dma"]["tile"] != tile:
                            self.err(i, "%s tile %d requested into a slot whose tile %d was read in the same barrier epoch %d" % (kind, tile, f["dma"]["tile"], self.epoch))
                    self.dmas.setdefault(piece, []).append(d)
            elif op == "ds_read_b128":
                f_, first, n = self.regs(a[0])
                m = re.search(r"offset:(\d+)", rest)
                addr_tok = a[1].split()[0]
                addr = self.val(addr_tok) + (int(m.group(1)) if m else 0)
                old = self.frag.get(first)
                if old is not None and not old["used"] and not old.get("prologue"):
                    self.err(i, "ds_read overwrites fragment buffer v%d before its consumer ran" % first)
                fr = dict(addr=addr, landed=False, used=False, epoch=self.epoch, issue_idx=steps)
                self.frag[first] = fr
                self.frag_history.append(fr)
                self.lgkm.append(("frag", first, addr))
                for r in range(n):
                    self.tag[("v", first + r)] = ("F", first)
            elif op == "ds_bpermute_b32":
                self.lgkm.append(("bperm",))
            elif op == "v_add_u32":
                self.reg[a[0]] = (self.val(a[1]) + self.val(a[2])) & 0xffffffff
            elif op == "v_mfma_f32_32x32x16_bf16":
                self.mfma(i, a)
            elif op in ("v_exp_f32", "v_mov_b32") and self.regs(a[0]) and self.regs(a[1]) and self.regs(a[1])[0] == "v":
                src = self.regs(a[1])[1]
                t = self.tag.get(("v", src))
                if t and t[0] == "S" and self.in_loop(i):
                    done = self.ready_at.get(src)
                    if done is None or self.n_mfma - done < 2:
                        self.err(i, "exponential reads score register v%d %s MFMAs behind the chain that wrote it (needs >= 2)" % (src, None if done is None else self.n_mfma - done))
                    self.tag[("v", self.regs(a[0])[1])] = ("E", t[1], t[2], src - G.SA[(t[1] & 1, t[2])])
                elif t is not None:
                    self.tag[("v", self.regs(a[0])[1])] = t if op == "v_mov_b32" else ("E?",)
            elif op == "v_cvt_pk_bf16_f32":
                lo, hi = self.tag.get(("v", self.regs(a[1])[1])), self.tag.get(("v", self.regs(a[2])[1]))
                dst = self.regs(a[0])[1]
                if self.in_loop(i):
                    if not (lo and hi and lo[0] == "E" and hi[0] == "E" and lo[1:3] == hi[1:3] and hi[3] == lo[3] + 1 and lo[3] % 2 == 0):
                        self.err(i, "cvt_pk packs %s and %s" % (lo, hi))
                    else:
                        self.tag[("v", dst)] = ("P", lo[1], lo[2], lo[3] // 2)
                else:
                    self.tag[("v", dst)] = ("P0", dst)
            # every other VALU instruction of the stream moves softmax values whose order the GPU bit-identity tests pin; the interpreter ignores them
        return self

    def in_loop(self, i):
        return i > self.labels[[k for k in self.labels if k.startswith("AQ2_LOOP")][0]]

    def mfma(self, i, a):
        dst, A, B, C = a
        self.n_mfma += 1
        fa = self.regs(A)
        is_qk = self.regs(dst) is not None and self.regs(dst)[0] == "v"
        if is_qk:
            half = 0 if self.regs(B)[1] < G.Q1 else 1
            n = self.qk_count[half]
            self.qk_count[half] += 1
            tile, inner = self.expect_k(n)
            self.check_fragment(i, "K", fa[1], tile, inner)
            qt = self.tag.get(("a", self.regs(B)[1]))
            if qt != ("Q", half, n % 8, 0):
                self.err(i, "QK^T MFMA %d of half %d multiplies Q fragment %s" % (n, half, qt))
            blk = n // 8
            d0 = self.regs(dst)[1]
            if d0 != G.SA[(blk & 1, half)]:
                self.err(i, "scores of block %d half %d written to v%d" % (blk, half, d0))
            first = (n % 8 == 0)
            if first and self.in_loop(i) and C != "v[%d:%d]" % (G.NEGM[half], G.NEGM[half] + 15):
                self.err(i, "a score chain in the loop must start from the -m block")
            if not first and C != dst:
                self.err(i, "score chain broken")
            if n % 8 == 7:
                for r in range(16):
                    self.tag[("v", d0 + r)] = ("S", blk, half)
                    self.ready_at[d0 + r] = self.n_mfma
            else:
                for r in range(16):
                    self.tag[("v", d0 + r)] = ("S-partial", blk, half)
        else:
            half = int(dst[3])
            n = self.pv_count[half]
            self.pv_count[half] += 1
            tile, inner, blk, g = self.expect_v(n)
            self.check_fragment(i, "V", fa[1], tile, inner)
            if dst != "%%[o%d%d]" % (half, g & 3) or C != dst:
                self.err(i, "PV MFMA %d accumulates into %s" % (n, dst))
            b0 = self.regs(B)[1]
            want = [("P", blk, half, 4 * (g >> 2) + w) for w in range(4)]
            got = [self.tag.get(("v", b0 + w)) for w in range(4)]
            if blk == 0:
                if b0 != G.PB[(0, half)] + 4 * (g >> 2):
                    self.err(i, "PV of block 0 takes its probabilities from v%d" % b0)
            elif blk <= 2 * self.nt - 1 and got != want:
                self.err(i, "PV MFMA of block %d half %d slab %d takes %s" % (blk, half, g >> 2, got))


def _simulate(cfg, nt, kbv=0):
    m = Machine(_stream_lines(cfg), cfg, nt, kbv)
    m.frag_history = []
    # the prologue's score MFMAs fill buffers four at a time and reuse them: mark those reads so that the overwrite check applies to the pipelined part only
    m.run()
    return m


ARMS = [G.DEFAULT] + [c for c in G.VARIANTS if not c.name.startswith("abl_")]


@pytest.mark.parametrize("cfg", ARMS, ids=[c.name for c in ARMS])
def test_stream_ring_discipline_waits_and_register_choreography(cfg):
    for nt in (1, 2, 3, 4, 5, 6, 7, 8, 11):
        for kbv in (0, 0x40400000):
            m = _simulate(cfg, nt, kbv)
            assert not m.errors, "%s, nt = %d: %d problems, first:\n  %s" % (cfg.name, nt, len(m.errors), "\n  ".join(m.errors[:6]))
            # the pipeline ran to the end: 2 nt blocks of scores + the tail's unused chain, PV of every real block
            assert m.qk_count[0] == m.qk_count[1] == 8 * (2 * nt + 2)
            assert m.pv_count[0] == m.pv_count[1] == 8 * 2 * nt
            assert m.epoch == nt + 1 and not m.vm and not m.lgkm


def test_the_interpreter_catches_seeded_bugs():
    """mutation check: the audit above must fail on streams with one wrong wait / slot / operand -- on the product stream (three tiles per trip, literal slots) and on the
    one-tile-per-trip arm (addresses stepped in registers)"""
    tile1 = next(c for c in G.VARIANTS if c.name == "tile1")

    def mutate(cfg, find, repl, nth=0):
        base = _stream_lines(cfg)
        loop = [i for i, l in enumerate(base) if l.startswith("AQ2_LOOP")][0]
        lines = list(base)
        hits = [i for i in range(loop, len(lines)) if find in lines[i]]
        assert len(hits) > nth, "the stream has no `%s` (occurrence %d) to mutate: update this test" % (find, nth)
        lines[hits[nth]] = lines[hits[nth]].replace(find, repl)
        m = Machine(lines, cfg, 7)
        m.frag_history = []
        try:
            m.run()
        except (AssertionError, KeyError):
            return ["crashed"]
        return m.errors
    for cfg in (G.DEFAULT, tile1):
        assert mutate(cfg, "s_waitcnt lgkmcnt(2)", "s_waitcnt lgkmcnt(3)", nth=5), "a wait one read too lax"
        assert mutate(cfg, "s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(4)"), "a DMA batch published before it was retired"
        assert mutate(cfg, "global_load_lds_dwordx4 %[ko0]", "global_load_lds_dwordx4 %[vo0]"), "K piece fetched through the V offsets"
        assert mutate(cfg, "v_exp_f32 v88, v226", "v_exp_f32 v88, v194"), "exponential of a score block that is still being accumulated"
        assert mutate(cfg, "v[128:131]", "v[132:135]", nth=0), "PV takes the other key slab's probabilities"
        lines = [l for l in _stream_lines(cfg) if l != "s_barrier"]
        m = Machine(lines, cfg, 5)
        m.frag_history = []
        m.run()
        assert m.errors, "no barrier at all"
    assert mutate(G.DEFAULT, "ds_read_b128 v[120:123], %[kx2] offset:16384", "ds_read_b128 v[120:123], %[kx2] offset:24576"), "wrong block of a K tile"
    assert mutate(G.DEFAULT, "s_add_u32 m0, %[ldsk], 16384", "s_add_u32 m0, %[ldsk], 0"), "K tile requested into the slot that is being read"
    assert mutate(G.DEFAULT, "%[vx0] offset:16384", "%[vx0] offset:0"), "V^T fragment from the previous trip's slot"
    assert mutate(tile1, "offset:8192", "offset:4096", nth=3), "wrong block of a K tile"
    assert mutate(tile1, "s_cselect_b32 s47, s52, s51", "s_cselect_b32 s47, s51, s51"), "K ring never wraps"
