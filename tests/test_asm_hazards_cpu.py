"""Static audit of every build's device code for the hazards hipcc cannot see inside inline asm (CPU test: hipcc cross-compiles gfx950 listings).

The hand-scheduled kernels issue loads from inline asm and wait for them with literal `s_waitcnt vmcnt(N)` counts.  hipcc takes an asm load's
destination for defined when the statement ends, so it may copy / spill / reuse that register before the counted wait -- a timing-dependent wrong
result (round 3's one-off one-ulp difference of the fused q / k epilogue was exactly that: `v_mov_b64` of the cos / sin registers in front of the
ragged-tile branch's wait).  tools/vmcnt_hazard_check.py walks the listing with the hardware's in-order vmcnt model (tools/vmcnt_order_probe.hip
measures that model on the GPU) and reports every instruction that touches a register of a load still in flight, and every wide asm store whose data
registers are overwritten within its two wait states."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vmcnt_hazard_check as hz  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _check_text(text, name="_Zsynthetic"):
    body = [(i + 1, l) for i, l in enumerate(text.split("\n"))]
    return hz.check(body, name)


def test_checker_flags_a_copy_in_front_of_the_wait_and_accepts_it_behind():
    bad = """
	;;#ASMSTART
	global_load_dwordx4 v[48:51], v4, s[40:41]
	;;#ASMEND
	v_add_u32_e32 v7, v8, v9
	v_mov_b64_e32 v[64:65], v[48:49]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mul_f32_e32 v1, v64, v2
"""
    good = bad.replace("	v_mov_b64_e32 v[64:65], v[48:49]\n	;;#ASMSTART\n	s_waitcnt vmcnt(0)\n	;;#ASMEND\n",
                       "	;;#ASMSTART\n	s_waitcnt vmcnt(0)\n	;;#ASMEND\n	v_mov_b64_e32 v[64:65], v[48:49]\n")
    assert len(_check_text(bad)) == 1
    assert _check_text(good) == []
    # two younger operations behind the load: vmcnt(3) waits for nothing, vmcnt(2) retires exactly the load that is read
    counted = """
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v4, s[40:41]
	;;#ASMEND
	global_store_dwordx4 v5, v[20:23], s[2:3]
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v4, s[42:43]
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(%d)
	;;#ASMEND
	v_mov_b32_e32 v30, v10
"""
    assert len(_check_text(counted % 3)) == 1
    assert _check_text(counted % 2) == []
    # loop-carried: the load of one trip is consumed at the top of the next without a wait
    loop = """
.LBB0_1:
	v_add_f32_e32 v1, v10, v1
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v4, s[40:41]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
"""
    assert len(_check_text(loop)) == 1
    # a > 64-bit asm store followed at once by a VALU write of its data registers; s_nop 1 in the string cures it
    store = """
	;;#ASMSTART
	global_store_dwordx4 v5, v[20:23], s[2:3]%s
	;;#ASMEND
	v_mov_b32_e32 v21, 0
"""
    assert len(_check_text(store % "")) == 1
    assert _check_text(store % "\n	s_nop 1") == []


@pytest.mark.timeout(900)
def test_no_kernel_of_the_build_touches_a_register_of_a_load_in_flight(tmp_path):
    from unitex_amd.csrc import build as b
    srcs = [(n, x) for n, x in b.SOURCES if n.endswith(".hip") and "asm volatile" in open(os.path.join(b.HERE, n)).read()]
    assert {"gemm_w4.hip", "gemm_pers.hip", "attention_glds.hip"} <= {n for n, _ in srcs}

    def listing(item):
        name, extra = item
        out = str(tmp_path / (name + ".s"))
        cmd = [HIPCC] + b.COMMON + extra + ["-S", "--cuda-device-only", os.path.join(b.HERE, name), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return name, out
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        outs = list(ex.map(listing, srcs))
    report, kernels = [], 0
    for name, path in outs:
        t = open(path).read().split("\n")
        import re
        starts = [(i, m.group(1)) for i, l in enumerate(t) for m in [re.match(r"^(_Z\w+):", l)] if m]
        for k, (i0, kn) in enumerate(starts):
            i1 = starts[k + 1][0] if k + 1 < len(starts) else len(t)
            body = []
            for i in range(i0 + 1, i1):
                if t[i].startswith(".Lfunc_end"):
                    break
                body.append((i + 1, t[i]))
            kernels += 1
            for ln, s_, l0, t0, regs in hz.check(body, kn):
                report.append("%s %s line %d: %s  <- line %d: %s (v%s)" % (name, kn, ln, s_, l0, t0, regs))
    assert kernels >= 20
    # scratch: the attention kernel's fast loop (two tiles per trip, 256 VGPRs) parks a few loop-invariant dwords in scratch AROUND its loop; no hot loop of the
    # attention kernels may touch scratch itself
    for name, path in outs:
        if not name.startswith("attention"):
            continue
        in_loop = False
        for l in open(path):
            if "Loop Header" in l or "in Loop:" in l:
                in_loop = True
            elif l.startswith(".LBB") or l.startswith("_Z"):
                in_loop = False
            assert not (in_loop and "scratch_" in l), "%s: scratch access inside a loop: %s" % (name, l.strip())
    assert not report, "registers of in-flight asm loads are touched before their wait:\n" + "\n".join(report[:20])


def _packed_cross_half_pairs(path, window=3):
    """(kernel, line number, text) of every packed-fp32 instruction whose LOW lane takes the HIGH half (op_sel bit 1 for that source) of a register pair that another
    packed-fp32 instruction wrote within the previous `window` instructions -- the pattern round 5's two-stream corruption was traced to (csrc/build.py NO_PK)."""
    import re

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return (int(m.group(1)), int(m.group(2)))
        m = re.match(r"v(\d+)$", tok)
        return (int(m.group(1)), int(m.group(1))) if m else None
    out, cur, recent = [], None, []
    for ln, l in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, recent = m.group(1), []
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_pk_") and op.endswith("_f32"):
            ops = [x.strip() for x in t[len(op):].split(",")]
            dst = regs(ops[0].split()[0])
            srcs = [regs(x.split()[0]) for x in ops[1:4]]
            msel = re.search(r"op_sel:\[([01,]+)\]", t)
            sel = [int(x) for x in msel.group(1).split(",")] if msel else []
            for i, sr in enumerate(srcs):
                if sr and i < len(sel) and sel[i] == 1 and any(d == sr for d, _ in recent):
                    out.append((cur, ln, t))
            recent = [(dst, 0)] + [(d, a + 1) for d, a in recent if a + 1 < window]
        else:
            recent = [(d, a + 1) for d, a in recent if a + 1 < window]
    return out


def test_the_packed_pattern_checker_flags_the_library_stream_that_failed(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_Zk:\n\tv_pk_mul_f32 v[54:55], v[6:7], v[50:51] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[50:51], v[10:11], v[50:51] op_sel_hi:[0,1]\n\tv_cvt_pk_bf16_f32 v46, v46, v47\n"
                   "\tv_pk_add_f32 v[56:57], v[54:55], v[50:51] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n")
    assert len(_packed_cross_half_pairs(str(bad))) == 1
    ok = tmp_path / "ok.s"
    ok.write_text("_Zk:\n\tv_pk_mul_f32 v[46:47], v[6:7], v[50:51]\n\tv_pk_mul_f32 v[48:49], v[10:11], v[52:53]\n\tv_pk_add_f32 v[46:47], v[46:47], v[48:49]\n")
    assert _packed_cross_half_pairs(str(ok)) == []


@pytest.mark.timeout(900)
def test_no_kernel_of_the_build_holds_the_dependent_packed_pair_that_lost_a_product(tmp_path):
    """every .hip translation unit, with the flags of csrc/build.py: a TU in which hipcc forms the pattern has to be built with NO_PK"""
    from unitex_amd.csrc import build as b
    srcs = [(n, x) for n, x in b.SOURCES if n.endswith(".hip")]

    def listing(item):
        name, extra = item
        out = str(tmp_path / (name + ".s"))
        r = subprocess.run([HIPCC] + b.COMMON + extra + ["-S", "--cuda-device-only", os.path.join(b.HERE, name), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return name, out
    with ThreadPoolExecutor(max_workers=8) as ex:
        outs = list(ex.map(listing, srcs))
    found = [(name,) + hit for name, path in outs for hit in _packed_cross_half_pairs(path)]
    assert not found, "dependent packed-fp32 pairs across the halves (build the TU with NO_PK):\n" + "\n".join("%s %s line %d: %s" % f for f in found[:20])


def _barriers_without_vmcnt0(path, window=48):
    """(kernel, line, context) of every s_barrier of a kernel that uses LDS-DMA (global_load_lds / buffer_load ... lds) which has no `s_waitcnt ... vmcnt(0)` among the
    `window` instructions in front of it.  An LDS-DMA'd tile is published by: this wave's vmcnt(0), THEN the barrier, then the reads (cdna_hip_programming.md 5.7); hipcc does
    not owe the DMA that wait at __syncthreads() -- round 5 found the attention kernels relying on it (csrc/attention_glds.hip, AG_BARRIER)."""
    import re
    out, cur, body = [], None, []

    def flush():
        if cur is None or not any("global_load_lds" in t or (t.startswith("buffer_load") and " lds" in t) for _, t in body):
            return
        for i, (ln, t) in enumerate(body):
            if t.split()[0] != "s_barrier":
                continue
            # walk back from the barrier: a vmcnt(0) must come before any instruction that issues vector memory traffic (VALU / SALU / LDS in between are harmless)
            ok, prev = False, []
            for _, x in reversed(body[max(0, i - window):i]):
                prev.append(x)
                if x.startswith("s_waitcnt") and "vmcnt(0)" in x:
                    ok = True
                    break
                if x.split()[0].startswith(("global_", "buffer_", "flat_", "scratch_")):
                    break
            if not ok:
                out.append((cur, ln, " | ".join(reversed(prev[:3]))))
    for ln, l in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            flush()
            cur, body = m.group(1), []
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        body.append((ln, re.sub(r"\s+", " ", t)))
    flush()
    return out


def test_the_barrier_checker_flags_a_fence_without_its_vmcnt(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_Zk:\n\tglobal_load_lds_dwordx4 v[0:1], off\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_endpgm\n")
    assert len(_barriers_without_vmcnt0(str(bad))) == 1
    ok = tmp_path / "ok.s"
    ok.write_text("_Zk:\n\tglobal_load_lds_dwordx4 v[0:1], off\n\ts_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_endpgm\n_Zplain:\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n")
    assert _barriers_without_vmcnt0(str(ok)) == []


@pytest.mark.timeout(900)
def test_every_barrier_of_the_attention_kernels_is_behind_a_vmcnt0(tmp_path):
    """the three LDS-DMA attention translation units (their barriers all publish DMA'd tiles), with the flags of csrc/build.py AND with the scheduling strategy that exposed the
    missing wait in round 5 (-mllvm -amdgpu-sched-strategy=max-memory-clause)"""
    from unitex_amd.csrc import build as b
    srcs = [(n, x) for n, x in b.SOURCES if n in ("attention_glds.hip", "attention_fp8.hip", "attention_q64.hip")]
    assert len(srcs) == 3
    jobs = [(n, x, tag, extra) for n, x in srcs for tag, extra in (("default", []), ("maxmem", ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]))]

    def listing(job):
        name, flags, tag, extra = job
        out = str(tmp_path / (name + "." + tag + ".s"))
        r = subprocess.run([HIPCC] + b.COMMON + flags + extra + ["-S", "--cuda-device-only", os.path.join(b.HERE, name), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return name + " (" + tag + ")", out
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(listing, jobs))
    found = [(name,) + hit for name, path in outs for hit in _barriers_without_vmcnt0(path)]
    assert not found, "barriers of LDS-DMA kernels without a vmcnt(0) in front of them:\n" + "\n".join("%s %s line %d: ... %s" % f for f in found[:20])
