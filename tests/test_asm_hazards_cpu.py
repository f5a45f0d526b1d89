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
