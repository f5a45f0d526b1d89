"""Instruction-class stream of one kernel's hot blocks, from hipcc's assembly (-S --cuda-device-only): one letter per
instruction, one line per basic block.  What the attention / GEMM schedules were tuned with: it shows at a glance whether
the VALU / LDS fillers really sit between the MFMAs, where hipcc inserted s_nop or AGPR copies, and what spilled.

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iunitex_amd/csrc -Iinclude -x hip -S --cuda-device-only \
        unitex_amd/csrc/attention_q64.hip -o /tmp/q64.s
    python tools/asm_stream.py /tmp/q64.s attn_fwd_q64_kernelILi1ELi0E [--hist]

M mfma   E v_exp   c v_cvt_pk   a v_accvgpr_*   v other VALU   D ds_read   G global/buffer load   W s_waitcnt   B s_barrier
n s_nop  . other SALU   ? anything else (scratch, stores ...)   >NN branch to block NN
--hist: opcode histogram of the first loop (from its "Loop Header" block to the closing s_cbranch_scc*)."""
import collections
import sys


def kernel_lines(path, name_part):
    out, on = [], False
    for l in open(path):
        t = l.rstrip("\n")
        if not on and t.startswith("_Z") and name_part in t and t.split(":")[0].endswith(t.split(":")[0]) and ":" in t:
            on = True
        if on:
            out.append(t)
            if t.strip().startswith("s_endpgm"):
                break
    return out


def classify(op, tail):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("v_exp"):
        return "E"
    if op.startswith("ds_read"):
        return "D"
    if op.startswith("v_accvgpr"):
        return "a"
    if op.startswith("v_cvt_pk"):
        return "c"
    if op.startswith("v_"):
        return "v"
    if op.startswith("s_waitcnt"):
        return "W"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return ">" + tail[-3:] + " "
    if op.startswith("global_load") or op.startswith("buffer_load"):
        return "G"
    if op.startswith("s_nop"):
        return "n"
    if op.startswith("s_"):
        return "."
    return "?"


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    if not lines:
        raise SystemExit("kernel symbol containing %r not found" % sys.argv[2])
    if "--hist" in sys.argv:
        hist, on = collections.Counter(), False
        for t in lines:
            s = t.strip()
            if "Loop Header" in t:
                on = True
            if on and s and not s.startswith(";") and not s.startswith("."):
                hist[s.split()[0]] += 1
                if s.startswith("s_cbranch_scc"):
                    break
        for op, n in hist.most_common(40):
            print("%6d  %s" % (n, op))
        return
    cur = ""
    for t in lines:
        s = t.strip()
        if not s or s.startswith(";"):
            continue
        if s.startswith(".LBB"):
            print(cur)
            cur = s.split(":")[0] + ": "
            continue
        parts = s.split()
        cur += classify(parts[0], parts[-1])
    print(cur)


if __name__ == "__main__":
    main()
