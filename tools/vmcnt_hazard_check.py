#!/usr/bin/env python
"""Static check of a hipcc -S listing (gfx950): does any instruction touch the destination registers of a vector-memory load that is still in
flight according to the in-order vmcnt model?

Why: the one-wave-per-SIMD GEMM and the attention kernels issue loads from inline asm and wait for them with literal `s_waitcnt vmcnt(N)` counts
the compiler knows nothing about.  hipcc believes an asm load's destination is defined when the asm statement ends, so it is free to COPY (live-range
split, phi copy across a back edge), SPILL or even re-use that register before the counted wait -- reading whatever the register held before.  Such a
copy is a timing-dependent wrong result: right whenever the load happens to have landed.  This tool walks every kernel in text order with the
hardware's counter model (loads, LDS-DMAs and stores retire in order through vmcnt; `s_waitcnt vmcnt(N)` leaves at most the N youngest in flight)
and reports every instruction that reads or writes a register of a load that may still be in flight.  Control flow is followed in text order only
(a label keeps the fall-through state, a backward branch is not re-walked), which is exact for the straight-line epilogues it was written for.
Only loads issued from INLINE ASM (between ;;#ASMSTART / ;;#ASMEND) are tracked as "in flight": a load the compiler emitted itself is waited for
by the compiler's own s_waitcnt insertion (which does follow control flow); every vector-memory operation still counts towards vmcnt.

Second check: an inline-asm store of more than 64 bits (global/buffer_store_dwordx3/x4) must be followed by two wait states before a VALU instruction
overwrites its data registers (the compiler pads this for its own stores, not for one inside an asm string: end the string with `s_nop 1`).

usage: vmcnt_hazard_check.py file.s [kernel-name-substring]      exit status 1 when something was flagged"""
import re
import sys

VM_OP = re.compile(r'^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)')
REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def walk(body, i0, i1, inflight, flagged, back_edges, labels):
    """text-order walk of body[i0:i1] starting with `inflight`; records backward branches (index of the branch, index of its label, state)"""
    in_asm = False
    pend_store = None
    for idx in range(i0, i1):
        ln, raw = body[idx]
        st = raw.strip()
        if st.startswith(';;#ASMSTART'):
            in_asm = True
        elif st.startswith(';;#ASMEND'):
            in_asm = False
        s = raw.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1))
                if n < len(inflight):
                    inflight = inflight[len(inflight) - n:] if n else []
            continue
        if op.startswith('s_endpgm'):
            inflight = []
            continue
        if back_edges is not None and (op.startswith('s_cbranch') or op == 's_branch'):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= idx and inflight:
                back_edges.append((idx, labels[tgt], list(inflight)))
        ops_txt = s[len(op):]
        touched = vregs(ops_txt)
        # wide asm store: its data registers must not be written by a VALU within two wait states
        if pend_store is not None:
            regs_, left_, l0_, t0_ = pend_store
            if op == 's_nop':
                left_ -= int(ops_txt.strip() or 0) + 1
            else:
                if op.startswith('v_') and not op.startswith('v_cmp') and vregs(ops_txt.split(',')[0]) & regs_:
                    flagged.append((ln, s, l0_, t0_, sorted(vregs(ops_txt.split(',')[0]) & regs_)))
                left_ -= 1
            pend_store = (regs_, left_, l0_, t0_) if left_ > 0 else None
        if in_asm and re.match(r'^(global|buffer|flat)_store_dwordx[34]', op):
            parts_ = ops_txt.split(',')
            pend_store = (vregs(parts_[1]) if op.startswith('global') or op.startswith('flat') else vregs(parts_[0]), 2, ln, s)
        if VM_OP.match(op) and '_load' in op and '_lds_' not in op:
            touched = vregs(','.join(ops_txt.split(',')[1:]))      # a load's own destination: loads retire in order, a second load into the same registers is ordered behind the first
        for (l0, t0, d0) in inflight:
            if d0 & touched:
                flagged.append((ln, s, l0, t0, sorted(d0 & touched)))
        if VM_OP.match(op):
            dest = set()
            is_load = ('_load' in op or ('atomic' in op and ' glc' in s)) and '_lds_' not in op
            if is_load and in_asm:
                first = ops_txt.split(',')[0]
                dest = vregs(first)
            inflight.append((ln, s, dest))
    return inflight


def check(body, name, verbose=False):
    labels = {}
    for idx, (ln, raw) in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', raw)
        if m:
            labels[m.group(1)] = idx
    flagged, back_edges = [], []
    walk(body, 0, len(body), [], flagged, back_edges, labels)
    # loop-carried loads: one more trip around every backward branch with the state the branch leaves behind
    for (bidx, lidx, state) in back_edges:
        walk(body, lidx, bidx, state, flagged, None, labels)
    seen, out = set(), []
    for f in flagged:
        if (f[0], f[2]) not in seen:
            seen.add((f[0], f[2]))
            out.append(f)
    return out


def main():
    t = open(sys.argv[1]).read().split('\n')
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    starts = [(i, m.group(1)) for i, l in enumerate(t) for m in [re.match(r'^(_Z\w+):', l)] if m]
    bad = 0
    for k, (i0, name) in enumerate(starts):
        if flt not in name:
            continue
        i1 = starts[k + 1][0] if k + 1 < len(starts) else len(t)
        body = []
        for i in range(i0 + 1, i1):
            if t[i].startswith('.Lfunc_end'):
                break
            body.append((i + 1, t[i]))
        fl = check(body, name)
        print("%s: %d instructions touch a register of a load that may be in flight" % (name, len(fl)))
        for ln, s, l0, t0, regs in fl[:40]:
            print("   line %d: %s\n        <- line %d: %s   (v%s)" % (ln, s, l0, t0, regs))
        bad += len(fl)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
