#!/usr/bin/env python
"""Per basic block of every kernel in a hipcc -S listing: instruction count, MFMAs, scratch (spill) accesses, SGPR-spill lane moves.
Shows at a glance whether spills sit in a hot loop (blocks with MFMAs) or in rarely executed set-up code.
usage: asm_blocks.py file.s [name-substring]"""
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z\w+):[^\n]*\n', t)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if flt not in name:
        continue
    body = body.split('.Lfunc_end')[0]
    blocks, cur = [], ['entry', 0, 0, 0, 0]
    for l in body.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur); cur = [m.group(1), 0, 0, 0, 0]
            continue
        ls = l.strip()
        if not ls or ls.startswith(';') or ls.startswith('.'):
            continue
        cur[4] += 1
        if 'v_mfma' in ls: cur[1] += 1
        if ls.startswith('scratch_'): cur[2] += 1
        if ls.startswith('v_writelane') or ls.startswith('v_readlane'): cur[3] += 1
    blocks.append(cur)
    print(name)
    print("   block            mfma scratch lane-moves instrs")
    for b in blocks:
        if b[1] or b[2] or b[3] > 4:
            print("   %-16s %4d %7d %10d %6d" % tuple(b))
