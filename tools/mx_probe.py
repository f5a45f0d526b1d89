#!/usr/bin/env python
"""Determine the operand and scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 empirically (the ISA document is not in the image).
Step 1: unit scales, random e4m3 data -> which (lane half, byte) <-> k association A and B share.  Step 2: all-ones data and ONE
lane's scale byte changed at a time -> which output rows / how many k that (lane, byte) scales, for every op_sel."""
import ctypes as C, os
import numpy as np, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "mx_probe.so"))
g = torch.Generator().manual_seed(0)
def run(A8, B8, sa, sb, opsel):
    d = torch.zeros(64, 16, device="cuda")
    a_d, b_d = A8.view(torch.uint8).cuda().contiguous(), B8.view(torch.uint8).cuda().contiguous()
    sa_d, sb_d = sa.cuda().contiguous(), sb.cuda().contiguous()
    assert lib.mx_probe(C.c_void_p(a_d.data_ptr()), C.c_void_p(b_d.data_ptr()), C.c_void_p(sa_d.data_ptr()), C.c_void_p(sb_d.data_ptr()),
                        C.c_void_p(d.data_ptr()), opsel, opsel) == 0
    D = d.cpu().numpy(); out = np.zeros((32, 32))
    for l in range(64):
        for r in range(16):
            out[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = D[l, r]
    return out
ones_s = torch.full((64, 4), 127, dtype=torch.uint8)
A8 = torch.randn(64, 32, generator=g).to(torch.float8_e4m3fn); B8 = torch.randn(64, 32, generator=g).to(torch.float8_e4m3fn)
out = run(A8, B8, ones_s, ones_s, 0)
Af, Bf = A8.float().numpy().astype(np.float64), B8.float().numpy().astype(np.float64)
# pairing (h, j) <-> (h, j): D[i][j] = sum over lanes with row i (both halves) and bytes
ref = np.zeros((32, 32))
for i in range(32):
    for j in range(32):
        ref[i, j] = (Af[i] * Bf[j]).sum() + (Af[i + 32] * Bf[j + 32]).sum()
print("unit scales, same-position pairing: D err %.3g, D^T err %.3g" % (np.abs(out - ref).max(), np.abs(out - ref.T).max()), flush=True)
one = torch.full((64, 32), 0x38, dtype=torch.uint8).view(torch.float8_e4m3fn)      # 1.0
for opsel in (0, 1, 2, 3):
    base = run(one, one, ones_s, ones_s, opsel)
    assert np.all(base == 64.0), base
    hits = {}
    for lane in range(64):
        for byte in range(4):
            sa = ones_s.clone(); sa[lane, byte] = 128
            o = run(one, one, sa, ones_s, opsel)
            diff = o - 64.0
            rows = np.nonzero(np.abs(diff).sum(1))[0]
            if len(rows):
                hits[(lane, byte)] = (rows.tolist(), float(diff[rows[0], 0]))
    # summarise
    by_byte = {}
    for (lane, byte), (rows, amt) in hits.items():
        by_byte.setdefault(byte, []).append((lane, rows, amt))
    print("opsel", opsel, "scale_A: bytes that matter:", sorted(by_byte), flush=True)
    for byte, lst in sorted(by_byte.items()):
        print("   byte", byte, "lanes:", [(l, r, a) for l, r, a in lst][:6], "... (%d lanes)" % len(lst), flush=True)
    # the same for scale_B (rows of B = output columns)
    hitsb = {}
    for lane in (0, 5, 31, 32, 37, 63):
        for byte in range(4):
            sb = ones_s.clone(); sb[lane, byte] = 128
            o = run(one, one, ones_s, sb, opsel)
            diff = o - 64.0
            cols = np.nonzero(np.abs(diff).sum(0))[0]
            if len(cols):
                hitsb[(lane, byte)] = (cols.tolist(), float(diff[0, cols[0]]))
    print("   scale_B sample:", hitsb, flush=True)
# which k does a lane's scale cover?  impulse in A at (lane la, byte ja), scale of lane ls doubled
for (ls, bs) in ((0, 0), (32, 0)):
    cov = []
    for la in (0, 32):
        for ja in range(32):
            Aimp = torch.zeros(64, 32, dtype=torch.uint8); Aimp[la, ja] = 0x38
            sa = ones_s.clone(); sa[ls, bs] = 128
            o = run(Aimp.view(torch.float8_e4m3fn), one, sa, ones_s, 0)
            cov.append((la, ja, float(o[0, 0])))
    print("scale lane", ls, "byte", bs, "doubles A elements (lane, byte):", [(la, ja) for la, ja, v in cov if v == 2.0], flush=True)
