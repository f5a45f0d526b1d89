#!/usr/bin/env python
"""Whole-chip sustained MFMA rate on random vs zero operands by instruction shape (tools/mfma_power_probe.hip)."""
import ctypes as C, os
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "mfma_power_probe.so"))
lib.mfma_burn.restype = C.c_double
for mode, name in ((0, "32x32x16 bf16"), (1, "16x16x32 bf16"), (2, "32x32x64 MX fp8")):
    for threads in (512, 256):
        for zero in (0, 1):
            for nwg in (256, 64):
                tf = lib.mfma_burn(mode, threads, 20000, zero, nwg)
                print("%-16s %d waves/SIMD  %-6s operands  %3d CUs busy: %7.1f TF/s (%.1f per CU)" % (name, threads // 256, "zero" if zero else "random", nwg, tf, tf / nwg), flush=True)
