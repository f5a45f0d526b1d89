#!/usr/bin/env python
"""Timeline of one workgroup of gemm256_w4_kernel (ablation build, ABL 144 = trace + no C stores): duration of every sub-stage and of
every epilogue, effective shader clock."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
import numpy as np
dev = "cuda"
M, N, K = 50688, 3072, 3072
A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
bias = torch.randn(N, device=dev).to(torch.bfloat16); C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
_lib.set_option("UTX_GEMM_TILE", 2564)
from unitex_amd.flux.ops import make_gemm_desc, ptr
import ctypes as C_
lib = _lib.load_library(); ctx = ops.get_ctx(0)
trace = torch.zeros(3 * 4000, dtype=torch.int64, device=dev)
for wg, abl in ((0, 128), (0, 144), (100, 128)):
    print("ABL", abl, "workgroup", wg); _lib.set_option("UTX_GEMM_DEBUG", abl << 5); _lib.set_option("UTX_GEMM_PERS_SCHED", 100 + wg)
    d = make_gemm_desc(A, B, C, bias=bias)
    d.zero_page = ptr(trace)
    for _ in range(3):
        trace.zero_()
        rc = lib.utx_gemm_bf16(ctx.handle, C_.byref(d), ctx.stream()); assert rc == 0, rc
        torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 3)
    n = int(np.argmax((t[:, 1] == 0)) if (t[:, 1] == 0).any() else len(t))
    t = t[:n]
    tag, wall, cyc = t[:, 0], (t[:, 1] - t[0, 1]) * 0.01, t[:, 2] - t[0, 2]
    print("workgroup %d: %d records, total %.1f us, shader clock / 100 MHz clock: %.3f GHz overall" % (wg, n, wall[-1], cyc[-1] / max(wall[-1], 1e-9) * 1e-3))
    # per tile: sub-stage durations, epilogue duration
    idx_e0 = np.where(tag == -1)[0]
    for k, i in enumerate(idx_e0[:10]):
        start = 0 if k == 0 else idx_e0[k - 1] + 2
        ss = np.diff(wall[start:i + 1])          # sub-stage durations of this tile (last = up to the epilogue start)
        cc = np.diff(cyc[start:i + 1])
        epi = wall[i + 1] - wall[i]
        print("  tile %d: %3d K-tiles, mean %.3f us (first 4: %s, last 4: %s), cycles/K-tile mean %.0f -> %.2f GHz | epilogue %.2f us (%d cycles)" % (
            k, len(ss), ss.mean(), " ".join("%.2f" % x for x in ss[:4]), " ".join("%.2f" % x for x in ss[-4:]), cc.mean(), cc.sum() / ss.sum() * 1e-3, epi, cyc[i + 1] - cyc[i]))
