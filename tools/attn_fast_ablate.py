#!/usr/bin/env python
"""Cost account of the attention kernel's fast loop (attention_glds.hip FAST; UTX_ATTN_VAR = 32 + bits in the ABLATION library: WRONG results by design):
1 v_exp -> move, 2 no fragment reads in the loop, 4 no DMA in the loop, 8 no barrier in the loop, 16 no softmax VALU.  Same process, interleaved; random operands
(the chip is power-limited: zeros would clock higher).  python unitex_amd/csrc/build.py --ablate first."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
_lib.use_ablation_library()
from unitex_amd.flux import ops
BF, H = torch.bfloat16, 24
ARMS = [int(x) for x in os.environ.get("UTX_ABL_ARMS", "32,33,34,64,36,40,44,46,65,66,67").split(",")]
NAMES = {32: "fast loop as it is (unsplit tail)", 33: "v_exp -> v_mov", 34: "no fragment reads", 36: "no DMA", 40: "no barrier", 48: "no softmax VALU", 44: "no DMA, no barrier",
         46: "no reads / DMA / barrier (MFMA + softmax)", 50: "no reads, no softmax VALU", 62: "MFMA only", 64: "reads issued, not waited for / not consumed", 65: "unrolled by two (correct results)", 66: "unrolled by two, no fragment reads", 67: "unrolled by two, reads issued not consumed"}
for S in tuple(int(x) for x in os.environ.get("UTX_AB_SIZES", "50240").split(",")):
    g = torch.Generator(device="cuda").manual_seed(S)
    Qh = (torch.randn(H, S, 128, generator=g, device="cuda") * (1.4426950408889634 / math.sqrt(128.0))).to(BF)
    Kh = torch.randn(H, S, 128, generator=g, device="cuda").to(BF)
    Vt = torch.randn(H, 128, S, generator=g, device="cuda").to(BF)
    out = torch.empty(S, H * 128, dtype=BF, device="cuda")
    fl = 4.0 * S * S * 128 * H
    times = {a: [] for a in ARMS}
    for _ in range(int(os.environ.get("UTX_AB_ROUNDS", "4"))):
        for a in ARMS:
            _lib.set_option("UTX_ATTN_VAR", a)
            ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(2): ops.attention(Qh, Kh, Vt, S=S, scale=0.0, out=out)
            e1.record(); torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / 2)
    nt = (S + 63) // 64
    rounds = math.ceil(((S + 255) // 256) * H / 256)
    for a in ARMS:
        t = sorted(times[a]); med = t[len(t) // 2]
        print("S=%6d VAR %2d %-44s med %8.3f ms -> %7.1f TF/s  (%.3f us per tile-step of a workgroup)" % (S, a, NAMES.get(a, ""), med, fl / med / 1e9, med * 1e3 / rounds / nt), flush=True)
_lib.set_option("UTX_ATTN_VAR", 0)
