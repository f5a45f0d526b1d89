"""Kernel-level timing of the DiT hot kernels at FLUX shapes (GPU box only).
   python tools/perf_ops.py [--quick]"""
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unitex_amd import _lib
if "--ablate" in sys.argv:      # timing ablations (wrong results): libunitex_hip_ablate.so, options still come from UTX_* at utx_init
    _lib.use_ablation_library()
from unitex_amd.flux import ops

BF = torch.bfloat16


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    quick = "--quick" in sys.argv
    attn_only, gemm_only = "--attn-only" in sys.argv, "--gemm-only" in sys.argv
    dev = "cuda"
    H = 24
    for S in ([] if gemm_only else ([13824] if quick else [9728, 13824, 50688])):
        S_pad = (S + 63) // 64 * 64
        q = (torch.randn(H, S_pad, 128, device=dev) * (0.1275 if os.environ.get("UTX_PERF_PRESC", "1") == "1" else 1.0)).to(BF)
        k = torch.randn(H, S_pad, 128, device=dev).to(BF)
        vt = torch.randn(H, 128, S_pad, device=dev).to(BF)
        if os.environ.get("UTX_PERF_ZERO") == "1":   # zero operands: same instruction stream, far less switching power -> shows the clock-unconstrained rate
            q.zero_(); k.zero_(); vt.zero_()
        out = torch.empty(S, H * 128, dtype=BF, device=dev)
        presc = os.environ.get("UTX_PERF_PRESC", "1") == "1"
        med, best = timeit(lambda: ops.attention(q, k, vt, S=S, out=out, scale=0.0 if presc else None), iters=5 if S < 30000 else 3)
        fl = 4.0 * S * S * 128 * H
        print("attn S=%6d  med %8.3f ms  best %8.3f ms  -> %7.1f TF/s (%.1f%% of 2500)" % (S, med, best, fl / med / 1e9, fl / med / 1e9 / 25.0))
        del q, k, vt, out
    if attn_only:
        return
    shapes = [(13824, 9216, 3072), (13824, 3072, 3072), (13824, 12288, 3072), (13824, 3072, 12288),
              (13824, 21504, 3072), (13824, 3072, 15360), (512, 9216, 3072)]
    if not quick:
        shapes += [(50688, 21504, 3072), (50688, 3072, 15360)]
    for M, N, K in shapes:
        A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(BF)
        B = torch.randn(N, K, device=dev).to(BF)
        bias = torch.randn(N, device=dev).to(BF)
        C = torch.empty(M, N, dtype=BF, device=dev)
        med, best = timeit(lambda: ops.gemm(A, B, bias=bias, out=C))
        fl = 2.0 * M * N * K
        print("gemm M=%6d N=%6d K=%6d  med %8.3f ms  best %8.3f -> %7.1f TF/s" % (M, N, K, med, best, fl / med / 1e9))
        # library reference point (hipBLASLt through torch) on the same data
        med2, _ = timeit(lambda: torch.nn.functional.linear(A, B, bias))
        print("      torch/hipBLASLt reference            med %8.3f ms             -> %7.1f TF/s" % (med2, fl / med2 / 1e9))
        del A, B, C
    # elementwise
    S, D = 13824, 3072
    x = torch.randn(S, D, device=dev).to(BF); sh = torch.randn(D, device=dev).to(BF); sc = torch.randn(D, device=dev).to(BF)
    y = torch.empty_like(x)
    med, _ = timeit(lambda: ops.ln_mod(x, sh, sc, out=y))
    print("ln_mod S=%d: %.3f ms -> %.0f GB/s" % (S, med, 2 * S * D * 2 / med / 1e6))


if __name__ == "__main__":
    main()
