#!/usr/bin/env python
"""Census of the GEMM launches of one denoise step (default bench workload) by shape and by the kernel they dispatch to, and the time of each
unique shape measured in isolation (events, median of 7): where the 128x128 kernel's 6 % of the step goes."""
import collections, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from unitex_amd.flux import ops
from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
from unitex_amd.flux.transformer import FluxDiT, FluxShape
dev = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "strip1024x6"
h_px, w_px, dual = (1024, 6144, 512) if wl == "strip1024x6" else (512, 3072, 512)
shape = FluxShape(); sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
model = FluxDiT(sd, shape, device=dev)
model.set_lora([(synthetic_lora(sd, shape, rank=64, seed=1, device=dev), 1.0), (synthetic_lora(sd, shape, rank=64, seed=2, device=dev), 0.0)])
HL, WL = h_px // 16, w_px // 16
S_txt, S_img = 512, 2 * HL * WL + (dual // 16) ** 2
model.set_positions(torch.zeros(S_txt, 3), torch.zeros(S_img, 3))
model.set_conditioning(torch.zeros(S_txt, shape.joint_dim, device=dev), torch.zeros(1, shape.pooled_dim, device=dev), 3.5)
plan = next(iter(model._plans.values()))["plan"]
gemm_fn = model.lib.utx_gemm_bf16
cnt = collections.Counter(); side = collections.Counter()
def walk(ops_, on_side):
    for fn, d in ops_:
        if fn == "par":
            walk(d[0], False); walk(d[1], True)
        elif fn is gemm_fn or (hasattr(fn, "__name__") and getattr(fn, "__name__", "") == "utx_gemm_bf16"):
            key = (d.M, d.N, d.K, d.K2, bool(d.gate), d.gelu_from < d.N, d.n_split < d.N)
            cnt[key] += 1
            if on_side: side[key] += 1
walk(plan, False)
def kernel_of(M, N, K, K2):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    return "pers256" if (N % 256 == 0 and tiles >= 192) else "128x128"
def t1(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
tot = collections.Counter()
print("%-40s %5s %5s %-8s %9s %9s" % ("M N K K2 gate gelu split", "calls", "side", "kernel", "us/call", "ms/step"))
for key, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    M, N, K, K2, gate, gelu, split = key
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device=dev); kw = {}
    if K2: kw.update(A2=torch.randn(M, K2, device=dev).to(torch.bfloat16), B2=torch.randn(N, K2, device=dev).to(torch.bfloat16), lora_n_limit=N, lora_seg_n=N)
    if gate: kw.update(gate=torch.randn(N, device=dev).to(torch.bfloat16), res=Cc)
    if gelu: kw.update(gelu_from=0)
    f = lambda: ops.gemm(A, B, out=Cc, bias=torch.zeros(N, device=dev, dtype=torch.bfloat16), **kw)
    f(); f(); ts = sorted(t1(f) for _ in range(7)); us = ts[3] * 1e3
    k = kernel_of(M, N, K, K2); tot[k] += us * n * 1e-3
    print("%-40s %5d %5d %-8s %9.1f %9.2f" % (" ".join(str(x) for x in key), n, side[key], k, us, us * n * 1e-3), flush=True)
print({k: round(v, 1) for k, v in tot.items()}, "ms per step (isolated timings)")
