#!/usr/bin/env python
"""Race screen of the MX fp8 one-wave-per-SIMD GEMM (gemm_w4.hip, MX): hand-counted vmcnt waits, asm MFMAs the compiler's hazard recogniser does not see,
scale registers loaded by asm -- a schedule bug would show as rare, run-dependent wrong bits.  Both MX kernels accumulate every output over ascending K with
the same instruction, so on every shape, repetition and epilogue (bias; bias + GELU + column split; gated residual; GELU tiles leaving as fp8) the
persistent kernel must equal the 128 x 128-tile MX kernel BIT FOR BIT, and repeated launches on the same operands must equal each other, also with
other work in flight on a second stream (clock / arrival-time perturbation).  usage: python tools/mx_race_screen.py [reps]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops, mx8
ctx = ops.get_ctx(0)
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_lib.set_option("UTX_GEMM_STREAMK", 0)      # the split tail changes fp32 summation order by design
shapes = [(256, 256, 128), (300, 512, 256), (1100, 512, 512), (2048, 3072, 3072), (13824, 3072, 3072), (13376, 9216, 3072), (13824, 3072, 15360),
          (50688, 3072, 3072), (50240, 12288, 3072), (50688, 3072, 12288), (34304, 21504, 3072)]
side = torch.cuda.Stream()
noise_a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K) in shapes:
    for r in range(reps):
        g = torch.Generator(device=dev).manual_seed(7919 * r + M % 977 + N + K)
        A = (torch.randn(M, K, device=dev, generator=g) * math.exp(float(torch.randn(1, generator=torch.Generator().manual_seed(r))))).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
        bias = (torch.rand(N, device=dev, generator=g) - 0.5).to(torch.bfloat16)
        gate = torch.randn(N, device=dev, generator=g).to(torch.bfloat16); res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        aq, a_s = mx8.quantize_act(A, ctx); wq, w_s = mx8.quantize_weight(B, ctx)
        _, a_p = mx8.quantize_act(A, ctx, packed=True); _, w_p = mx8.quantize_weight(B, ctx, packed=True)
        split = N // 2 if (N // 2) % 256 == 0 else 0
        def run(sa, sb, fp8_out=False):
            outs = []
            C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm(aq, wq, bias=bias, out=C, a_scale=sa, b_scale=sb, sk_work=None); outs.append(C)
            if split:
                c0 = torch.zeros(M, split, device=dev, dtype=torch.bfloat16); c1 = torch.zeros(M, N - split, device=dev, dtype=torch.bfloat16)
                kw = {}
                if fp8_out:
                    qo = torch.zeros(M, N - split, dtype=torch.uint8, device=dev); so = mx8.packed_scale_buffer(M, N - split, dev)
                    kw["q_out"] = (qo, so, 0)
                ops.gemm(aq, wq, bias=bias, out=c0, gelu_from=split, n_split=split, C1=c1, a_scale=sa, b_scale=sb, sk_work=None, **kw)
                outs.append(c0)
                if fp8_out:
                    outs += [qo, so]
                else:
                    outs.append(c1)
            R = res.clone()
            ops.gemm(aq, wq, bias=bias, out=R, gate=gate, res=R, a_scale=sa, b_scale=sb, sk_work=None); outs.append(R)
            return outs
        ref = run(a_s, w_s)
        for trial in range(3):
            if trial == 2:      # perturb: a big bf16 GEMM on a second stream beside the launches
                with torch.cuda.stream(side):
                    torch.matmul(noise_a, noise_a)
            got = run(a_p, w_p)
            torch.cuda.synchronize()
            for i, (x, y) in enumerate(zip(got, ref)):
                if not torch.equal(x, y):
                    d = (x.float() - y.float()).abs()
                    print("MISMATCH M=%d N=%d K=%d rep=%d trial=%d output %d: %d elems, max %.4g" % (M, N, K, r, trial, i, int((d > 0).sum()), float(d.max())))
                    bad += 1
        if split:       # fp8 output of the GELU tiles == quantiser applied to the bf16 output
            got = run(a_p, w_p, fp8_out=True)
            c1_ref = ref[2]
            q_ref, s_ref = mx8.quantize_act(c1_ref, ctx, packed=True)
            torch.cuda.synchronize()
            if not (torch.equal(got[2], q_ref) and torch.equal(got[3][: (N - split) // 128], s_ref.data[: (N - split) // 128])):
                print("MISMATCH M=%d N=%d K=%d rep=%d: fp8 output of the GELU tiles differs from bf16 + quantiser" % (M, N, K, r)); bad += 1
    print("shape M=%6d N=%6d K=%6d : %d reps x 3 trials screened" % (M, N, K, reps), flush=True)
print("MX RACE SCREEN:", "FAILED (%d)" % bad if bad else "clean")
_lib.set_option("UTX_GEMM_STREAMK", 1)
