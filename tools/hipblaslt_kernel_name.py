#!/usr/bin/env python
"""Which kernel does the vendor library (hipBLASLt through torch) pick for the FLUX GEMM shapes?  Run under
`rocprofv3 --kernel-trace --stats`: the kernel NAME encodes its macro tile, MFMA shape, workgroup size and staging scheme."""
import math, torch
dev = "cuda"
for (M, N, K) in [(13824, 21504, 3072), (50688, 3072, 3072), (13824, 3072, 15360)]:
    A = (torch.randn(M, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(A, B, bias)
    torch.cuda.synchronize()
