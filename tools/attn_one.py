#!/usr/bin/env python
"""The bench's attention launch in isolation (PMC passes serialise kernels: profiling the whole bench takes minutes per pass):
24 heads, the executed sequence of the default workload (text dedup: 64 + 50176 tokens, key weight 2^3 on tile 0), random data.
UTX_ONE_ABLATE=1 binds the ablation library (UTX_ATTN_VAR arms: wrong results by design; their launches carry no key multiplicity)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
abl = os.environ.get("UTX_ONE_ABLATE", "0") == "1"
if abl:
    _lib.use_ablation_library()
from unitex_amd.flux import ops
S, H = int(sys.argv[1]) if len(sys.argv) > 1 else 50240, 24
q = (torch.randn(H, S, 128, device="cuda") * 0.1275).to(torch.bfloat16); k = torch.randn(H, S, 128, device="cuda").to(torch.bfloat16)
vt = torch.randn(H, 128, S, device="cuda").to(torch.bfloat16); out = torch.empty(S, H * 128, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.attention(q, k, vt, S=S, out=out, scale=0.0, key_bias_log2=0.0 if (abl or os.environ.get("UTX_ONE_KB", "1") == "0") else 3.0)
torch.cuda.synchronize()
