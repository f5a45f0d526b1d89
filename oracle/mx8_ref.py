"""CPU restatement of the OCP Microscaling (MX) fp8 format used by the mx8 GEMM path (BASELINE configs[4] "fp8 MFMA weights").

TEST INFRASTRUCTURE: only tests/ import this file; the product quantises with HIP (utx_quant_mx8) / torch-on-GPU at load
(unitex_amd/flux/mx8.py) and multiplies with v_mfma_scale_f32_32x32x64_f8f6f4.  The reference itself has no fp8 path (it runs FLUX in
bf16: /root/reference/pipeline.py:96-103); fp8 is BASELINE.json's configs[4] extension, so this oracle restates the published format
-- OCP Microscaling Formats (MX) v1.0: element e4m3 (OCP "fn": no inf, max 448), block 32, shared scale E8M0 = 2^(byte - 127),
shared exponent floor(log2(amax)) - emax_elem with emax(e4m3) = 8 -- and the arithmetic of an MX dot product (exact products,
fp32 accumulation).  Parity of the HIP path against it is bit-exact for the quantiser and <= 1 bf16 ulp for the GEMM.
"""
import numpy as np
import torch


def quantize(x: torch.Tensor):
    """x [R, K] (bf16 or fp32 holding bf16 values), K % 32 == 0 -> (q uint8 [R, K] e4m3 bytes, s uint8 [R, K/32] E8M0)."""
    xf = x.to(torch.float32)
    R, K = xf.shape
    b = xf.view(R, K // 32, 32)
    amax = b.abs().amax(dim=-1)
    bits = amax.view(torch.int32)
    e = ((bits >> 23) & 0xff) - 127 - 8
    e = torch.where(amax == 0, torch.full_like(e, -127), e).clamp(-127, 127)
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64), (-e).to(torch.float64)).to(torch.float32)   # exact powers of two
    y = (b * inv[..., None]).clamp(-448.0, 448.0)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8).view(R, K)
    return q, (e + 127).to(torch.uint8)


def dequantize(q: torch.Tensor, s: torch.Tensor):
    R, K = q.shape
    v = q.view(torch.float8_e4m3fn).to(torch.float32).view(R, K // 32, 32)
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64), s.to(torch.float64) - 127.0).to(torch.float32)
    return (v * sc[..., None]).view(R, K)


def gemm(aq, a_s, bq, b_s):
    """fp32 reference of the MX dot products: dequantised operands, fp64 accumulate (products of e4m3 x 2^k values are exact)."""
    return (dequantize(aq, a_s).double() @ dequantize(bq, b_s).double().t()).float()
