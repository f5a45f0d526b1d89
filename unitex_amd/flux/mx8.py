"""OCP MX fp8 (e4m3 elements, E8M0 scale per 32 along K) operands for the mx8 GEMM path (include/unitex_hip.h `utx_gemm_desc.mx8`,
BASELINE configs[4] "fp8 MFMA weights").  Weights are quantised ONCE at load, activations per call, both by the HIP kernel
utx_quant_mx8 (oracle/mx8_ref.py, the checker, restates the definition): e = floor(log2(max|block|)) - 8, q = e4m3_rne(clamp(x 2^-e, +-448))."""
import ctypes as C

import torch

from .._lib import ptr


def quantize_weight(W: torch.Tensor, ctx):
    """W [N, K] bf16 on the GPU, K % 128 == 0 -> (q uint8 [N, K], s uint8 [N, K/32]).  The same HIP quantiser as the activations
    (one definition, one implementation; a torch-op version flushed fp32 denormals on the GPU and differed in tiny blocks)."""
    assert W.is_cuda and W.dim() == 2 and W.shape[1] % 128 == 0
    return quantize_act(W.contiguous(), ctx)


def quantize_act(x: torch.Tensor, ctx, out=None):
    """x [M, K] bf16 (rows may be strided) -> (q uint8 [M, K], s uint8 [M, K/32]) by the HIP kernel, on torch's current stream."""
    M, K = x.shape
    if out is None:
        out = (torch.empty(M, K, dtype=torch.uint8, device=x.device), torch.empty(M, K // 32, dtype=torch.uint8, device=x.device))
    q, s = out
    ctx.check(ctx.lib.utx_quant_mx8(ctx.handle, ptr(x), x.stride(0), ptr(q), q.stride(0), ptr(s), s.stride(0), M, K, ctx.stream()))
    return q, s
