"""OCP MX fp8 (e4m3 elements, E8M0 scale per 32 along K) operands for the mx8 GEMM path (include/unitex_hip.h `utx_gemm_desc.mx8`,
BASELINE configs[4] "fp8 MFMA weights").  Weights are quantised ONCE at load, activations per call, both by the HIP kernel
utx_quant_mx8 (oracle/mx8_ref.py, the checker, restates the definition): e = floor(log2(max|block|)) - 8, q = e4m3_rne(clamp(x 2^-e, +-448)).

Two scale layouts, identical values: row-major [rows, K/32] bytes (mx8 = 1: the 128 x 128-tile kernel, any shape) and TILE-PACKED dwords
[K/128][row blocks of 128][32][4] (mx8 = 2: the persistent one-wave-per-SIMD kernel, N and every column boundary a multiple of 256) --
`PackedScales` below carries the latter with its row-block count."""
import ctypes as C

import torch

from .._lib import ptr


class PackedScales:
    """tile-packed E8M0 scales of a [rows, K] matrix: `data` uint8 [K/128, row_blocks, 512]; row_blocks >= ceil(rows / 128)."""

    def __init__(self, data, rows, K):
        self.data, self.rows, self.K = data, rows, K
        assert data.dim() == 3 and data.shape[2] == 512 and data.stride(2) == 1 and data.stride(1) == 512 and data.stride(0) % 512 == 0
        self.row_blocks = data.stride(0) // 512       # row blocks per K-tile slab = the slab stride the kernels address with (a row-sliced view keeps it)

    def rowmajor(self):
        """[rows, K/32] uint8 (tests): byte (r, b) = dword [b / 4][r / 128][r % 32][(r % 128) / 32] byte b % 4"""
        nb = (self.rows + 127) // 128
        d = self.data[: self.K // 128, :nb].reshape(self.K // 128, nb, 32, 4, 4)            # kt, rb, l, im, j
        return d.permute(1, 3, 2, 0, 4).reshape(nb * 128, self.K // 32)[: self.rows].contiguous()

    def row_slice(self, r0, r1):
        """the scales of rows [r0, r1) as an operand of their own (r0 % 128 == 0: weight rows of a fused projection)"""
        assert r0 % 128 == 0
        return PackedScales(self.data[:, r0 // 128: (r1 + 127) // 128], r1 - r0, self.K)


def packed_scale_buffer(rows, K, device):
    rb = (rows + 127) // 128
    return torch.zeros(K // 128, rb, 512, dtype=torch.uint8, device=device)


def quantize_weight(W: torch.Tensor, ctx, packed=False):
    """W [N, K] bf16 on the GPU, K % 128 == 0 -> (q uint8 [N, K], s uint8 [N, K/32] | PackedScales).  The same HIP quantiser as the activations
    (one definition, one implementation; a torch-op version flushed fp32 denormals on the GPU and differed in tiny blocks)."""
    assert W.is_cuda and W.dim() == 2 and W.shape[1] % 128 == 0
    return quantize_act(W.contiguous(), ctx, packed=packed)


def quantize_act(x: torch.Tensor, ctx, out=None, packed=False):
    """x [M, K] bf16 (rows may be strided) -> (q uint8 [M, K], s) by the HIP kernel, on torch's current stream; s is uint8 [M, K/32], or a
    PackedScales with packed=True (out = (q, PackedScales) re-uses buffers: the scale buffer may be larger than needed in both K and rows)."""
    M, K = x.shape
    if out is None:
        q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
        s = PackedScales(packed_scale_buffer(M, K, x.device), M, K) if packed else torch.empty(M, K // 32, dtype=torch.uint8, device=x.device)
    else:
        q, s = out
    if isinstance(s, PackedScales):
        assert K % 128 == 0 and s.data.shape[0] >= K // 128 and s.row_blocks >= (M + 127) // 128
        ctx.check(ctx.lib.utx_quant_mx8_packed(ctx.handle, ptr(x), x.stride(0), ptr(q), q.stride(0), ptr(s.data), s.row_blocks, M, K, ctx.stream()))
    else:
        ctx.check(ctx.lib.utx_quant_mx8(ctx.handle, ptr(x), x.stride(0), ptr(q), q.stride(0), ptr(s), s.stride(0), M, K, ctx.stream()))
    return q, s
