"""AutoencoderKL of FLUX.1-dev on the HIP kernels of libunitex_hip.so (MI355X).

Replaces diffusers' AutoencoderKL [3p] at the reference's call sites:
  flux_piplines/texturing/pipeline.py:226-238  _encode_vae_image: vae.encode(image).latent_dist.sample(generator),
                                               then (z - shift_factor) * scaling_factor
  flux_piplines/texturing/pipeline.py:683-692  latents / scaling_factor + shift_factor -> vae.decode

How it maps to the hardware (activations NHWC bf16 [H*W, C], one image at a time):
  * every 3x3 convolution with Cin >= 64 (all but the two stems) is an implicit GEMM: utx_gemm_bf16 in conv mode
    gathers the A-operand rows per tap straight from the NHWC activation with global_load_lds (zero page for the
    border), so there is no im2col buffer; the stride-2 downsampler's F.pad(0,1,0,1) and the decoder's nearest-2x
    upsampling are address arithmetic inside that gather; bias and the residual add ride in the GEMM epilogue;
  * the stems (3 -> 128, 16 -> 512 channels) are a thin direct convolution (utx_conv3x3_thin);
  * GroupNorm(32) + SiLU is a fused two-pass streaming kernel (utx_group_norm);
  * the single-head 512-wide mid-block attention is three GEMMs around an in-place row softmax: scores [S, S] are
    materialised in bf16 (S = H*W/64 latent pixels: 24 576 for a 512x3072 strip -> 1.2 GB, 98 304 for 1024x6144 ->
    19 GB; both are small change in 288 GB of HBM), V is produced already transposed, and its bias is added after
    the PV product (softmax rows sum to one).
Host side: descriptor plumbing only; no torch compute on the data path except layout changes at the module
boundary (NCHW image <-> NHWC) and the Gaussian sampling of the latent (CPU generator semantics of the reference).
"""
import ctypes as C
import math

import torch

from . import ops
from .._lib import ptr

BF = torch.bfloat16


class DiagonalGaussian:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.std = torch.exp(0.5 * torch.clamp(logvar.float(), -30.0, 20.0)).to(self.mean.dtype)

    def sample(self, generator=None):
        # diffusers randn_tensor: a CPU generator draws on CPU then moves to the device (one CPU generator is shared
        # by the noise / dual / control draws, reference pipeline.py:152)
        dev = self.mean.device
        gdev = generator.device if generator is not None else dev
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(dev)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def _log2(n):
    l = int(round(math.log2(n)))
    assert (1 << l) == n, "channel count must be a power of two"
    return l


class AutoencoderKL:
    scaling_factor = 0.3611
    shift_factor = 0.1159
    latent_channels = 16
    block_out_channels = (128, 256, 512, 512)

    def __init__(self, state_dict, device="cuda:0"):
        self.device = torch.device(device)
        self.ctx = ops.get_ctx(self.device.index)
        self.zero_page = torch.zeros(256, dtype=BF, device=self.device)
        self.ones = torch.ones(512, dtype=BF, device=self.device)
        self.gn_work = torch.empty(int(self.ctx.lib.utx_group_norm_workspace_bytes()), dtype=torch.uint8, device=self.device)
        self.w = {}
        for k, v in state_dict.items():
            v = v.detach()
            if v.dim() == 4 and v.shape[2] == 3:
                cout, cin = v.shape[0], v.shape[1]
                if cin < 64:    # stems: [9*Cin][Cout], tap-major, output channel contiguous
                    t = v.permute(2, 3, 1, 0).reshape(9 * cin, cout)
                else:           # implicit GEMM: [Cout (padded to 8)][(ky, kx, cin)]
                    t = v.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
                    if cout % 8:
                        t = torch.cat([t, torch.zeros(8 - cout % 8, 9 * cin, dtype=t.dtype)], 0)
            elif v.dim() == 4:  # 1x1 shortcut
                t = v.reshape(v.shape[0], v.shape[1])
            else:
                t = v
                if k.endswith("conv_out.bias") and v.shape[0] % 8:
                    t = torch.cat([v, torch.zeros(8 - v.shape[0] % 8, dtype=v.dtype)], 0)
            self.w[k] = t.to(device=self.device, dtype=BF).contiguous()

    @classmethod
    def synthetic(cls, seed=0, device="cuda:0"):
        from .synthetic import synthetic_vae_state_dict
        return cls(synthetic_vae_state_dict(seed), device=device)

    # ------------------------------------------------------------------ primitive layers
    def _gemm(self, d):
        self.ctx.check(self.ctx.lib.utx_gemm_bf16(self.ctx.handle, C.byref(d), self.ctx.stream()))

    def _conv(self, x, H, W, name, stride=1, up=0, res=None):
        """3x3 convolution of NHWC x [H*W, Cin]; returns (y [Ho*Wo, Cout], Ho, Wo).  res: residual added in the epilogue."""
        wt, b = self.w[name + ".weight"], self.w[name + ".bias"]
        cin = x.shape[1]
        if cin < 64:
            cout = wt.shape[1]
            y = torch.empty(H * W, cout, dtype=BF, device=self.device)
            self.ctx.check(self.ctx.lib.utx_conv3x3_thin(self.ctx.handle, ptr(x), H, W, cin, ptr(wt), ptr(b), cout, ptr(y), self.ctx.stream()))
            return y, H, W
        if stride == 2:
            Ho, Wo, pad = H // 2, W // 2, 0       # F.pad(x, (0,1,0,1)) + stride-2 valid conv
        else:
            Ho, Wo, pad = H << up, W << up, 1
        n = wt.shape[0]
        y = torch.empty(Ho * Wo, n, dtype=BF, device=self.device)
        d = ops.make_gemm_desc(x, wt, y, bias=b, gate=self.ones[:n] if res is not None else None, res=res)
        d.M, d.N, d.K = Ho * Wo, n, 9 * cin
        d.conv_Hi, d.conv_Wi, d.conv_Wo, d.conv_cin_log2 = H, W, Wo, _log2(cin)
        d.conv_stride, d.conv_pad, d.conv_up, d.zero_page = stride, pad, up, ptr(self.zero_page)
        self._gemm(d)
        return y, Ho, Wo

    def _norm(self, x, name, silu):
        y = torch.empty_like(x)
        self.ctx.check(self.ctx.lib.utx_group_norm(self.ctx.handle, ptr(x), x.shape[0], x.shape[1], ptr(self.w[name + ".weight"]),
                                                   ptr(self.w[name + ".bias"]), 1e-6, int(silu), ptr(y), ptr(self.gn_work), self.ctx.stream()))
        return y

    def _resnet(self, x, H, W, name):
        h = self._norm(x, name + ".norm1", True)
        h, _, _ = self._conv(h, H, W, name + ".conv1")
        h = self._norm(h, name + ".norm2", True)
        if (name + ".conv_shortcut.weight") in self.w:
            x = ops.gemm(x, self.w[name + ".conv_shortcut.weight"], bias=self.w[name + ".conv_shortcut.bias"])
        y, _, _ = self._conv(h, H, W, name + ".conv2", res=x)
        return y

    def _attn(self, x, name):
        S, Cc = x.shape
        if S % 64:
            raise ValueError("VAE mid attention needs H*W/64 latent pixels to be a multiple of 64 (got %d)" % S)
        w = self.w
        h = self._norm(x, name + ".group_norm", False)
        q = ops.gemm(h, w[name + ".to_q.weight"], bias=w[name + ".to_q.bias"])
        k = ops.gemm(h, w[name + ".to_k.weight"], bias=w[name + ".to_k.bias"])
        vt = ops.gemm(w[name + ".to_v.weight"], h)                                   # V^T [C, S]; bias added after PV
        # scores are materialised one block of query rows at a time ([QB, S] bf16, <= ~2 GiB): softmax and P V are row-wise, so the
        # result is bit-identical to the whole [S, S] matrix (18.9 GB at 1024 x 6144, 550 GB at 2048^2 x 8 views)
        QB = max(256, min(S, ((1 << 30) // S) // 256 * 256))
        a = torch.empty(S, Cc, dtype=BF, device=self.device)
        sbuf = torch.empty(min(QB, S), S, dtype=BF, device=self.device)
        for r0 in range(0, S, QB):
            r1 = min(S, r0 + QB)
            sblk = sbuf[: r1 - r0]
            ops.gemm(q[r0:r1], k, alpha=1.0 / math.sqrt(Cc), out=sblk)              # scores [QB, S]
            self.ctx.check(self.ctx.lib.utx_softmax_rows(self.ctx.handle, ptr(sblk), r1 - r0, sblk.stride(0), S, self.ctx.stream()))
            ops.gemm(sblk, vt, bias=w[name + ".to_v.bias"], out=a[r0:r1])            # P V + b_v
        del sbuf
        return ops.gemm(a, w[name + ".to_out.0.weight"], bias=w[name + ".to_out.0.bias"], gate=self.ones[:Cc], res=x)

    def _mid(self, x, H, W, name):
        x = self._resnet(x, H, W, name + ".resnets.0")
        x = self._attn(x, name + ".attentions.0")
        return self._resnet(x, H, W, name + ".resnets.1")

    # ------------------------------------------------------------------ module surface (NCHW in / out, batch 1)
    @torch.no_grad()
    def encode(self, image):
        assert image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 3
        _, _, H, W = image.shape
        assert H % 8 == 0 and W % 8 == 0
        x = image[0].to(device=self.device, dtype=BF).permute(1, 2, 0).reshape(H * W, 3).contiguous()
        x, _, _ = self._conv(x, H, W, "encoder.conv_in")
        n = len(self.block_out_channels)
        for i in range(n):
            for j in range(2):
                x = self._resnet(x, H, W, "encoder.down_blocks.%d.resnets.%d" % (i, j))
            if i < n - 1:
                x, H, W = self._conv(x, H, W, "encoder.down_blocks.%d.downsamplers.0.conv" % i, stride=2)
        x = self._mid(x, H, W, "encoder.mid_block")
        x = self._norm(x, "encoder.conv_norm_out", True)
        m, _, _ = self._conv(x, H, W, "encoder.conv_out")
        return DiagonalGaussian(m.reshape(H, W, -1).permute(2, 0, 1)[None].contiguous())

    @torch.no_grad()
    def decode(self, z):
        assert z.dim() == 4 and z.shape[0] == 1 and z.shape[1] == self.latent_channels
        _, Cz, H, W = z.shape
        x = z[0].to(device=self.device, dtype=BF).permute(1, 2, 0).reshape(H * W, Cz).contiguous()
        x, _, _ = self._conv(x, H, W, "decoder.conv_in")
        x = self._mid(x, H, W, "decoder.mid_block")
        n = len(self.block_out_channels)
        for i in range(n):
            for j in range(3):
                x = self._resnet(x, H, W, "decoder.up_blocks.%d.resnets.%d" % (i, j))
            if i < n - 1:
                x, H, W = self._conv(x, H, W, "decoder.up_blocks.%d.upsamplers.0.conv" % i, up=1)
        x = self._norm(x, "decoder.conv_norm_out", True)
        y, _, _ = self._conv(x, H, W, "decoder.conv_out")
        return y[:, :3].reshape(H, W, 3).permute(2, 0, 1)[None].contiguous()
