"""ORACLE (test infrastructure, not product): fp32 PyTorch restatement of the FLUX.1-dev AutoencoderKL.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.  The product VAE is
unitex_amd/flux/vae_hip.py (hand-written HIP kernels behind the C ABI).

Architecture restated from the public diffusers AutoencoderKL [3p] (absent from /root/reference and from this
image -> parity unpinned against diffusers itself): block_out_channels (128, 256, 512, 512), 2 layers / block
(3 in the decoder), 16 latent channels, GroupNorm(32, eps 1e-6), SiLU, single-head mid attention, stride-2
downsampler with F.pad (0,1,0,1), nearest-2x upsampler, scaling 0.3611, shift 0.1159.  Call sites in the reference:
flux_piplines/texturing/pipeline.py:226-238 (encode + sample + shift/scale) and :683-692 (unscale + decode).
Parameter names follow diffusers so the same state dict drives the oracle and the product."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)
        a = self.to_out[0](a).transpose(1, 2).reshape(B, C, H, W)
        return x + a


class Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class MidBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])
        self.attentions = nn.ModuleList([AttnBlock(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=16, layers=2):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cin = ch[0]
        for i, c in enumerate(ch):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else c, c) for j in range(layers)])
            blk.downsamplers = nn.ModuleList([Downsample(c)]) if i < len(ch) - 1 else None
            self.down_blocks.append(blk)
            cin = c
        self.mid_block = MidBlock(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                x = r(x)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=16, layers=2):
        super().__init__()
        rch = list(reversed(ch))
        self.conv_in = nn.Conv2d(latent, rch[0], 3, padding=1)
        self.mid_block = MidBlock(rch[0])
        self.up_blocks = nn.ModuleList()
        cin = rch[0]
        for i, c in enumerate(rch):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else c, c) for j in range(layers + 1)])
            blk.upsamplers = nn.ModuleList([Upsample(c)]) if i < len(rch) - 1 else None
            self.up_blocks.append(blk)
            cin = c
        self.conv_norm_out = nn.GroupNorm(32, rch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rch[-1], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for blk in self.up_blocks:
            for r in blk.resnets:
                x = r(x)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussian:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))

    def sample(self, generator=None):
        # diffusers randn_tensor: a CPU generator draws on CPU then moves to the device (A19: one CPU
        # generator shared by noise / dual / control draws, pipeline.py:152)
        dev = self.mean.device
        gdev = generator.device if generator is not None else dev
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(dev)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    scaling_factor = 0.3611
    shift_factor = 0.1159
    latent_channels = 16
    block_out_channels = (128, 256, 512, 512)

    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder()

    @torch.no_grad()
    def encode(self, x):
        return DiagonalGaussian(self.encoder(x))

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(z)

    @classmethod
    def from_state_dict(cls, sd):
        m = cls()
        m.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
        return m.eval()

    @classmethod
    def synthetic(cls, seed=0, device="cpu", dtype=torch.float32):
        g = torch.Generator().manual_seed(seed)
        m = cls()
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() > 1:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) / (fan_in ** 0.5))
                else:
                    p.copy_(1.0 + 0.02 * torch.randn(p.shape, generator=g) if "norm" in "" else 0.02 * torch.randn(p.shape, generator=g))
            for mod in m.modules():
                if isinstance(mod, nn.GroupNorm):
                    mod.weight.fill_(1.0)
                    mod.bias.zero_()
        return m.to(device=device, dtype=dtype).eval()
