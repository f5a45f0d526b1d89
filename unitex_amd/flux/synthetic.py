"""Synthetic FLUX.1-dev-shaped weights generated directly in HBM (no checkpoints exist in the build or
GPU containers and there is no network).  A dict-like object with the diffusers FluxTransformer2DModel
key names: tensors are produced on demand on the device, deterministically per (seed, key)."""
import math
import zlib

import torch


class SyntheticFluxStateDict:
    def __init__(self, shape, seed=0, device="cuda:0", dtype=torch.bfloat16):
        self.sh, self.seed, self.device, self.dtype = shape, seed, torch.device(device), dtype
        D, sh = shape.dim, shape
        self._shapes = {}
        lin = self._lin
        lin("x_embedder", D, sh.in_channels)
        lin("context_embedder", D, sh.joint_dim)
        for nm, kin in (("timestep_embedder", 256), ("guidance_embedder", 256), ("text_embedder", sh.pooled_dim)):
            lin("time_text_embed.%s.linear_1" % nm, D, kin)
            lin("time_text_embed.%s.linear_2" % nm, D, D)
        for i in range(sh.num_double):
            p = "transformer_blocks.%d." % i
            lin(p + "norm1.linear", 6 * D, D, 0.5)
            lin(p + "norm1_context.linear", 6 * D, D, 0.5)
            for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
                lin(p + "attn." + nm, D, D)
            for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                self._shapes[p + "attn.%s.weight" % nm] = ((128,), "norm")
            lin(p + "ff.net.0.proj", sh.mlp_ratio * D, D)
            lin(p + "ff.net.2", D, sh.mlp_ratio * D)
            lin(p + "ff_context.net.0.proj", sh.mlp_ratio * D, D)
            lin(p + "ff_context.net.2", D, sh.mlp_ratio * D)
        for i in range(sh.num_single):
            p = "single_transformer_blocks.%d." % i
            lin(p + "norm.linear", 3 * D, D, 0.5)
            lin(p + "proj_mlp", sh.mlp_ratio * D, D)
            lin(p + "proj_out", D, (1 + sh.mlp_ratio) * D)
            for nm in ("to_q", "to_k", "to_v"):
                lin(p + "attn." + nm, D, D)
            for nm in ("norm_q", "norm_k"):
                self._shapes[p + "attn.%s.weight" % nm] = ((128,), "norm")
        lin("norm_out.linear", 2 * D, D, 0.5)
        lin("proj_out", sh.in_channels, D)

    def _lin(self, name, out_f, in_f, gain=1.0):
        self._shapes[name + ".weight"] = ((out_f, in_f), gain / math.sqrt(in_f))
        self._shapes[name + ".bias"] = ((out_f,), 0.02)

    def __contains__(self, k):
        return k in self._shapes

    def keys(self):
        return self._shapes.keys()

    def __getitem__(self, k):
        shp, std = self._shapes[k]
        g = torch.Generator(device=self.device)
        g.manual_seed((self.seed * 1000003 + zlib.crc32(k.encode())) & 0x7FFFFFFF)
        t = torch.randn(shp, generator=g, device=self.device, dtype=torch.float32)
        if std == "norm":
            return (1.0 + 0.1 * t).to(self.dtype)
        return (t * std).to(self.dtype)


def synthetic_lora(state, shape, rank=64, seed=1, targets_double=None, device="cuda:0", dtype=torch.bfloat16):
    """{module: (A [r,in], B [out,r])} for the reference's LoRA targets (trainer.py:282-295)."""
    from .transformer import LORA_TARGETS_DOUBLE
    out = {}
    names = ["transformer_blocks.%d.%s" % (i, t) for i in range(shape.num_double) for t in LORA_TARGETS_DOUBLE]
    names += ["single_transformer_blocks.%d.attn.%s" % (i, t) for i in range(shape.num_single) for t in ("to_q", "to_k", "to_v")]
    for n in names:
        (o, i), _ = state._shapes[n + ".weight"]
        g = torch.Generator(device=device)
        g.manual_seed((seed * 7919 + zlib.crc32(n.encode())) & 0x7FFFFFFF)
        A = torch.randn(rank, i, generator=g, device=device) / math.sqrt(i)
        B = torch.randn(o, rank, generator=g, device=device) * (0.1 / math.sqrt(rank))
        out[n] = (A.to(dtype), B.to(dtype))
    return out


# ---- FLUX AutoencoderKL (diffusers key names / shapes [3p]) -------------------------------------------------
VAE_CHANNELS = (128, 256, 512, 512)
VAE_LATENT = 16


def vae_param_shapes(ch=VAE_CHANNELS, latent=VAE_LATENT, layers=2):
    """{diffusers parameter name: shape} of the FLUX AutoencoderKL (no quant convs)."""
    shp = {}

    def conv(name, cin, cout, k):
        shp[name + ".weight"] = (cout, cin, k, k)
        shp[name + ".bias"] = (cout,)

    def norm(name, c):
        shp[name + ".weight"] = (c,)
        shp[name + ".bias"] = (c,)

    def lin(name, cin, cout):
        shp[name + ".weight"] = (cout, cin)
        shp[name + ".bias"] = (cout,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    def mid(name, c):
        resnet(name + ".resnets.0", c, c)
        norm(name + ".attentions.0.group_norm", c)
        for t in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(name + ".attentions.0." + t, c, c)
        resnet(name + ".resnets.1", c, c)

    conv("encoder.conv_in", 3, ch[0], 3)
    cin = ch[0]
    for i, c in enumerate(ch):
        for j in range(layers):
            resnet("encoder.down_blocks.%d.resnets.%d" % (i, j), cin if j == 0 else c, c)
        if i < len(ch) - 1:
            conv("encoder.down_blocks.%d.downsamplers.0.conv" % i, c, c, 3)
        cin = c
    mid("encoder.mid_block", ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", ch[-1], 2 * latent, 3)
    rch = list(reversed(ch))
    conv("decoder.conv_in", latent, rch[0], 3)
    mid("decoder.mid_block", rch[0])
    cin = rch[0]
    for i, c in enumerate(rch):
        for j in range(layers + 1):
            resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), cin if j == 0 else c, c)
        if i < len(rch) - 1:
            conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, c, c, 3)
        cin = c
    norm("decoder.conv_norm_out", rch[-1])
    conv("decoder.conv_out", rch[-1], 3, 3)
    return shp


def synthetic_vae_state_dict(seed=0):
    """random-init VAE weights (fan-in scaled; norms near identity), CPU fp32 rounded to bf16-representable values
    so that the product (bf16) and the oracle (fp32) see identical parameters."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vae_param_shapes().items():
        if len(shape) > 1:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif ".norm" in name or "group_norm" in name or "conv_norm_out" in name:
            t = (1.0 + 0.05 * torch.randn(shape, generator=g)) if name.endswith(".weight") else 0.05 * torch.randn(shape, generator=g)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        sd[name] = t.to(torch.bfloat16).float()
    return sd
