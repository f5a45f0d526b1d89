"""GPU parity of the HIP AutoencoderKL (unitex_amd/flux/vae_hip.py) against the fp32 oracle (oracle/vae_ref.py).

Tolerances: inputs and parameters are bf16-representable on both sides; the oracle runs fp32 end to end, the
product rounds every activation tensor to bf16 (as the reference's bf16 VAE does, pipeline.py:106), so layer
tests use the GEMM bound (<= 2 bf16 ulp on O(1) values) and the ~30-layer end-to-end tests a relative bound."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def _vae(seed=0):
    from unitex_amd.flux.synthetic import synthetic_vae_state_dict
    from unitex_amd.flux.vae_hip import AutoencoderKL
    sd = synthetic_vae_state_dict(seed)
    return AutoencoderKL(sd, device=DEV), sd


def _nhwc(x):   # [1,C,H,W] -> [H*W, C] bf16 on the GPU
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).to(device=DEV, dtype=BF).contiguous()


def _rel(out, ref):
    return ((out - ref).abs() / ref.abs().clamp_min(1.0)).max().item()


@pytest.mark.parametrize("C,H,W,silu", [(128, 16, 24, True), (256, 9, 7, False), (512, 8, 16, True)])
def test_group_norm_silu(C, H, W, silu):
    vae, _ = _vae()
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(1, C, H, W, generator=g) * 1.5 + 0.3).to(BF).float()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).float()
    beta = (0.1 * torch.randn(C, generator=g)).to(BF).float()
    vae.w["t.weight"], vae.w["t.bias"] = gamma.to(DEV, BF), beta.to(DEV, BF)
    y = vae._norm(_nhwc(x), "t", silu).float().cpu().reshape(H, W, C).permute(2, 0, 1)[None]
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if silu:
        ref = F.silu(ref.to(BF).float())
    assert _rel(y, ref) < 1.6e-2


@pytest.mark.parametrize("cin,cout,H,W,mode", [(128, 128, 16, 24, "s1"), (256, 512, 10, 6, "s1"), (128, 128, 16, 24, "down"),
                                              (512, 512, 6, 10, "up"), (128, 3, 16, 16, "s1"), (512, 32, 8, 8, "s1")])
def test_conv3x3_implicit_gemm(cin, cout, H, W, mode):
    from unitex_amd.flux.vae_hip import AutoencoderKL
    g = torch.Generator().manual_seed(cin + cout + H)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(BF).float()
    b = (0.1 * torch.randn(cout, generator=g)).to(BF).float()
    vae = AutoencoderKL({"t.weight": w, "t.bias": b} if cout != 3 else {"t.conv_out.weight": w, "t.conv_out.bias": b}, device=DEV)
    name = "t" if cout != 3 else "t.conv_out"
    x = torch.randn(1, cin, H, W, generator=g).to(BF).float()
    if mode == "down":
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        y, Ho, Wo = vae._conv(_nhwc(x), H, W, name, stride=2)
    elif mode == "up":
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
        y, Ho, Wo = vae._conv(_nhwc(x), H, W, name, up=1)
    else:
        ref = F.conv2d(x, w, b, padding=1)
        y, Ho, Wo = vae._conv(_nhwc(x), H, W, name)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    got = y.float().cpu()[:, :cout].reshape(Ho, Wo, cout).permute(2, 0, 1)[None]
    assert _rel(got, ref) < 1.6e-2
    if mode == "s1" and cout == cin:   # residual epilogue
        r = torch.randn(1, cout, H, W, generator=g).to(BF).float()
        y2, _, _ = vae._conv(_nhwc(x), H, W, name, res=_nhwc(r))
        got2 = y2.float().cpu().reshape(H, W, cout).permute(2, 0, 1)[None]
        assert _rel(got2, (r + ref.to(BF).float())) < 1.6e-2


@pytest.mark.parametrize("cin,cout", [(3, 128), (16, 512)])
def test_conv3x3_thin(cin, cout):
    from unitex_amd.flux.vae_hip import AutoencoderKL
    g = torch.Generator().manual_seed(cin)
    H, W = 12, 20
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(BF).float()
    b = (0.1 * torch.randn(cout, generator=g)).to(BF).float()
    vae = AutoencoderKL({"t.weight": w, "t.bias": b}, device=DEV)
    x = torch.randn(1, cin, H, W, generator=g).to(BF).float()
    y, _, _ = vae._conv(_nhwc(x), H, W, "t")
    got = y.float().cpu().reshape(H, W, cout).permute(2, 0, 1)[None]
    assert _rel(got, F.conv2d(x, w, b, padding=1)) < 1.6e-2


def test_softmax_rows():
    from unitex_amd.flux import ops
    g = torch.Generator().manual_seed(3)
    s = (torch.randn(192, 320, generator=g) * 3).to(BF)
    s[5, 17] = 40.0
    d = s.to(DEV).clone()
    ctx = ops.get_ctx(0)
    from unitex_amd._lib import ptr
    ctx.check(ctx.lib.utx_softmax_rows(ctx.handle, ptr(d), 192, d.stride(0), 320, ctx.stream()))
    ref = torch.softmax(s.float(), -1)
    assert (d.float().cpu() - ref).abs().max().item() < 4e-3      # bf16 output rounding of values <= 1


def test_vae_encode_decode_match_oracle():
    from oracle import vae_ref
    vae, sd = _vae(seed=1)
    ref = vae_ref.AutoencoderKL.from_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    H, W = 64, 128                                   # latent 8 x 16 = 128 pixels (mid attention needs a multiple of 64)
    img = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(BF).float()
    with torch.no_grad():
        mom_ref = ref.encoder(img)
    dist = vae.encode(img.to(DEV))
    mean_ref, logvar_ref = mom_ref.chunk(2, dim=1)
    scale = mean_ref.abs().max().item()
    err = (dist.mean.float().cpu() - mean_ref).abs().max().item()
    assert err < 0.05 * max(scale, 1.0), "encoder mean: max err %g (scale %g)" % (err, scale)
    std_ref = torch.exp(0.5 * logvar_ref.clamp(-30, 20))
    assert ((dist.std.float().cpu() - std_ref).abs() / std_ref.clamp_min(1e-3)).max().item() < 0.1
    # same CPU-generator sampling semantics as the reference: identical noise for identical seeds
    z1 = dist.sample(torch.Generator().manual_seed(5)); z2 = dist.sample(torch.Generator().manual_seed(5))
    assert torch.equal(z1, z2) and z1.shape == (1, 16, H // 8, W // 8)
    # decoder
    z = torch.randn(1, 16, H // 8, W // 8, generator=g).to(BF).float()
    with torch.no_grad():
        dec_ref = ref.decode(z)
    dec = vae.decode(z.to(DEV)).float().cpu()
    assert dec.shape == (1, 3, H, W)
    scale = dec_ref.abs().max().item()
    err = (dec - dec_ref).abs().max().item()
    assert err < 0.05 * max(scale, 1.0), "decoder: max err %g (scale %g)" % (err, scale)
    assert (dec - dec_ref).abs().mean().item() < 0.01 * max(scale, 1.0)
