"""The floating-point end of the contract, measured end to end.

Reference loop: /root/reference/flux_piplines/texturing/pipeline.py:633-692 -- K denoise steps (transformer -> flow-match Euler
step, condition tail re-pinned before every call), cut off the condition tail, unpack, VAE decode, postprocess to uint8.

Product: PBRFluxPipeline.denoise (HIP FluxDiT + fused utx_sched_step) -> HIP AutoencoderKL.decode -> uint8.
Oracle : oracle/dit_ref.denoise_loop(emulate_bf16=True) (fp32 arithmetic, rounded to bf16 at every tensor boundary of the bf16
         reference) -> oracle/vae_ref (fp32) -> the same postprocess.

STATED TOLERANCE (asserted below, measured values printed and quoted in DESIGN.md section 3):
  * latents after K = 4 steps: max |d| <= 0.02 max|latent|, mean |d| <= 0.0015 max|latent|
  * decoded uint8 image: >= 99 % of the pixels within 2 LSB, >= 99.9 % within 4 LSB, no pixel further than 8 LSB
  (measured on MI355X, profiles/r03_e2e_tol_a.log: max 0.005-0.008, mean 0.0004-0.0005 of max|latent|; 54 % of the pixels equal, 94 % within 1 LSB,
  99.7 % within 2, none beyond 4 -- the same for the tiny 2 + 4-block network and the full-width 1 + 1-block one at 9728 tokens)
BASELINE's "1e-3 max-abs" is below half a bf16 ulp at 1.0 (3.9e-3) and cannot be met by ANY bf16 evaluation order that differs from
the reference's own (the reference itself is not reproducible to 1e-3 across GPU kernels libraries); what can be stated is the
drift of this implementation's bf16 kernels against an fp32-accumulating restatement with the reference's rounding points.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dit_ref, pipeline_ref, vae_ref

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def _postprocess_u8(img, em=None):
    """product (em None): the pipeline's own postprocess on the HIP VAE's bf16 output; oracle: oracle/pipeline_ref.postprocess_u8 -- the reference denormalises
    the VAE's bf16 output tensor in bf16 (VaeImageProcessor [3p], pinned through fixture G1), the plain-fp32 leg in fp32"""
    if em is None:
        from unitex_amd.flux.pipeline import PBRFluxPipeline
        return np.asarray(PBRFluxPipeline._postprocess(img)[0])[None]
    return pipeline_ref.postprocess_u8(img, em)[None]


def _run_both(cfg, shape, S_txt, zero_text, lat_hw, dual_hw, K, lora_rank, n_threads=None, fp8_too=False, fp32_too=False):
    """-> {"bf16": (latents, uint8), ["fp8": ...]} of the product, {"emulated": (latents, uint8), ["fp32": ...]} of the oracle"""
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import synthetic_vae_state_dict
    from unitex_amd.flux.transformer import FluxDiT
    from unitex_amd.flux.vae_hip import AutoencoderKL
    hl, wl = lat_hw                              # packed token grid of the noise strip (latent = 2 hl x 2 wl, image = 16 hl x 16 wl)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=lora_rank, seed=2)
    loras = [(la, 1.0)]
    g = torch.Generator().manual_seed(63)
    n_noise = hl * wl
    noise = torch.randn(n_noise, 64, generator=g).to(BF)
    ids = [dit_ref.latent_image_ids(hl, wl), dit_ref.latent_image_ids(hl, wl, offset_y=hl)]
    n_cond = hl * wl
    if dual_hw:
        ids.append(dit_ref.latent_image_ids(dual_hw[0], dual_hw[1], offset_x=wl, offset_y=hl))
        n_cond += dual_hw[0] * dual_hw[1]
    cond = torch.randn(n_cond, 64, generator=g).to(BF)
    noise_ids, cond_ids = ids[0], torch.cat(ids[1:], 0)
    if zero_text:       # what the reference feeds (pipeline.py:538-543): the text-token de-duplication path of FluxDiT
        enc = torch.zeros(S_txt, cfg.joint_dim).to(BF); pooled = torch.zeros(1, cfg.pooled_dim).to(BF)
    else:
        enc = (0.5 * torch.randn(S_txt, cfg.joint_dim, generator=g)).to(BF); pooled = (0.5 * torch.randn(1, cfg.pooled_dim, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    vsd = synthetic_vae_state_dict(1)
    H, W = 16 * hl, 16 * wl
    # ---- product
    vae = AutoencoderKL(vsd, device=DEV)
    got = {}
    for mode in (("bf16", "fp8") if fp8_too else ("bf16",)):
        tr = FluxDiT(sd, shape, device=DEV, fp8_weights=(mode == "fp8"))
        pipe = PBRFluxPipeline(tr, vae, device=DEV)
        pipe.load_lora_weights(la, "texture")
        pipe.set_adapters(["texture"], [1.0])
        lat = pipe.denoise(noise[None], noise_ids, cond[None], cond_ids, enc.to(DEV)[None], pooled.to(DEV), txt_ids, K, 3.5)
        z = pipe._unpack_latents(lat, H, W, 8)
        z = (z / vae.scaling_factor) + vae.shift_factor
        img = vae.decode(z.to(BF))
        torch.cuda.synchronize()
        got[mode] = (lat[0].float().cpu(), _postprocess_u8(img))
        del tr, pipe
        torch.cuda.empty_cache()
    # ---- oracle
    if n_threads:
        torch.set_num_threads(n_threads)
    vref = vae_ref.AutoencoderKL.from_state_dict(vsd)
    ref = {}
    for name, em in ((("emulated", True), ("fp32", False)) if fp32_too else (("emulated", True),)):
        ref_lat = dit_ref.denoise_loop(sd, cfg, noise.float(), cond.float(), enc.float(), pooled.float(), txt_ids,
                                       torch.cat([noise_ids, cond_ids], 0), K, guidance=3.5, loras=loras, emulate_bf16=em)
        rz = dit_ref.unpack_latents(ref_lat[None], H, W, 8)
        rz = dit_ref._rb(dit_ref._rb(rz / vae_ref.AutoencoderKL.scaling_factor, em) + vae_ref.AutoencoderKL.shift_factor, em)
        ref[name] = (ref_lat, _postprocess_u8(vref.decode(rz), em))
    return got, ref


def _report(tag, got_lat, ref_lat, got_u8, ref_u8):
    mx = ref_lat.abs().max().item()
    d = (got_lat - ref_lat).abs()
    du = np.abs(got_u8.astype(np.int32) - ref_u8.astype(np.int32))
    hist = {k: float((du <= k).mean()) for k in (0, 1, 2, 4, 8)}
    # a decoded image that is constant (saturated) would make the uint8 comparison vacuous
    spread = float(ref_u8.std())
    print("\n[e2e tolerance] %s: latents max|d| %.4g = %.4g of max|lat| %.3g, mean|d| %.4g = %.4g of max; uint8: ==0 %.4f, <=1 %.4f, <=2 %.4f, "
          "<=4 %.4f, <=8 %.4f, max %d LSB (image std %.1f LSB)" % (tag, d.max().item(), d.max().item() / mx, mx, d.mean().item(), d.mean().item() / mx,
                                                                   hist[0], hist[1], hist[2], hist[4], hist[8], int(du.max()), spread))
    return mx, d, du, hist, spread


@pytest.mark.parametrize("zero_text", [True, False])
def test_k_step_denoise_and_vae_decode_tiny_full_depth_pattern(zero_text):
    """2 double + 4 single blocks (the depth pattern: double blocks feeding single blocks, both more than once), 2 heads, rank-16 LoRA, 4 real
    steps with re-pin, 128 x 384 strip + control + 64 x 64 dual; zero text embeddings (128 tokens: the de-duplicated path) and random ones."""
    from unitex_amd.flux.transformer import FluxShape
    cfg = dit_ref.tiny_config(heads=2, double=2, single=4, joint_dim=64, pooled_dim=64)
    shape = FluxShape(num_heads=2, num_double=2, num_single=4, joint_dim=64, pooled_dim=64)
    got, ref = _run_both(cfg, shape, 128 if zero_text else 64, zero_text, (8, 24), (4, 4), 4, 16)
    (got_lat, got_u8), (ref_lat, ref_u8) = got["bf16"], ref["emulated"]
    mx, d, du, hist, spread = _report("tiny 2+4 blocks, %s text" % ("zero" if zero_text else "random"), got_lat, ref_lat, got_u8, ref_u8)
    assert torch.isfinite(got_lat).all() and spread > 4.0
    assert d.max().item() <= 0.02 * mx and d.mean().item() <= 0.0015 * mx
    assert hist[2] >= 0.99 and hist[4] >= 0.999 and du.max() <= 8


def test_k_step_denoise_and_vae_decode_full_width_at_config1_shape():
    """BASELINE configs[0] shape at the real FLUX width: D = 3072, 24 heads, rank-64 LoRA, 512 zero text tokens, 512 x 2048 strip (4096 noise
    tokens) + control + 512^2 dual = 9728 joint tokens; depth cut to 1 + 1 blocks so the fp32 oracle does its steps in minutes.  Round 4: K = 8 steps
    (round 3: 4), the MX fp8 product (speedup_mode="fp8") against the SAME bf16-emulating oracle run, and the distance of everything to the plain fp32
    evaluation (emulate_bf16=False) printed beside it.
    STATED TOLERANCE, bf16: as the tiny cases.  fp8 (a different numerics contract, never the default): latents max |d| <= 0.03 max|latent|, mean <= 0.003;
    decoded image >= 97 % of the pixels within 2 LSB, >= 99.9 % within 4, none beyond 12.  Measured (profiles/r04_e2e_tol_a.log): bf16 max 0.0097 / mean
    0.00037 of max|latent|, 54 % equal, 94.6 % within 1 LSB, 99.8 % within 2, max 5; MX fp8 0.0097 / 0.00096, 49.7 % / 91.2 % / 99.4 %, max 6."""
    from unitex_amd.flux.transformer import FluxShape
    cfg = dit_ref.FluxConfig(num_double=1, num_single=1)
    shape = FluxShape(num_double=1, num_single=1)
    nt = max(1, min(len(os.sched_getaffinity(0)), 64))
    # the plain-fp32 leg doubles the oracle's CPU time (8 full-width steps are ~3 minutes on 64 host threads): opt-in, run once per round and quoted in DESIGN
    fp32_too = os.environ.get("UTX_TEST_FP32_ORACLE", "0") == "1"
    got, ref = _run_both(cfg, shape, 512, True, (32, 128), (32, 32), 8, 64, n_threads=nt, fp8_too=True, fp32_too=fp32_too)
    ref_lat, ref_u8 = ref["emulated"]
    mx, d, du, hist, spread = _report("full width 1+1 blocks, S = 9728, K = 8, bf16 vs emulating oracle", got["bf16"][0], ref_lat, got["bf16"][1], ref_u8)
    assert torch.isfinite(got["bf16"][0]).all() and spread > 4.0
    assert d.max().item() <= 0.02 * mx and d.mean().item() <= 0.0015 * mx
    assert hist[2] >= 0.99 and hist[4] >= 0.999 and du.max() <= 8
    mx8, d8, du8, hist8, _ = _report("full width 1+1 blocks, S = 9728, K = 8, MX fp8 vs emulating oracle", got["fp8"][0], ref_lat, got["fp8"][1], ref_u8)
    assert torch.isfinite(got["fp8"][0]).all()
    assert d8.max().item() <= 0.03 * mx8 and d8.mean().item() <= 0.003 * mx8
    assert hist8[2] >= 0.97 and hist8[4] >= 0.999 and du8.max() <= 12
    # the distance to the plain fp32 evaluation (no bf16 rounding anywhere in the oracle): reported, and bounded loosely -- it is the reference's own bf16 error too
    if fp32_too:
        _report("    the emulating oracle itself vs plain fp32", ref_lat, ref["fp32"][0], ref_u8, ref["fp32"][1])
        _report("    bf16 product vs plain fp32", got["bf16"][0], ref["fp32"][0], got["bf16"][1], ref["fp32"][1])
        _report("    MX fp8 product vs plain fp32", got["fp8"][0], ref["fp32"][0], got["fp8"][1], ref["fp32"][1])


def _smooth_image(h, w, seed):
    """a synthetic RGB uint8 picture with structure at several scales (a flat or white-noise image would make the VAE round trip trivial)"""
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(1, 3, h, w)
    for s_ in (4, 16, 64):
        img = img + torch.nn.functional.interpolate(torch.randn(1, 3, max(1, h // s_), max(1, w // s_), generator=g), size=(h, w), mode="bicubic", align_corners=False)
    img = (img - img.min()) / (img.max() - img.min())
    return (img[0].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)


@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp8-attn"])
def test_full_schedule_28_steps_two_passes_with_uint8_handoff(mode):
    """THE REAL SCHEDULE: /root/reference/pipeline.py:246-289 -- texture pass, 28 steps (control image + reference image, texture adapter
    on) -> uint8 image -> delight pass, 28 more steps with THAT image as the control image (delight adapter on) -> uint8 image; each pass is one
    PBRFluxPipeline.__call__ (flux_piplines/texturing/pipeline.py:404-700: VAE encode + posterior sample from the shared generator, pack, denoise with
    re-pin, unpack, VAE decode, postprocess).  Product: unitex_amd PBRFluxPipeline on the HIP FluxDiT / VAE, bf16 and MX fp8 (speedup_mode="fp8").
    Oracle: oracle/pipeline_ref.texturing_call twice (bf16-emulating) on the same seed; the plain fp32 evaluation of the same schedule beside it.
    Tiny 2 + 4-block network (2 heads) so that 2 x 56 CPU evaluations + 12 VAE passes stay in the minute range: 128 x 384 strip, 64 x 64 reference image.
    STATED TOLERANCE on the final (delighted) image and on the intermediate one -- asserted below, measured values printed and quoted in DESIGN section 3."""
    from PIL import Image
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import synthetic_vae_state_dict
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.flux.vae_hip import AutoencoderKL
    cfg = dit_ref.tiny_config(heads=2, double=2, single=4, joint_dim=64, pooled_dim=64)
    shape = FluxShape(num_heads=2, num_double=2, num_single=4, joint_dim=64, pooled_dim=64)
    H, W, steps, S_txt = 128, 384, 28, 128
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
    la = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2)
    lb = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=3)
    vsd = synthetic_vae_state_dict(1)
    control = _smooth_image(H, W, 5)
    reference = _smooth_image(64, 64, 6)
    # ---- product: the reference's call sequence
    tr = FluxDiT(sd, shape, device=DEV, fp8_weights=(mode != "bf16"), fp8_attention=(mode == "fp8-attn"))
    if mode == "fp8-attn":
        assert tr.fp8_attention
    pipe = PBRFluxPipeline(tr, AutoencoderKL(vsd, device=DEV), device=DEV)
    pipe.load_lora_weights(la, "texture")
    pipe.load_lora_weights(lb, "delight")
    g = torch.Generator().manual_seed(2024)
    pipe.set_adapters(["texture", "delight"], [1.0, 0.0])
    tex = pipe(prompt="[MVFLUX]", control_image=Image.fromarray(control), dual_image=Image.fromarray(reference), height=H, width=W, n_rows=1, n_cols=6,
               num_inference_steps=steps, guidance_scale=3.5, max_sequence_length=S_txt, generator=g).images[0]
    pipe.set_adapters(["texture", "delight"], [0.0, 1.0])
    out = pipe(prompt="[MVFLUX]", control_image=tex, height=H, width=W, n_rows=1, n_cols=6, num_inference_steps=steps, guidance_scale=3.5,
               max_sequence_length=S_txt, generator=g).images[0]
    torch.cuda.synchronize()
    got_tex, got_out = np.asarray(tex), np.asarray(out)
    # ---- oracle: bf16-emulating and plain fp32
    vref = vae_ref.AutoencoderKL.from_state_dict(vsd)
    ref = {}
    for name, em in (("emulated", True), ("fp32", False)):
        g2 = torch.Generator().manual_seed(2024)

        def call(ctrl, dual, loras):
            return pipeline_ref.texturing_call(sd, cfg, vref, ctrl, dual, H, W, g2, steps, loras, max_sequence_length=S_txt, emulate_bf16=em)
        t_ = call(control, reference, [(la, 1.0), (lb, 0.0)])
        ref[name] = (t_, call(t_, None, [(la, 0.0), (lb, 1.0)]))

    def hist_of(a, b):
        du = np.abs(a.astype(np.int32) - b.astype(np.int32))
        return {k: float((du <= k).mean()) for k in (0, 1, 2, 4, 8, 16)}, int(du.max()), float(du.mean())
    rows = []
    for what, a, b in (("texture pass (28 steps)", got_tex, ref["emulated"][0]), ("delight pass (28 + 28 steps, uint8 hand-off)", got_out, ref["emulated"][1]),
                       ("  same, product vs plain fp32", got_out, ref["fp32"][1]), ("  emulating oracle vs plain fp32", ref["emulated"][1], ref["fp32"][1])):
        h, mxd, mean = hist_of(a, b)
        rows.append((what, h, mxd, mean))
        print("\n[full schedule, %s] %s: uint8 ==0 %.4f, <=1 %.4f, <=2 %.4f, <=4 %.4f, <=8 %.4f, <=16 %.4f, max %d LSB, mean %.3f LSB (image std %.1f)" % (
            mode, what, h[0], h[1], h[2], h[4], h[8], h[16], mxd, mean, float(b.std())))
    assert float(ref["emulated"][1].std()) > 4.0 and float(ref["emulated"][0].std()) > 4.0, "a constant image would make the comparison vacuous"
    (_, h_tex, mx_tex, _), (_, h_out, mx_out, _) = rows[0], rows[1]
    # measured on MI355X (profiles/r04_e2e_tol_a.log): bf16 -- texture pass 52 % equal / 92.5 % within 1 LSB / 99.5 % within 2 / max 4, delight pass (56 steps,
    # uint8 hand-off) 52 % / 92.8 % / 99.6 % / max 4; MX fp8 -- 46 % / 86 % / 98 % / max 6 and 5.  The bf16-emulating oracle itself is 17 % / 26 % / 34 % / max 45
    # from the plain fp32 evaluation of the same schedule (mean 5.9 LSB): that is the bf16 network's own distance to fp32, and the product's is the same.
    if mode == "bf16":
        assert h_tex[2] >= 0.99 and h_tex[4] >= 0.999 and mx_tex <= 8
        assert h_out[2] >= 0.99 and h_out[4] >= 0.999 and mx_out <= 8
    elif mode == "fp8":
        assert h_tex[2] >= 0.95 and h_tex[4] >= 0.995 and mx_tex <= 12
        assert h_out[2] >= 0.95 and h_out[4] >= 0.995 and mx_out <= 12
    else:
        # MX fp8 linears AND MX fp8 attention (opt-in): e4m3 Q / K enter through the exponential -- a per-layer perturbation of the attention branch of a few
        # per cent (tests/test_attention_fp8_gpu.py); what it does to the image after the full schedule is STATED here, asserted loosely
        # measured (profiles/r04_attn_fp8_e2e.log): 48 % equal / 86.5 % within 1 LSB / 98.3 % within 2 / max 5 -- on this tiny 2-head network indistinguishable from
        # fp8 linears alone; a peaked full-size softmax is perturbed more per layer (10 % relative Frobenius on the attention output in test_attention_fp8_gpu.py)
        assert h_tex[2] >= 0.95 and h_tex[4] >= 0.995 and mx_tex <= 12
        assert h_out[2] >= 0.95 and h_out[4] >= 0.995 and mx_out <= 12
    # the product is no further from plain fp32 than the reference's own bf16 arithmetic is (the emulating oracle): mean LSB distance within 10 % of each other
    # (fp8 attention: within 35 %)
    assert rows[2][3] <= (1.35 if mode == "fp8-attn" else 1.10) * rows[3][3] + 0.1


class _HostView:
    """the device-resident synthetic state dict seen from the oracle: a weight is generated on the GPU (deterministic per key) and copied to the host
    when the oracle touches it -- the 11.9 G parameters never sit in host memory at once (24 GB in bf16, 48 GB in fp32)."""

    def __init__(self, dev_sd):
        self.d = dev_sd

    def __contains__(self, k):
        return k in self.d

    def __getitem__(self, k):
        return self.d[k].cpu()

    def get(self, k, default=None):
        return self.d[k].cpu() if k in self.d else default


def test_full_depth_full_width_forward_matches_oracle():
    """ALL 19 double + 38 single blocks at the real FLUX width (D = 3072, 24 heads, 11.9 G synthetic parameters, rank-16 LoRA) on a short sequence
    (64 text + 512 + 512 + 64 image tokens = 1152), one transformer evaluation against the bf16-emulating fp32 oracle: the drift of the HIP kernels'
    bf16 evaluation order through the full depth (SURVEY 7 "hard parts": 57 layers), which the 1 + 1-block full-width tests cannot show.
    STATED TOLERANCE: max |d| <= 0.03 max|out|, mean |d| <= 0.005 max|out| (measured on MI355X, profiles/r03_full_depth_a.log: 0.012 and 0.0023)."""
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    cfg = dit_ref.FluxConfig()
    shape = FluxShape()
    sd = SyntheticFluxStateDict(shape, seed=0, device=DEV)
    la_dev = synthetic_lora(sd, shape, rank=16, seed=2, device=DEV)
    la = {k: (a.cpu(), b.cpu()) for k, (a, b) in la_dev.items()}
    g = torch.Generator().manual_seed(63)
    S_txt = 64
    img_ids = torch.cat([dit_ref.latent_image_ids(16, 32), dit_ref.latent_image_ids(16, 32, offset_y=16),
                         dit_ref.latent_image_ids(8, 8, offset_x=32, offset_y=16)], 0)
    S_img = img_ids.shape[0]
    lat = torch.randn(S_img, 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, cfg.joint_dim, generator=g)).to(BF); pooled = (0.5 * torch.randn(1, cfg.pooled_dim, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    m = FluxDiT(sd, shape, device=DEV)
    m.set_lora([(la_dev, 1.0)])
    m.set_positions(txt_ids, img_ids)
    m.set_conditioning(enc.to(DEV), pooled.to(DEV), 3.5)
    out = m.forward(lat.to(DEV), 0.4375).float().cpu()
    torch.cuda.synchronize()
    del m
    torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ref = dit_ref.flux_forward(_HostView(sd), cfg, lat.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids, loras=[(la, 1.0)],
                               emulate_bf16=True)
    mx = ref.abs().max().item()
    d = (out - ref).abs()
    print("\n[full depth 19 + 38 blocks, full width, S = %d] max|d| %.4g = %.4g of max|out| %.3g, mean|d| %.4g = %.4g of max" % (
        S_txt + S_img, d.max().item(), d.max().item() / mx, mx, d.mean().item(), d.mean().item() / mx))
    assert torch.isfinite(out).all()
    assert d.max().item() <= 0.03 * mx and d.mean().item() <= 0.005 * mx


def test_full_depth_full_width_fp8_numerics_stated():
    """BASELINE configs[4] ("fp8 MFMA weights") at FULL DEPTH (round 5): the same 19 + 38 blocks at D = 3072, 24 heads, rank-16 LoRA, 1152 tokens as the bf16 test above,
    in the product's three numerics modes -- bf16, MX fp8 linears (fp8_weights), MX fp8 linears + MX fp8 attention (fp8_attention) --
      (a) ONE transformer evaluation of each against the bf16-emulating fp32 oracle AND against the bf16 product: max / mean |d| in units of max|out|;
      (b) K = 4 real denoise steps (re-pin, fused Euler step) of each product -> HIP VAE decode -> uint8: histogram of LSB differences against the bf16 product.
    STATED (measured on MI355X, profiles/r05_full_depth_fp8.log; asserted with a factor ~2 of margin):
      one evaluation, |d| / max|out| vs the emulating oracle:  bf16 max 0.0122 mean 0.00226 | fp8 linears max 0.0509 mean 0.00896 | + fp8 attention max 0.0517 mean 0.00895
      (the fp8 modes are 4.0 x the bf16 product's distance to the oracle; fp8 attention adds nothing measurable to fp8 linears at this sequence length)
      K = 4 steps + VAE decode, uint8 vs the bf16 product:  fp8 30.6 % equal / 58.4 % <= 1 / 79.4 % <= 2 / 96.6 % <= 4 / 99.98 % <= 8 LSB, max 11;  + fp8 attention: the same, max 12."""
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora, synthetic_vae_state_dict
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.flux.vae_hip import AutoencoderKL
    cfg = dit_ref.FluxConfig()
    shape = FluxShape()
    sd = SyntheticFluxStateDict(shape, seed=0, device=DEV)
    la_dev = synthetic_lora(sd, shape, rank=16, seed=2, device=DEV)
    la = {k: (a.cpu(), b.cpu()) for k, (a, b) in la_dev.items()}
    g = torch.Generator().manual_seed(63)
    S_txt, hl, wl = 64, 16, 32
    noise_ids = dit_ref.latent_image_ids(hl, wl)
    cond_ids = torch.cat([dit_ref.latent_image_ids(hl, wl, offset_y=hl), dit_ref.latent_image_ids(8, 8, offset_x=wl, offset_y=hl)], 0)
    img_ids = torch.cat([noise_ids, cond_ids], 0)
    n_noise = hl * wl
    lat0 = torch.randn(img_ids.shape[0], 64, generator=g).to(BF)
    enc = (0.5 * torch.randn(S_txt, cfg.joint_dim, generator=g)).to(BF); pooled = (0.5 * torch.randn(1, cfg.pooled_dim, generator=g)).to(BF)
    txt_ids = torch.zeros(S_txt, 3)
    vae = AutoencoderKL(synthetic_vae_state_dict(1), device=DEV)
    H, W, K = 16 * hl, 16 * wl, 4
    fwd, img = {}, {}
    for mode, kw in (("bf16", {}), ("fp8", dict(fp8_weights=True)), ("fp8-attn", dict(fp8_weights=True, fp8_attention=True))):
        m = FluxDiT(sd, shape, device=DEV, **kw)
        m.set_lora([(la_dev, 1.0)])
        m.set_positions(txt_ids, img_ids)
        m.set_conditioning(enc.to(DEV), pooled.to(DEV), 3.5)
        fwd[mode] = m.forward(lat0.to(DEV), 0.4375).float().cpu()
        pipe = PBRFluxPipeline(m, vae, device=DEV)
        pipe.load_lora_weights(la_dev, "texture")
        pipe.set_adapters(["texture"], [1.0])
        lat = pipe.denoise(lat0[None, :n_noise], noise_ids, lat0[None, n_noise:], cond_ids, enc.to(DEV)[None], pooled.to(DEV), txt_ids, K, 3.5)
        z = pipe._unpack_latents(lat, H, W, 8)
        z = (z / vae.scaling_factor) + vae.shift_factor
        img[mode] = _postprocess_u8(vae.decode(z.to(BF)))
        torch.cuda.synchronize()
        del m, pipe
        torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ref = dit_ref.flux_forward(_HostView(sd), cfg, lat0.float(), enc.float(), pooled.float(), 0.4375, 3.5, txt_ids, img_ids, loras=[(la, 1.0)], emulate_bf16=True)
    mx = ref.abs().max().item()
    stats = {}
    for mode in ("bf16", "fp8", "fp8-attn"):
        d_or, d_pr = (fwd[mode] - ref).abs(), (fwd[mode] - fwd["bf16"]).abs()
        du = np.abs(img[mode].astype(np.int32) - img["bf16"].astype(np.int32))
        hist = {k: float((du <= k).mean()) for k in (0, 1, 2, 4, 8)}
        stats[mode] = (d_or.max().item() / mx, d_or.mean().item() / mx, d_pr.max().item() / mx, d_pr.mean().item() / mx, hist, int(du.max()))
        print("\n[full depth 19 + 38, full width, S = %d] %-8s one evaluation vs emulating oracle: max %.4g mean %.4g of max|out| (%.3g); vs bf16 product: max %.4g mean %.4g; "
              "K = %d steps + VAE decode, uint8 vs bf16 product: ==0 %.4f <=1 %.4f <=2 %.4f <=4 %.4f <=8 %.4f max %d LSB (image std %.1f)" % (
                  S_txt + img_ids.shape[0], mode, stats[mode][0], stats[mode][1], mx, stats[mode][2], stats[mode][3], K, hist[0], hist[1], hist[2], hist[4], hist[8],
                  stats[mode][5], float(img["bf16"].std())))
        assert torch.isfinite(fwd[mode]).all()
    assert float(img["bf16"].std()) > 4.0, "a constant decoded image would make the uint8 comparison vacuous"
    # the bf16 leg is the test above's; the fp8 legs: bounded (measured values in the printed table; a broken quantiser or scale layout is off by O(1))
    assert stats["bf16"][0] <= 0.03 and stats["bf16"][1] <= 0.005
    for mode in ("fp8", "fp8-attn"):
        assert stats[mode][0] <= 0.10 and stats[mode][1] <= 0.02, (mode, stats[mode])
        assert stats[mode][4][4] >= 0.93 and stats[mode][4][8] >= 0.995 and stats[mode][5] <= 24, (mode, stats[mode])
