"""CPU oracle (TEST INFRASTRUCTURE ONLY) for ONE CALL of the reference's texturing pipeline and for the two-pass schedule around it.

Only tests/ may import this module; the product (unitex_amd/) never does.

What it restates, and from where (paths relative to /root/reference):
  * PBRFluxPipeline.__call__ for batch 1 -- flux_piplines/texturing/pipeline.py:404-700: zero prompt embeddings (:538-543), prepare_latents_and_image_ids
    (:277-402: RNG draw order noise -> dual -> control from ONE generator, VAE encode + posterior sample, pack, ids with offsets), condition = control ++ dual
    (:573-592), the denoise loop with re-pin (:594-681, oracle/dit_ref.denoise_loop), cut of the condition tail, unpack, VAE decode, postprocess to uint8 (:683-692);
  * the orchestrator's two passes -- pipeline.py:246-289: texture pass (control image + reference image, texture adapter weights) -> uint8 image ->
    delight pass (that uint8 image as the control image, no dual image, delight adapter weights) -> uint8 image.
  VaeImageProcessor.preprocess / postprocess, AutoencoderKL, DiagonalGaussianDistribution.sample and the scheduler are diffusers [3p] (unpinned, not
  installed): restated from their published behaviour, as in oracle/dit_ref.py and oracle/vae_ref.py ==> PARITY UNPINNED for those parts; the
  orchestration (draw order, ids, re-pin, mu) is pinned by fixture G1.

emulate_bf16=True rounds through bf16 wherever the reference's bf16 pipeline holds a bf16 tensor (latents, VAE moments, the posterior sample and its
affine map); the VAE itself stays fp32 arithmetic (oracle/vae_ref.py).  emulate_bf16=False is the plain fp32 evaluation of the same call on the same
random draws (the draws are made in bf16, as the reference makes them, and widened) -- the number the bf16 figures are set beside.
"""
import numpy as np
import torch

from . import dit_ref, vae_ref

F32 = torch.float32
BF = torch.bfloat16


def preprocess(image_u8):
    """VaeImageProcessor.preprocess of an RGB uint8 array [H, W, 3] whose size needs no resize: [0, 1] -> [-1, 1], NCHW [3p]."""
    t = torch.from_numpy(np.asarray(image_u8, dtype=np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    return 2.0 * t - 1.0


def postprocess_u8(image, em=True):
    """VaeImageProcessor.postprocess(output_type='pil') up to the PIL wrapper [3p]: denormalize = (x / 2 + 0.5).clamp(0, 1) ON THE VAE'S OUTPUT TENSOR -- a bf16
    tensor in the reference's pipeline, so the halving and the add are bf16 operations (em) -- then .float(), x 255, round, uint8.  Pinned through fixture G1
    (the reference's own pipeline ran this on a stand-in VAE's bf16 output: tests/test_golden_cpu.py)."""
    x = dit_ref._rb(image.float(), em)
    x = dit_ref._rb(dit_ref._rb(x / 2, em) + 0.5, em).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    return (x * 255).round().astype(np.uint8)[0]


def encode_image(vae, image_u8, generator, em):
    """_encode_vae_image (pipeline.py:226-238): retrieve_latents(vae.encode(image), generator) = posterior SAMPLE, then (z - shift) * scale; the reference's
    VAE and latents are bf16 tensors: moments, std, the draw, the sample and both affine steps are bf16 values."""
    x = dit_ref._rb(preprocess(image_u8), em)
    mom = dit_ref._rb(vae.encoder(x), em)
    mean, logvar = mom.chunk(2, dim=1)
    std = dit_ref._rb(torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)), em)
    noise = torch.randn(mean.shape, generator=generator, dtype=BF).to(F32)        # randn_tensor in the VAE's dtype from the shared CPU generator
    z = dit_ref._rb(mean + dit_ref._rb(std * noise, em), em)
    return dit_ref._rb(dit_ref._rb(z - vae.shift_factor, em) * vae.scaling_factor, em)


@torch.no_grad()
def texturing_call(sd, cfg, vae, control_u8, dual_u8, height, width, generator, num_steps, loras, guidance=3.5, max_sequence_length=512,
                   emulate_bf16=True, return_latents=False, encode_fn=None, decode_fn=None, forward_fn=None):
    """one PBRFluxPipeline.__call__ (batch 1, zero prompt embeddings) -> uint8 image [H, W, 3] (and the final noise-token latents).
    encode_fn(image_u8, generator) -> scaled latents [1, 16, h, w], decode_fn(z) -> image, forward_fn(latents, t, img_ids) -> velocity: stand-ins for the
    third-party seams (VAE, transformer), so that THIS function's own logic -- draw order, ids, condition order, re-pin, unpack, postprocess -- can be pinned
    against fixture G1, which the reference's own pipeline produced with the same stand-ins (tests/test_golden_cpu.py)."""
    em = emulate_bf16
    if encode_fn is None:
        encode_fn = lambda img, gen: encode_image(vae, img, gen, em)       # noqa: E731
    if decode_fn is None:
        decode_fn = vae.decode
    HL, WL = 2 * (height // 16), 2 * (width // 16)
    noise = torch.randn((1, 16, HL, WL), generator=generator, dtype=BF).to(F32)                # draw 1
    noise_tok = dit_ref.pack_latents(noise)[0]
    ids = [dit_ref.latent_image_ids(HL // 2, WL // 2)]
    dual_tok = control_tok = None
    if dual_u8 is not None:                                                                     # draw 2
        dl = encode_fn(dual_u8, generator)
        dual_tok = dit_ref.pack_latents(dl)[0]
        dual_ids = dit_ref.latent_image_ids(dl.shape[2] // 2, dl.shape[3] // 2, offset_x=WL // 2, offset_y=HL // 2)
    if control_u8 is not None:                                                                  # draw 3
        cl = encode_fn(control_u8, generator)
        control_tok = dit_ref.pack_latents(cl)[0]
        control_ids = dit_ref.latent_image_ids(cl.shape[2] // 2, cl.shape[3] // 2, offset_x=0, offset_y=HL // 2)
    cond, cond_ids = [], []
    if control_tok is not None:
        cond.append(control_tok); cond_ids.append(control_ids)
    if dual_tok is not None:
        cond.append(dual_tok); cond_ids.append(dual_ids)
    cond_t = torch.cat(cond, 0) if cond else None
    img_ids = torch.cat(ids + cond_ids, 0)
    enc = torch.zeros(max_sequence_length, cfg.joint_dim) if cfg is not None else None        # (a stand-in transformer takes neither)
    pooled = torch.zeros(1, cfg.pooled_dim) if cfg is not None else None
    txt_ids = torch.zeros(max_sequence_length, 3)
    lat = dit_ref.denoise_loop(sd, cfg, noise_tok, cond_t, enc, pooled, txt_ids, img_ids, num_steps, guidance=guidance, loras=loras, emulate_bf16=em,
                               forward_fn=forward_fn)
    z = dit_ref.unpack_latents(lat[None], height, width, 8)
    z = dit_ref._rb(dit_ref._rb(z / vae.scaling_factor, em) + vae.shift_factor, em)
    img = postprocess_u8(decode_fn(z), em)
    return (img, lat) if return_latents else img


def two_pass_schedule(sd, cfg, vae, control_u8, reference_u8, height, width, generator, num_steps, lora_texture, lora_delight, emulate_bf16=True):
    """infer_mv of the orchestrator (pipeline.py:246-289) without its view permutations (fixture G3): texture pass, uint8 hand-off, delight pass."""
    tex = texturing_call(sd, cfg, vae, control_u8, reference_u8, height, width, generator, num_steps, [(lora_texture, 1.0), (lora_delight, 0.0)],
                         emulate_bf16=emulate_bf16)
    out = texturing_call(sd, cfg, vae, tex, None, height, width, generator, num_steps, [(lora_texture, 0.0), (lora_delight, 1.0)],
                         emulate_bf16=emulate_bf16)
    return tex, out
