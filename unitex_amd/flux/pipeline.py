"""PBRFluxPipeline -- MI355X-native drop-in for the reference's FLUX texturing / delight pipeline
(/root/reference/flux_piplines/texturing/pipeline.py:190-700; delight/pipeline.py is byte-identical).

Same call surface as the reference uses it (pipeline.py:245-278 of the reference orchestrator):
    pipe.set_adapters(adapter_names, adapter_weights)
    out = pipe(prompt, control_image=PIL, dual_image=PIL|None, prompt_embeds=None,
               pooled_prompt_embeds=None, height, width, n_rows, n_cols, num_inference_steps,
               guidance_scale, max_sequence_length, generator);  out.images[0] -> PIL

Semantics kept: T5/CLIP are never run -- zero embeddings; one joint sequence
[text | noise | control | dual]; RNG draw order noise -> dual -> control from ONE shared CPU generator;
mu from the noise-token count only; condition tail re-pinned every step; Euler step in fp32.
The transformer is FluxDiT (HIP kernels behind the C ABI); the scheduler update is the fused HIP
kernel utx_sched_step; the VAE runs on the same library (vae_hip.py).
"""
import os
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from . import ops
from .scheduler import FlowMatchEulerScheduler, calculate_shift
from .transformer import FluxDiT

BF16 = torch.bfloat16


class PBRFluxPipelineOutput:
    def __init__(self, images):
        self.images = images


class PBRFluxPipeline:
    vae_scale_factor = 8

    def __init__(self, transformer: FluxDiT, vae, scheduler: Optional[FlowMatchEulerScheduler] = None,
                 device="cuda:0"):
        self.transformer = transformer
        self.vae = vae
        self.scheduler = scheduler or FlowMatchEulerScheduler()
        self.device = torch.device(device)
        self.text_encoder = None       # reference: text_encoder=None, text_encoder_2=None (pipeline.py:104-105)
        self.text_encoder_2 = None
        self._adapters = {}            # name -> lora dict
        self._num_inference_steps = 28
        self._guidance_scale = 3.5
        self.last_latents = None

    # ---- LoRA adapter control (peft-backed in the reference: pipeline.py:108-112,245,263)
    def load_lora_weights(self, lora, adapter_name):
        if isinstance(lora, str):
            from .lora_io import load_lora_safetensors
            lora = load_lora_safetensors(lora)
        self._adapters[adapter_name] = lora

    def set_adapters(self, adapter_names: List[str], adapter_weights: List[float]):
        self.transformer.set_lora([(self._adapters[n], float(w)) for n, w in zip(adapter_names, adapter_weights)
                                   if n in self._adapters])

    # ---- static helpers with the reference's signatures (pipeline.py:241-275)
    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width, pixel_shuffle=True):
        if pixel_shuffle:
            latents = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
            latents = latents.permute(0, 2, 4, 1, 3, 5)
            return latents.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)
        latents = latents.permute(0, 2, 3, 1)
        return latents.reshape(batch_size, height * width, num_channels_latents)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        batch_size, num_patches, channels = latents.shape
        height = 2 * (int(height) // (vae_scale_factor * 2))
        width = 2 * (int(width) // (vae_scale_factor * 2))
        latents = latents.view(batch_size, height // 2, width // 2, channels // 4, 2, 2)
        latents = latents.permute(0, 3, 1, 4, 2, 5)
        return latents.reshape(batch_size, channels // 4, height, width)

    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype, offset_x=0, offset_y=0, offset_z=0):
        ids = torch.zeros(height, width, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(offset_y, offset_y + height)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(offset_x, offset_x + width)[None, :]
        if offset_z != 0:
            ids[..., 0] = ids[..., 0] + offset_z
        return ids.reshape(height * width, 3).to(device=device, dtype=dtype)

    # ---- image <-> tensor (diffusers VaeImageProcessor [3p])
    def _preprocess(self, image: Image.Image, height, width):
        # VaeImageProcessor(vae_scale_factor=16).preprocess (reference pipeline.py:218,316,365): the target size is floored to a multiple of 16 -- one 2 x 2
        # patch of 8x-compressed latents -- and the image Lanczos-resized when that changes it [3p]
        f = self.vae_scale_factor * 2
        width, height = width - width % f, height - height % f
        if image.size != (width, height):
            image = image.resize((width, height), Image.LANCZOS)
        arr = np.asarray(image.convert("RGB"), dtype=np.float32) / 255.0
        t = torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0)
        return 2.0 * t - 1.0

    @staticmethod
    def _postprocess(image: torch.Tensor):
        img = (image / 2 + 0.5).clamp(0, 1).float().cpu().permute(0, 2, 3, 1).numpy()
        img = (img * 255).round().astype("uint8")
        return [Image.fromarray(i) for i in img]

    def _encode_vae_image(self, image, generator):
        z = self.vae.encode(image).sample(generator)
        return (z - self.vae.shift_factor) * self.vae.scaling_factor

    def prepare_latents_and_image_ids(self, batch_size, num_channels_latents, height, width, dtype, device,
                                      generator, dual_image=None, redux_image=None, control_image=None):
        """reference pipeline.py:277-402 (redux path is dead code there and is not carried)."""
        HL = 2 * (int(height) // (self.vae_scale_factor * 2))
        WL = 2 * (int(width) // (self.vae_scale_factor * 2))
        gdev = generator.device if generator is not None else device
        noise = torch.randn((batch_size, num_channels_latents, HL, WL), generator=generator, device=gdev,
                            dtype=dtype).to(device)                                       # draw 1: noise
        noise_latents = self._pack_latents(noise, batch_size, num_channels_latents, HL, WL)
        noise_ids = self._prepare_latent_image_ids(batch_size, HL // 2, WL // 2, "cpu", torch.float32)
        dual_latents = dual_ids = control_latents = control_ids = None
        if dual_image is not None:
            WD, HD = dual_image.size
            di = self._preprocess(dual_image, HD, WD).to(device=device, dtype=BF16)
            dl = self._encode_vae_image(di, generator).to(dtype)                          # draw 2: dual
            _, CDL, HDL, WDL = dl.shape
            assert HDL == 2 * (HD // (self.vae_scale_factor * 2)) and WDL == 2 * (WD // (self.vae_scale_factor * 2))
            dual_latents = self._pack_latents(dl, batch_size, CDL, HDL, WDL)
            dual_ids = self._prepare_latent_image_ids(batch_size, HDL // 2, WDL // 2, "cpu", torch.float32,
                                                      offset_x=WL // 2, offset_y=HL // 2)
        if control_image is not None:
            WC, HC = control_image.size
            ci = self._preprocess(control_image, HC, WC).to(device=device, dtype=BF16)
            cl = self._encode_vae_image(ci, generator).to(dtype)                          # draw 3: control
            _, CCL, HCL, WCL = cl.shape
            assert HCL == 2 * (HC // (self.vae_scale_factor * 2)) and WCL == 2 * (WC // (self.vae_scale_factor * 2))
            control_latents = self._pack_latents(cl, batch_size, CCL, HCL, WCL)
            control_ids = self._prepare_latent_image_ids(batch_size, HCL // 2, WCL // 2, "cpu", torch.float32,
                                                         offset_x=0, offset_y=HL // 2)
        return noise_latents, noise_ids, dual_latents, dual_ids, None, None, control_latents, control_ids

    # ---- the denoise loop on packed latents (reference pipeline.py:594-681)
    def denoise(self, noise_latents, noise_ids, condition_latents, condition_ids, prompt_embeds, pooled, text_ids,
                num_inference_steps, guidance_scale, step_callback=None):
        tr = self.transformer
        n_noise = noise_latents.shape[1]
        mu = calculate_shift(n_noise, self.scheduler.config.base_image_seq_len, self.scheduler.config.max_image_seq_len,
                             self.scheduler.config.base_shift, self.scheduler.config.max_shift)
        timesteps = self.scheduler.set_timesteps(num_inference_steps, mu)
        if condition_latents is not None:
            latents = torch.cat([noise_latents, condition_latents], dim=1)[0].to(self.device, BF16).contiguous()
            ids = torch.cat([noise_ids, condition_ids], dim=0)
            cond = condition_latents[0].to(self.device, BF16).contiguous()
        else:
            latents = noise_latents[0].to(self.device, BF16).contiguous()
            ids, cond = noise_ids, None
        tr.set_positions(text_ids, ids)          # full id tensors: a sequence-parallel transformer slices them itself
        tr.set_output_rows(n_noise)     # the prediction of the condition tail is never read (re-pinned every step, cut off at the end)
        tr.set_conditioning(prompt_embeds, pooled, guidance_scale)
        # sequence parallel (one job over several GPUs, flux/ulysses.py): every rank drew the same latents from the same CPU
        # generator; it keeps its contiguous slice of the image tokens, the Euler step / re-pin are per token, and one all-gather
        # of the 64-channel latents (6.4 MB at 50 176 tokens) re-assembles the result for the VAE at the end
        sp = getattr(tr, "sp", None)
        n_noise_loc, S_img_full = n_noise, latents.shape[0]
        if sp is not None and sp[1] > 1:
            i0, i1 = tr.local_image_range(S_img_full)
            cond = latents[max(i0, n_noise):i1].clone() if i1 > n_noise else None
            n_noise_loc = min(max(n_noise - i0, 0), i1 - i0)
            latents = latents[i0:i1].clone()
        use_graph = os.environ.get("UTX_HIP_GRAPH", "0") == "1" and num_inference_steps > 2 and hasattr(tr, "capture_graph") \
            and sp is None
        for i in range(num_inference_steps):
            # timestep = t.expand(B).to(latents.dtype); transformer(timestep=timestep / 1000)
            t_bf = torch.tensor(float(timesteps[i]), dtype=torch.float32).to(BF16)
            t_in = float((t_bf / 1000).to(torch.float32))
            if use_graph and i == 1 and not getattr(tr, "_graphs", None):
                tr.capture_graph(warm=False)      # step 0 ran eagerly (first-use setup); the remaining steps replay one HIP graph
            v = tr.forward(latents, t_in)
            # Euler step on the noise tokens + re-pin of the clean condition tail, one fused kernel
            ops.sched_step(latents, v, self.scheduler.dsigma(i), n_noise_tokens=n_noise_loc, cond=cond)
            if step_callback is not None:
                step_callback(i, timesteps[i], latents)
        if sp is not None and sp[1] > 1:
            from ..texturetools.distributed import _all_gather
            full = torch.empty(S_img_full, latents.shape[1], dtype=latents.dtype, device=latents.device)
            _all_gather(full, latents.contiguous(), sp[2])
            latents = full
        return latents[:n_noise].unsqueeze(0)

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, dual_image=None, redux_image=None, control_image=None,
                 height=None, width=None, n_rows=None, n_cols=None, num_inference_steps=28, timesteps=None,
                 guidance_scale=3.5, num_images_per_prompt=1, generator=None, latents=None, prompt_embeds=None,
                 pooled_prompt_embeds=None, output_type="pil", return_dict=True, joint_attention_kwargs=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",),
                 max_sequence_length=512):
        height = height or 1024
        width = width or 1024
        if redux_image is not None:
            raise NotImplementedError("redux_image is dead code in the reference (SURVEY 0.2) and is not supported")
        if timesteps is not None:
            # the reference hands `timesteps` AND its own sigmas to diffusers' retrieve_timesteps (pipeline.py:603-610), which refuses the pair [3p]
            raise ValueError("custom timesteps are not supported (the reference's call raises for them as well)")
        if num_images_per_prompt not in (None, 1):
            raise NotImplementedError("num_images_per_prompt > 1: the reference's texture pipeline always asks for one image (pipeline.py:246-261)")
        batch_size = 1
        sh = self.transformer.shape
        # zero embeddings stand in for CLIP / T5 (reference pipeline.py:538-543)
        if self.text_encoder is None:
            pooled_prompt_embeds = torch.zeros((batch_size, sh.pooled_dim), device=self.device, dtype=BF16)
        if self.text_encoder_2 is None:
            prompt_embeds = torch.zeros((batch_size, max_sequence_length, sh.joint_dim), device=self.device, dtype=BF16)
        text_ids = torch.zeros(prompt_embeds.shape[1], 3)
        (noise_latents, noise_ids, dual_latents, dual_ids, _, _, control_latents, control_ids) = \
            self.prepare_latents_and_image_ids(batch_size, self.vae.latent_channels, height, width, BF16,
                                               self.device, generator, dual_image=dual_image,
                                               control_image=control_image)
        if dual_latents is not None and control_latents is not None:
            condition_latents = torch.cat([control_latents, dual_latents], dim=1)
            condition_ids = torch.cat([control_ids, dual_ids], dim=0)
        elif dual_latents is not None:
            condition_latents, condition_ids = dual_latents, dual_ids
        elif control_latents is not None:
            condition_latents, condition_ids = control_latents, control_ids
        else:
            condition_latents = condition_ids = None
        step_callback = None
        if callback_on_step_end is not None:
            # reference pipeline.py:664-671: callback(self, i, t, {name: tensor}) after every scheduler step; a returned "latents" replaces the running latents
            # (here: [1, tokens, 64] bf16 -- noise tokens followed by the clean condition tail; a sequence-parallel rank sees its own token slice)
            unknown = [k for k in callback_on_step_end_tensor_inputs if k not in ("latents", "prompt_embeds")]
            if unknown:
                raise ValueError("callback_on_step_end_tensor_inputs %s: only 'latents' and 'prompt_embeds' exist in this loop" % unknown)

            def step_callback(i, t, cur):
                kw = {}
                if "latents" in callback_on_step_end_tensor_inputs:
                    kw["latents"] = cur.unsqueeze(0)
                if "prompt_embeds" in callback_on_step_end_tensor_inputs:
                    kw["prompt_embeds"] = prompt_embeds
                out = callback_on_step_end(self, i, torch.as_tensor(float(t)), kw) or {}
                new = out.get("latents")
                if new is not None and new.data_ptr() != cur.data_ptr():
                    cur.copy_(new.reshape(cur.shape).to(cur.dtype))
        lat = self.denoise(noise_latents, noise_ids, condition_latents, condition_ids, prompt_embeds,
                           pooled_prompt_embeds, text_ids, num_inference_steps, guidance_scale, step_callback=step_callback)
        self.last_latents = lat
        if output_type == "latent":
            return PBRFluxPipelineOutput(images=lat)
        lat = self._unpack_latents(lat, height, width, self.vae_scale_factor)
        lat = (lat / self.vae.scaling_factor) + self.vae.shift_factor
        image = self.vae.decode(lat.to(BF16))
        images = self._postprocess(image)
        if not return_dict:
            return (images,)
        return PBRFluxPipelineOutput(images=images)
