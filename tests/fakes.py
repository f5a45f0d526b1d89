"""Deterministic stand-ins for the third-party seams (diffusers VAE / transformer / image processor),
IDENTICAL in behaviour to the ones tests/golden/make_golden.py plugged into the reference when the
fixtures were captured.  Exact arithmetic only (powers of two, max-pool, nearest upsample), so CPU and GPU
agree bit for bit.  Test infrastructure."""
import types

import torch


def fake_velocity(hidden_states, timestep, img_ids):
    """hidden [B,S,64] (any float dtype), timestep [B], img_ids [S,3] -> same dtype as hidden."""
    h = hidden_states.float()
    ids = img_ids.to(h.device).float()
    v = 0.5 * h + 0.25 * torch.roll(h, 1, dims=1) - 0.125 * (ids[None, :, 1:2] / 64.0 - ids[None, :, 2:3] / 256.0)
    return (v + timestep.float().to(h.device)[:, None, None]).to(hidden_states.dtype)


class FakeVAE:
    dtype = torch.bfloat16
    latent_channels = 16
    scaling_factor = 0.25
    shift_factor = 0.5
    config = types.SimpleNamespace(block_out_channels=(128, 256, 512, 512), latent_channels=16,
                                   scaling_factor=0.25, shift_factor=0.5)

    def encode(self, x):
        pooled = torch.nn.functional.max_pool2d(x.float(), 8)
        mean = torch.stack([pooled[:, c % 3] * ((c + 1) / 16.0) for c in range(16)], dim=1).to(x.dtype)

        class _D:
            def sample(s, generator=None):
                gdev = generator.device if generator is not None else mean.device
                n = torch.randn(mean.shape, generator=generator, dtype=mean.dtype, device=gdev).to(mean.device)
                return mean + 0.5 * n
        return _D()

    def decode(self, z):
        return torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest").to(z.dtype)


class FakeDiT:
    """FluxDiT-shaped object (set_positions / set_conditioning / forward) computing fake_velocity with torch ops
    on whatever device the latents live on -- lets the product PBRFluxPipeline's host logic be pinned against
    the fixture without involving the real transformer."""

    class _Shape:
        pooled_dim, joint_dim = 768, 4096

    def __init__(self):
        self.shape = self._Shape()
        self.calls = []

    def set_lora(self, adapters):
        self.adapters = adapters

    def set_positions(self, txt_ids, img_ids):
        self.txt_ids, self.img_ids = txt_ids, img_ids

    def set_output_rows(self, n):
        self.out_rows = n         # the real FluxDiT prunes its last block to these rows; the fake computes every row

    def set_conditioning(self, enc, pooled, guidance):
        self.cond_absmax = max(float(enc.abs().max()), float(pooled.abs().max()))
        self.guidance = guidance

    def forward(self, hidden_states, timestep):
        self.calls.append((hidden_states.float().cpu().numpy().copy(), timestep))
        t = torch.tensor([timestep], dtype=torch.float32)
        return fake_velocity(hidden_states[None], t, self.img_ids)[0]
