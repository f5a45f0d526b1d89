"""unitex_amd -- MI355X-native (gfx950) hot path of UniTEX: the FLUX-DiT multi-view texture
denoise loop and the TextureTools render / UV back-projection, behind the reference's
CustomRGBTextureFullPipeline call surface.  Host code is Python on PyTorch-ROCm (memory, streams,
torch.distributed); all compute goes through the C ABI of libunitex_hip.so (include/unitex_hip.h).
"""
__version__ = "0.1.0"
