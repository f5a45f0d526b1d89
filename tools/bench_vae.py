#!/usr/bin/env python
"""Wall time of the HIP AutoencoderKL at the reference's strip sizes (decode of the texture strip, encode of the
control strip + the 512^2 reference image).  usage: python tools/bench_vae.py [H W]..."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.flux.vae_hip import AutoencoderKL
vae = AutoencoderKL.synthetic(seed=0, device="cuda:0")
sizes = [(512, 512), (512, 3072), (1024, 6144)]
for (H, W) in sizes:
    img = (torch.rand(1, 3, H, W, device="cuda:0") * 2 - 1).to(torch.bfloat16)
    z = torch.randn(1, 16, H // 8, W // 8, device="cuda:0").to(torch.bfloat16)
    for name, fn in (("encode", lambda: vae.encode(img).mean), ("decode", lambda: vae.decode(z))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("vae %s %4dx%-5d: %8.2f ms   peak mem %.1f GB  finite=%s" % (name, H, W, dt * 1e3, torch.cuda.max_memory_allocated() / 2**30,
                                                                      bool(torch.isfinite(out.float()).all())), flush=True)
