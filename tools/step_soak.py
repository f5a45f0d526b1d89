#!/usr/bin/env python
"""Soak of the WHOLE denoise step of the bench workload (57 blocks, full width, S = 50 688 -> 50 240 executed, LoRA r64, the default launch path = C plan replay) on FIXED inputs:
every forward's output must equal the first one's bits -- the generated attention stream and the generated GEMM K loop inside the real plan, with the chip's power state moving between
attention / GEMM / elementwise phases as it does in production.  usage: python tools/step_soak.py [forwards]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unitex_amd import _lib
from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
from unitex_amd.flux.transformer import FluxDiT, FluxShape
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
S_txt, n_noise, n_ctrl, n_dual = bench.token_counts("strip1024x6")
S_img = n_noise + n_ctrl + n_dual
shape = FluxShape()
sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
model = FluxDiT(sd, shape, device=dev)
model.set_lora([(synthetic_lora(sd, shape, rank=64, seed=1, device=dev), 1.0)])
h_px, w_px, dual_px, _ = bench.WORKLOADS["strip1024x6"]
HL, WL = h_px // 16, w_px // 16
ids = [torch.zeros(HL, WL, 3), torch.zeros(HL, WL, 3), torch.zeros(dual_px // 16, dual_px // 16, 3)]      # noise strip, control strip, dual image: bench.py's position ids
for t, (oy, ox) in zip(ids, [(0, 0), (HL, 0), (HL, WL)]):
    t[..., 1] += torch.arange(oy, oy + t.shape[0])[:, None]
    t[..., 2] += torch.arange(ox, ox + t.shape[1])[None, :]
img_ids = torch.cat([t.reshape(-1, 3) for t in ids], 0)
assert img_ids.shape[0] == S_img
model.set_positions(torch.zeros(S_txt, 3), img_ids)
model.set_output_rows(n_noise)      # last-block pruning, as the bench and the texturing pipeline run it
model.set_conditioning(torch.zeros(S_txt, shape.joint_dim, dtype=torch.bfloat16, device=dev), torch.zeros(1, shape.pooled_dim, dtype=torch.bfloat16, device=dev), 3.5)
g = torch.Generator(device=dev).manual_seed(11)
lat = torch.randn(S_img, shape.in_channels, generator=g, device=dev).to(torch.bfloat16)
ref = model.forward(lat, 0.5).clone(); torch.cuda.synchronize()
assert torch.isfinite(ref.float()).all()
bad, t0 = 0, time.time()
for i in range(reps):
    out = model.forward(lat, 0.5)
    torch.cuda.synchronize()
    nd = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
    bad += nd != 0
    if nd:
        print("forward %d: %d of %d elements differ from the first forward" % (i, nd, out.numel()), flush=True)
print("full-size step (strip1024x6, %d tokens executed, options %s): %d forwards, %.1f s, %d differing from the first" %
      (model.text_rows + S_img if model.text_rows else S_txt + S_img, {k: v for k, v in _lib.get_options().items() if k in ("UTX_ATTN_Q64", "UTX_GEMM_FASTK", "UTX_GEMM_STREAMK")}, reps, time.time() - t0, bad), flush=True)
sys.exit(1 if bad else 0)
