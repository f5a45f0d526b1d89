#!/usr/bin/env python
"""End-to-end run of CustomRGBTextureFullPipeline at the reference's operating point (FLUX.1-dev-shaped synthetic
weights, rank-64 LoRAs, 512x3072 strip, 2 x 28 steps, HIP VAE, 2048^2 atlas, 120-frame turntable) on a synthetic
UV-mapped mesh: metric (ii) of SURVEY 8d, seconds per mesh texture, measured rather than composed.
usage: python tools/run_full_pipeline.py [--faces 50000] [--steps 28] [--no-uv]"""
import argparse, os, sys, tempfile, time
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.pipeline import CustomRGBTextureFullPipeline
from unitex_amd.texturetools import meshes

ap = argparse.ArgumentParser()
ap.add_argument("--faces", type=int, default=50000)
ap.add_argument("--steps", type=int, default=28)
ap.add_argument("--no-uv", action="store_true", help="feed a mesh without UVs (exercises clean / unwrap)")
ap.add_argument("--out", default=None)
ap.add_argument("--view", type=int, default=512, help="per-view resolution: 512 = reference, 1024 = BASELINE configs[1..2]")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--fp8", action="store_true", help="speedup_mode='fp8': the big linears on OCP MX fp8 operands (BASELINE configs[4] numerics)")
a = ap.parse_args()
out = a.out or tempfile.mkdtemp(prefix="utx_full_")
v, f, uv = meshes.sphere_with_faces(a.faces)
mesh_path = os.path.join(out, "in.obj")
meshes.save_obj(mesh_path, v, f, None if a.no_uv else uv)
yy, xx = np.mgrid[0:768, 0:768]
Image.fromarray(np.stack([xx % 256, yy % 256, (xx + yy) % 256], -1).astype(np.uint8)).save(os.path.join(out, "ref.png"))
t0 = time.perf_counter()
pipe = CustomRGBTextureFullPipeline(pretrain_models=None, super_resolutions=False, seed=63, num_inference_steps=a.steps, view_size=a.view,
                                    speedup_mode="fp8" if a.fp8 else None)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("pipeline construction (synthetic 12B-parameter weights): %.1f s" % (t1 - t0), flush=True)
for rep in range(a.reps):
    t2 = time.perf_counter()
    png, glb = pipe(os.path.join(out, "run%d" % rep), os.path.join(out, "ref.png"), mesh_path)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("run %d: sec_per_mesh_texture = %.2f s  (%d steps x 2 passes)  -> %s (%.1f MB)" %
          (rep, t3 - t2, a.steps, glb, os.path.getsize(glb) / 1e6), flush=True)
tex = np.asarray(Image.open(os.path.join(out, "run%d" % (a.reps - 1), "cache", "wo_LTM", "completed_uv.png")))
print("atlas", tex.shape, "mean", float(tex.mean()), "peak mem %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))
