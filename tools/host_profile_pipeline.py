#!/usr/bin/env python
"""cProfile of one CustomRGBTextureFullPipeline call with a short denoise (2 steps): where the HOST time of the stages around the DiT goes
(file I/O, PNG / GLB codecs, mesh parsing, synchronising copies).  The GPU work of those stages is ~25 ms; everything else in their
wall time is host.  usage: python tools/host_profile_pipeline.py [--faces 50000]"""
import argparse, cProfile, os, pstats, sys, tempfile
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.pipeline import CustomRGBTextureFullPipeline
from unitex_amd.texturetools import meshes
ap = argparse.ArgumentParser()
ap.add_argument("--faces", type=int, default=50000)
a = ap.parse_args()
out = tempfile.mkdtemp(prefix="utx_prof_")
v, f, uv = meshes.sphere_with_faces(a.faces)
mesh_path = os.path.join(out, "in.obj")
meshes.save_obj(mesh_path, v, f, uv)
yy, xx = np.mgrid[0:768, 0:768]
Image.fromarray(np.stack([xx % 256, yy % 256, (xx + yy) % 256], -1).astype(np.uint8)).save(os.path.join(out, "ref.png"))
pipe = CustomRGBTextureFullPipeline(pretrain_models=None, super_resolutions=False, seed=63, num_inference_steps=2, view_size=512)
pipe(os.path.join(out, "warm"), os.path.join(out, "ref.png"), mesh_path)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pipe(os.path.join(out, "run"), os.path.join(out, "ref.png"), mesh_path)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(30)
