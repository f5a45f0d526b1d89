#!/usr/bin/env python
"""Which side-stream kernel has to run beside the image half's q / k / v post-processing for the sporadic corruption of a Q / K row to appear?  fp8 full-width
pruned plan, Python launch list, arms = surgery on the first two-stream section of the double block:
   base       : as built (text half = ln_mod, LoRA-down GEMM, QKV GEMM, qkv_post on the side stream)
   post_main  : the text half's qkv_post moved behind the join, onto the main stream
   gemm_main  : the text half's GEMMs + qkv_post moved behind the join (side stream = ln_mod only)
usage: two_stream_probe.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
if os.environ.get("UTX_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UTX_LIB"])      # A/B of a differently built library (e.g. dit_elementwise.hip without packed fp32 instructions)
from oracle import dit_ref
from unitex_amd.flux.transformer import FluxDiT, FluxShape
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
cfg = dit_ref.FluxConfig(num_double=1, num_single=2)
shape = FluxShape(num_double=1, num_single=2)
S_txt = 512
img_ids = torch.cat([dit_ref.latent_image_ids(32, 128), dit_ref.latent_image_ids(32, 128, offset_y=32), dit_ref.latent_image_ids(32, 32, offset_x=128, offset_y=32)], 0)
enc = torch.zeros(S_txt, cfg.joint_dim).to(BF).cuda(); pooled = torch.zeros(1, cfg.pooled_dim).to(BF).cuda()
sd = dit_ref.make_synthetic_state_dict(cfg, seed=0)
lora = dit_ref.make_synthetic_lora(cfg, sd, rank=64, seed=2)
g = torch.Generator().manual_seed(9)
lat = torch.randn(img_ids.shape[0], 64, generator=g).to(BF).cuda()
junk = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
ARMS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["base", "post_main", "gemm_main", "base"]
for arm in ARMS:
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
    if arm != "no_lora":
        m.set_lora([(lora, 1.0)])
    m.set_positions(torch.zeros(S_txt, 3), img_ids)
    m.set_output_rows(4096)
    m.set_conditioning(enc, pooled, 3.5)
    p = next(iter(m._plans.values()))
    m.lib.utx_plan_free(p["cplan"]); p["cplan"] = None
    new, done = [], False
    for fn, d in p["plan"]:
        if fn == "par" and not done and arm in ("post_main", "gemm_main"):
            main_ops, side_ops, e0, e1 = d
            keep = 1 if arm == "gemm_main" else len(side_ops) - 1
            new.append(("par", (main_ops, side_ops[:keep], e0, e1)))
            new.extend(side_ops[keep:])
            done = True
        elif fn == "par" and not done and arm == "qkv_main":
            # side = ln_mod + LoRA-down GEMM; the text half's QKV GEMM (the MX fp8 small-M kernel in this plan) + qkv_post run behind the join on the main stream
            main_ops, side_ops, e0, e1 = d
            new.append(("par", (main_ops, side_ops[:2], e0, e1)))
            new.extend(side_ops[2:])
            done = True
        elif fn == "par" and not done and arm == "lora_down_main":
            # side = ln_mod, QKV GEMM, qkv_post; the LoRA-down GEMM (side_ops[1]) runs on the main stream IN FRONT of the section
            main_ops, side_ops, e0, e1 = d
            new.append(side_ops[1])
            new.append(("par", (main_ops, [side_ops[0]] + side_ops[2:], e0, e1)))
            done = True
        else:
            new.append((fn, d))
    p["plan"] = new
    ref = m.forward(lat, 0.5)[:4096].clone()
    bad = 0
    for i in range(reps):
        if i % 3 == 1:
            junk.fill_(i & 255)
        o = m.forward(lat, 0.5)[:4096]
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            bad += 1
    print("%-10s: %d of %d forwards differ from the first" % (arm, bad, reps), flush=True)
    del m
