#!/usr/bin/env python
"""Back-to-back forwards of the full-width plan (1 double + 2 single blocks, 9280 executed tokens), the comparison being the only synchronisation: how many
differ from the first, per configuration (bf16 / MX fp8, last block pruned or not, one or two streams, C replay or Python launch list).
usage: python tools/plan_determinism_matrix.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dit_ref
from unitex_amd import _lib
from unitex_amd.flux.transformer import FluxDiT, FluxShape
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
CASES = {"all": ((True, 4096, True, True), (True, None, True, True), (False, 4096, True, True), (False, None, True, True), (True, 4096, False, True), (True, 4096, True, False)),
         "streams": ((True, 4096, False, True), (True, 4096, True, True), (False, 4096, True, True), (False, None, True, True)),
         "soak": ((True, 4096, True, True), (False, 4096, True, True), (True, None, True, True))}
for fp8, rows, two_streams, cplan in CASES[sys.argv[2] if len(sys.argv) > 2 else "all"]:
    m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=fp8)
    m.set_text_stream(two_streams)
    m.set_lora([(lora, 1.0)])
    m.set_positions(torch.zeros(S_txt, 3), img_ids)
    m.set_output_rows(rows)
    m.set_conditioning(enc, pooled, 3.5)
    p = next(iter(m._plans.values()))
    if not cplan:
        m.lib.utx_plan_free(p["cplan"]); p["cplan"] = None
    n = p["n_out"]
    ref = m.forward(lat, 0.5)[:n].clone()
    bad = 0
    for i in range(reps):
        if i % 3 == 1:
            junk.fill_(i & 255)
        o = m.forward(lat, 0.5)[:n]
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            bad += 1
    print("fp8=%s pruned=%s two_streams=%s c_replay=%s: %d of %d forwards differ from the first" % (fp8, rows is not None, two_streams, cplan, bad, reps), flush=True)
    del m
