#!/usr/bin/env python
"""Second bisect of the fp8 pruned plan's rare run-to-run difference: device-side snapshots INSIDE the step (copies inserted into the Python launch list at the
block boundaries, stream-ordered, no host synchronisation), so the first buffer that differs is seen where it is produced.  usage: fp8_plan_bisect2.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dit_ref
from unitex_amd.flux.transformer import FluxDiT, FluxShape
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
MODE = sys.argv[2] if len(sys.argv) > 2 else "plain"      # "flush": a 512 MB fill in front of every attention (evicts every L2); "kv": snapshots of Qh / Kh / Vt text rows in front of the first attention
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
junk2 = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
m = FluxDiT(sd, shape, device="cuda:0", fp8_weights=True)
m.set_lora([(lora, 1.0)])
m.set_positions(torch.zeros(S_txt, 3), img_ids)
m.set_output_rows(4096)
m.set_conditioning(enc, pooled, 3.5)
p = next(iter(m._plans.values()))
m.lib.utx_plan_free(p["cplan"]); p["cplan"] = None
ws, plan = p["ws"], p["plan"]
orig_launch = FluxDiT._launch
def launch(self, fn, d, st, timed=True):
    if fn == "dbg_copy":
        d[1].copy_(d[0]); return
    if fn == "dbg_flush":
        junk2.fill_(7); return
    return orig_launch(self, fn, d, st, timed)
FluxDiT._launch = launch
# find the block boundaries: a top-level utx_ln_mod entry over all S rows starts a single block; the first "par" entry is the double block's first half
S = p["S_txt"] + p["S_img"]
snaps = {}
def snap_entry(name, t):
    snaps[name] = torch.empty_like(t)
    return ("dbg_copy", (t, snaps[name]))
new, n_ln, n_par = [], 0, 0
for e in plan:
    fn, d = e
    if fn is m.lib.utx_ln_mod and d.n_tok == S:
        new.append(snap_entry("h before single block %d" % n_ln, ws["h"])); n_ln += 1
    if fn == "par":
        if n_par == 0:
            new.append(snap_entry("mod (all AdaLN vectors)", ws["mod"])); new.append(snap_entry("h after the embedders", ws["h"]))
        n_par += 1
    if fn is m.lib.utx_attn_fwd_bf16_ws:
        if MODE == "flush":
            new.append(("dbg_flush", None))
        first_attn = not any(n_.startswith("Kh BEFORE") for n_ in snaps)
        if MODE == "kv" and first_attn:
            new.append(snap_entry("Qh BEFORE attention 0", ws["Qh"])); new.append(snap_entry("Kh BEFORE attention 0", ws["Kh"])); new.append(snap_entry("Vt BEFORE attention 0", ws["Vt"]))
    new.append(e)
    if fn is m.lib.utx_attn_fwd_bf16_ws and MODE == "kv" and first_attn:
        new.append(snap_entry("Qh AFTER attention 0", ws["Qh"])); new.append(snap_entry("Kh AFTER attention 0", ws["Kh"])); new.append(snap_entry("Vt AFTER attention 0", ws["Vt"]))
    if fn == "par" and n_par == 1:
        new.append(snap_entry("qkv after the double block's first half", ws["qkv"]))
    if fn is m.lib.utx_attn_fwd_bf16_ws:
        k = sum(1 for n_ in snaps if n_.startswith("attention output"))
        new.append(snap_entry("attention output %d" % k, ws["attn"] if k == 0 else ws["cat"][:, :3072]))
new.append(snap_entry("out", ws["out"]))
p["plan"] = new
order = list(snaps.keys())
print("snapshots:", order, flush=True)
m.forward(lat, 0.5); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in snaps.items()}
nbad = 0
for i in range(reps):
    if i % 3 == 1:
        junk.fill_(i & 255)
    m.forward(lat, 0.5)
    if torch.equal(snaps["out"][:4096].view(torch.int16), ref["out"][:4096].view(torch.int16)):
        continue
    nbad += 1
    diffs = [(k, int((snaps[k] != ref[k]).sum())) for k in order if not torch.equal(snaps[k], ref[k])]
    print("rep %d: FIRST differing snapshot: %s (%d elements); all: %s" % (i, diffs[0][0], diffs[0][1], [(a[:28], b) for a, b in diffs]), flush=True)
    a, b = snaps[diffs[0][0]], ref[diffs[0][0]]
    if a.dim() == 3:
        idx = torch.nonzero(a != b)
        print("      differing (head, row, channel): %s" % idx.tolist(), flush=True)
        h_, r_ = int(idx[0, 0]), int(idx[0, 1])
        print("      now %s" % [round(float(x), 5) for x in a[h_, r_, :].float()][:24], flush=True)
        print("      ref %s" % [round(float(x), 5) for x in b[h_, r_, :].float()][:24], flush=True)
        # the qkv row this Q row was computed from (the workspace still holds THIS forward's last-block qkv, not the double block's: use the snapshot)
        qs_ = snaps.get("qkv after the double block's first half")
        if qs_ is not None:
            row = qs_[r_, h_ * 128:(h_ + 1) * 128].float(); rr = ref["qkv after the double block's first half"][r_, h_ * 128:(h_ + 1) * 128].float()
            print("      qkv snapshot row equal to ref: %s ; sum of squares %.9g" % (bool(torch.equal(row, rr)), float((row * row).sum())), flush=True)
    if a.dim() == 2:
        bad = (a != b)
        rows = torch.nonzero(bad.any(1)).flatten(); cols = torch.nonzero(bad.any(0)).flatten()
        print("      rows %d..%d (%d distinct: %s ...), cols %d..%d (%d distinct: %s ...)" % (int(rows.min()), int(rows.max()), rows.numel(), rows[:6].tolist(), int(cols.min()), int(cols.max()), cols.numel(), cols[:6].tolist()), flush=True)
        r_, c_ = int(rows[0]), int(cols.min())
        print("      row %d cols %d..: now %s | ref %s | max |d| over the buffer %.4g" % (r_, c_, [round(float(x), 4) for x in a[r_, c_:c_ + 6].float()], [round(float(x), 4) for x in b[r_, c_:c_ + 6].float()],
              float((a.float() - b.float()).abs().max())), flush=True)
print("%d of %d forwards differ" % (nbad, reps), flush=True)
