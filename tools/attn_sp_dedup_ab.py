#!/usr/bin/env python
"""A sequence-parallel rank's attention launch on one GPU, both forms (round 6):
  periodic -- every rank's 64 text rows among the keys: S = P (64 + S_img / P) queries AND keys, key weight 8 / P on the first tile of every rank's block (8 x 32 KBP loop)
  dedup    -- the ranks' identical text rows ONCE among the keys (utx_sp_unpack_qkv_dedup): the same S queries over S_k = 64 + S_img keys in the single-GPU order, key weight 8
              on tile 0 only: the 4 x 64 stream.  Operands are built from one set of rows, so both forms compute the same attention; they differ in summation order only.
Shapes: Hg = the head group of one launch at P = 2 / 4 / 8 (ulysses.pick_head_groups: 3 / 2 / 3 heads), BASELINE's strip (S_img = 50 176)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.flux import ops
from unitex_amd.flux.ulysses import pick_head_groups
BF = torch.bfloat16
S_img, T = 50176, 64
for P in (2, 4, 8):
    I = S_img // P
    S_loc, S, S_k = T + I, P * (T + I), T + S_img
    Hp = 24 // P
    G = pick_head_groups(Hp, S, 256)
    Hg = Hp // G
    g = torch.Generator(device="cuda").manual_seed(P)
    txt_q = torch.randn(Hg, T, 128, generator=g, device="cuda"); txt_k = torch.randn(Hg, T, 128, generator=g, device="cuda"); txt_v = torch.randn(Hg, T, 128, generator=g, device="cuda")
    img_q = torch.randn(Hg, S_img, 128, generator=g, device="cuda"); img_k = torch.randn(Hg, S_img, 128, generator=g, device="cuda"); img_v = torch.randn(Hg, S_img, 128, generator=g, device="cuda")
    sc = 1.4426950408889634 / math.sqrt(128.0)

    def per_rank(t, i):      # rows ordered (rank, [text | image slice])
        return torch.cat([torch.cat([t, i[:, r * I:(r + 1) * I]], 1) for r in range(P)], 1)
    Qh = (per_rank(txt_q, img_q) * sc).to(BF).contiguous()
    K_per, V_per = per_rank(txt_k, img_k).to(BF).contiguous(), per_rank(txt_v, img_v).to(BF).transpose(1, 2).contiguous()
    K_ded, V_ded = torch.cat([txt_k, img_k], 1).to(BF).contiguous(), torch.cat([txt_v, img_v], 1).to(BF).transpose(1, 2).contiguous()
    o_per, o_ded = torch.empty(S, Hg * 128, dtype=BF, device="cuda"), torch.empty(S, Hg * 128, dtype=BF, device="cuda")
    forms = {"periodic": lambda: ops.attention(Qh, K_per, V_per, S=S, scale=0.0, key_bias_log2=math.log2(8.0 / P), key_bias_period=S_loc // 64, out=o_per),
             "dedup": lambda: ops.attention(Qh, K_ded, V_ded, S=S_k, S_q=S, scale=0.0, key_bias_log2=3.0, out=o_ded)}
    for f in forms.values():
        f()
    torch.cuda.synchronize()
    d = (o_per.float() - o_ded.float()).abs()
    print("P = %d: launch of %d heads (G = %d), %d queries; periodic %d keys vs dedup %d keys: max|d| %.3g (max|o| %.3g), %.4f of elements differ" %
          (P, Hg, G, S, S, S_k, d.max().item(), o_ded.float().abs().max().item(), (o_per != o_ded).float().mean().item()), flush=True)
    t = {k: [] for k in forms}
    for _ in range(5):
        for name, f in forms.items():
            f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _r in range(3):
                f()
            b.record(); torch.cuda.synchronize()
            t[name].append(a.elapsed_time(b) / 3)
    fl = {"periodic": 4.0 * S * S * 128 * Hg, "dedup": 4.0 * S * S_k * 128 * Hg}
    m = {k: sorted(v)[2] for k, v in t.items()}
    for k in forms:
        print("   %-9s med %8.3f ms -> %7.1f TF/s (executed FLOPs)" % (k, m[k], fl[k] / m[k] / 1e9), flush=True)
    print("   dedup / periodic time: %.3f  (x %.3f faster); per layer and rank: %d launches -> %.3f vs %.3f ms" % (m["dedup"] / m["periodic"], m["periodic"] / m["dedup"], G, G * m["dedup"], G * m["periodic"]), flush=True)
    del Qh, K_per, V_per, K_ded, V_ded, img_q, img_k, img_v
    torch.cuda.empty_cache()
