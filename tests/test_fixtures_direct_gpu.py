"""Reference fixtures straight against the HIP kernels -- no oracle in between.

The other GPU tests compare HIP with `oracle/`, and `tests/test_golden_cpu.py` compares `oracle/` with the fixtures the reference's own Python produced
(tests/golden/make_golden.py).  That chain is sound but makes the oracle a single point of failure exactly where the reference's code IS available
(VERDICT r5, weak #3).  Here the arrays of G2 (attention core, flux_piplines/texturing/attention_processor.py:31-110), G5 (pull-push
TextureTools/texturetools/image/mip.py:51-95, lens blur image/lens_blur.py:260-280, visibility dilation) and G6/G7 (uv_to_pcd +
bake_mv_to_uv_reproject_blur, render/nvdiffrast/renderer_inverse.py:243-365, 574-633) are loaded and fed through the C ABI; the expected values are the
reference's outputs stored in the same .npz.  This module must never import `oracle` (asserted below)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _no_oracle():
    """this module imports nothing from oracle/ (checked on its syntax tree: a whole-suite run has the oracle in sys.modules through other test modules)"""
    import ast
    tree = ast.parse(open(os.path.abspath(__file__)).read())
    for n in ast.walk(tree):
        if isinstance(n, ast.Import):
            assert not any(a.name.split(".")[0] == "oracle" for a in n.names)
        if isinstance(n, ast.ImportFrom):
            assert (n.module or "").split(".")[0] != "oracle"


# ------------------------------------------------------------------------------------------------ G2: the attention core
def _g2_hip(f, x_tok, c_tok, single):
    """utx_gemm_bf16 (q|k|v projection, bias) -> utx_qkv_post (RMS norm, RoPE, head-major relayout, Q pre-scaled) -> utx_attn_fwd_bf16 -> out projections.
    Fixture shape: D = 256 = 2 heads x 128, 8 text + 56 image tokens (one 64-key tile)."""
    from unitex_amd.flux import ops
    H, D = 2, 256
    dev = "cuda"
    t = lambda k: torch.from_numpy(f[k]).to(dev)
    W = lambda names: torch.cat([t(n + ".weight") for n in names], 0).to(BF).contiguous()
    Bv = lambda names: torch.cat([t(n + ".bias") for n in names], 0).to(BF).contiguous()
    cos, sin = t("cos").contiguous(), t("sin").contiguous()
    S_txt, S_img = c_tok.shape[0], x_tok.shape[0]
    S = S_txt + S_img
    S_pad = (S + 63) // 64 * 64
    Qh = torch.zeros(H, S_pad, 128, dtype=BF, device=dev)
    Kh = torch.zeros_like(Qh)
    Vt = torch.zeros(H, 128, S_pad, dtype=BF, device=dev)
    q_scale = 1.4426950408889634 / math.sqrt(128.0)      # the product's form: Q leaves qkv_post as Q * scale * log2(e), the kernel exponentiates base 2
    img_names = ("to_q", "to_k", "to_v")
    txt_names = img_names if single else ("add_q_proj", "add_k_proj", "add_v_proj")
    nq, nk = ("norm_q", "norm_k")
    tq, tk = (nq, nk) if single else ("norm_added_q", "norm_added_k")
    qkv_x = ops.gemm(x_tok.to(BF).contiguous(), W(img_names), bias=Bv(img_names))
    qkv_c = ops.gemm(c_tok.to(BF).contiguous(), W(txt_names), bias=Bv(txt_names))
    ops.qkv_post(qkv_x, 0, D, 2 * D, t(nq + ".weight").to(BF), t(nk + ".weight").to(BF), cos, sin, Qh, Kh, Vt, S_img, S_txt, H, q_scale=q_scale)
    ops.qkv_post(qkv_c, 0, D, 2 * D, t(tq + ".weight").to(BF), t(tk + ".weight").to(BF), cos, sin, Qh, Kh, Vt, S_txt, 0, H, q_scale=q_scale)
    a = ops.attention(Qh, Kh, Vt, S=S, scale=0.0)          # [S, H * 128] bf16
    if single:
        torch.cuda.synchronize()
        return a.float().cpu()
    out_c = ops.gemm(a[:S_txt].contiguous(), W(("to_add_out",)), bias=Bv(("to_add_out",)))
    out_x = ops.gemm(a[S_txt:].contiguous(), W(("to_out.0",)), bias=Bv(("to_out.0",)))
    torch.cuda.synchronize()
    return out_x.float().cpu(), out_c.float().cpu()


def test_g2_attention_core_fixture_through_the_hip_kernels():
    """Tolerance, stated: the fixture is the reference processor's fp32 evaluation; the HIP path holds activations, weights and P in bf16 (8 significant
    bits) with fp32 accumulation, rounding at five tensor boundaries (projection, normalised / rotated q and k, probabilities, attention output, output
    projection).  Bound asserted: max |d| <= 2.5 % and mean |d| <= 0.4 % of max |reference output| -- the same form and size as the full-width DiT bound in
    test_fullsize_gpu.py; measured on MI355X: see profiles/r06_fixtures_direct.log."""
    _no_oracle()
    f = _load("g2_attn_core.npz")
    x, c = torch.from_numpy(f["x"][0]).cuda(), torch.from_numpy(f["c"][0]).cuda()
    out_x, out_c = _g2_hip(f, x, c, single=False)
    a1 = _g2_hip(f, x, c, single=True)
    for name, got, ref in (("out_x", out_x, f["out_x"][0]), ("out_c", out_c, f["out_c"][0]), ("out_single", a1, f["out_single"][0])):
        ref = torch.from_numpy(ref)
        mx = ref.abs().max().item()
        d = (got - ref).abs()
        print("G2 %s: max |d| %.5f (%.3f %% of max |ref| %.3f), mean |d| %.6f (%.4f %%)" % (name, d.max().item(), 100 * d.max().item() / mx, mx, d.mean().item(), 100 * d.mean().item() / mx))
        assert torch.isfinite(got).all()
        assert d.max().item() <= 0.025 * mx, "%s: max |d| %g vs max |ref| %g" % (name, d.max().item(), mx)
        assert d.mean().item() <= 0.004 * mx, "%s: mean |d| %g" % (name, d.mean().item())


# ------------------------------------------------------------------------------------------------ G5: image operations
def test_g5_pull_push_lens_blur_dilation_fixtures_through_the_hip_kernels():
    """pull-push: the HIP kernel is bit-identical to the oracle, and the oracle is within 2e-6 of the reference (the reference's bilinear up-sampling is a torch
    interpolate whose weights are computed in another order) -- so 2e-6 against the fixture, and bit-exact on the texels the mask keeps untouched.  Lens blur:
    the collapsed real 7 x 7 kernel against the reference's five separable complex passes: 1e-5.  Visibility dilation (3 x 3 `or` inside the coverage): exact."""
    _no_oracle()
    from unitex_amd.texturetools import ops
    f = _load("g5_image_ops.npz")
    for n in (64, 256):
        kd = torch.from_numpy(np.ascontiguousarray(f["pp%d_kd" % n][0].transpose(1, 2, 0))).cuda()
        mask = torch.from_numpy(f["pp%d_mask" % n][0, 0].astype(np.uint8)).cuda()
        out = ops.pull_push(kd, mask).cpu().numpy().transpose(2, 0, 1)
        ref = f["pp%d_out" % n][0]
        d = np.abs(out - ref)
        print("G5 pull_push %d: max |d| %.3g, texels equal %.4f" % (n, d.max(), (out == ref).mean()))
        assert d.max() < 2e-6, "pull_push %d: %g" % (n, d.max())
        keep = f["pp%d_mask" % n][0, 0]
        assert np.array_equal(out[:, keep], f["pp%d_kd" % n][0][:, keep]), "masked-in texels must pass through untouched"
    src = torch.from_numpy(np.ascontiguousarray(f["lb_in"][0].transpose(1, 2, 0))).cuda()
    seam = torch.ones(src.shape[:2], dtype=torch.uint8, device="cuda")
    blur = ops.lens_blur_seam(src, seam).cpu().numpy().transpose(2, 0, 1)
    d = np.abs(blur - f["lb_out"][0]).max()
    print("G5 lens blur: max |d| %.3g" % d)
    assert d < 1e-5
    none = ops.lens_blur_seam(src, torch.zeros_like(seam)).cpu().numpy()
    assert np.array_equal(none, src.cpu().numpy()), "off the seam the blur must copy"
    din = torch.from_numpy(f["dil_in"][..., 0].astype(np.uint8)).cuda().contiguous()
    rast = torch.ones(din.shape[1], din.shape[2], 4, dtype=torch.float32, device="cuda")
    got = ops.dilate_visibility(din, torch.ones_like(din), rast).cpu().numpy().astype(bool)
    assert np.array_equal(got, f["dil_out"][..., 0]), "visibility dilation"


# ------------------------------------------------------------------------------------------------ G6 / G7: the back-projection chain through infer()
def test_g67_backprojection_fixture_through_infer():
    """NVDiffRendererInverse.infer (method='reproject', the pipeline's path) on the fixture's mesh, cameras and images at its 96^2 atlas / 48^2 views.  The fixture's
    alpha carries holes punched by the generator (so that unseen texels exist): infer() computes the view alpha itself (coverage -- checked against `mv_alpha`,
    exact), the test substitutes the holed alpha at that one seam, as the generator did on the reference's side.
    Exact: coverage mask, per-view visibility (<= 2 texels: alpha > 0.999 is a knife edge under the bilinear sample), the composite's winner index wherever the
    visibility agrees.  Colours: 2e-6 per view layer; final atlas median < 1e-6 and < 0.2 % of texels beyond 1e-4 (NN-fill ties / blur powf), as the oracle itself."""
    _no_oracle()
    from unitex_amd.texturetools.renderer_inverse import NVDiffRendererInverse, PRIORITY
    f = _load("g67_backprojection.npz")
    HW, T = f["images"].shape[1], f["mask_2d"].shape[1]
    seen = {}

    class Inv(NVDiffRendererInverse):
        def mv_to_pcd(self, *a, **kw):
            out = super().mv_to_pcd(*a, **kw)
            seen["mv_alpha"] = out["alpha"].clone()
            out["alpha"] = torch.from_numpy(f["alpha"][..., 0]).to(out["alpha"].device).contiguous()
            return out
    inv = Inv(device="cuda").update_from_arrays(f["verts"], f["faces"], f["uvs"])
    assert inv.index == PRIORITY
    textured, mask_vis, mask_2d, color_2d, layers, vis = inv.infer(None, c2ws=f["c2ws"], intrinsics=f["intr"], image_attrs=torch.from_numpy(f["images"]), H=HW, W=HW, H2D=T, W2D=T,
                                                                    perspective=False, ray_normal_angle_threhold=100.0, method="reproject", filt_gradient_points=False, return_layers=True)
    torch.cuda.synchronize()
    assert np.array_equal(seen["mv_alpha"].cpu().numpy(), f["mv_alpha"][..., 0]), "view-space coverage (mv_to_pcd alpha)"
    assert np.array_equal(mask_2d.cpu().numpy(), f["mask_2d"]), "atlas coverage mask"
    got_vis, ref_vis = mask_vis.cpu().numpy()[..., 0], f["mask_2d_visiable"][..., 0]
    mism = int((got_vis != ref_vis).sum())
    print("G67 visibility: %d of %d texel-views differ" % (mism, ref_vis.size))
    assert mism <= 2
    both = got_vis & ref_vis
    ref_cols = np.zeros((6, T, T, 3), np.float32)
    ref_cols[ref_vis] = f["vis_colors"]                    # masked_select order = view-major, row-major
    lay = layers.cpu().numpy()
    assert np.abs(ref_cols[both] - lay[both]).max() < 2e-6 and np.abs(f["colors_2d"][both] - lay[both]).max() < 2e-6, "per-view gathered colours"
    # the composite's winner = the first view in priority order that sees the texel (renderer_inverse.py:44, 585-596); bit-exact where the visibility agrees on every view
    agree = (got_vis == ref_vis).all(0)
    exp = np.full((T, T), -1, np.int32)
    for v in reversed(PRIORITY):
        exp[ref_vis[v]] = v
    win = inv.last["winner"].cpu().numpy().astype(np.int32)
    assert np.array_equal(win[agree], exp[agree]), "composite winner index"
    assert (win >= 0).sum() > 0.3 * f["mask_2d"].sum() and ((win < 0) & f["mask_2d"][0, ..., 0]).sum() > 50, "the case must have seen and unseen texels"
    err = np.abs(color_2d.cpu().numpy()[0] - f["color_2d"][0])
    print("G67 final atlas: max |d| %.3g, median %.3g, share beyond 1e-4: %.5f" % (err.max(), np.median(err), (err > 1e-4).mean()))
    assert (err > 1e-4).mean() < 2e-3 and np.median(err) < 1e-6
    tex = textured.texture
    assert tex.shape == (T, T, 3) and tex.dtype == np.uint8
    exp_u8 = np.clip(np.floor(color_2d.cpu().numpy()[0] * 255.0), 0, 255).astype(np.uint8)[::-1]      # tensor_to_image truncates (utils/image.py), FLIP_TOP_BOTTOM
    assert np.abs(tex.astype(np.int32) - exp_u8.astype(np.int32)).max() <= 1
