"""GPU tests of the drop-in call surface: PBRFluxPipeline host logic against the reference fixture (with the
same stand-in transformer / VAE the reference was run with), and CustomRGBTextureFullPipeline end to end on a
synthetic mesh with a tiny DiT, its back-projection stage checked against the CPU oracle."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import geom_ref as G
from tests import fakes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,with_dual", [("tex", True), ("delight", False)])
def test_product_pipeline_orchestration_matches_reference_fixture(tag, with_dual):
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    f = np.load(os.path.join(GOLD, "g1_pipeline.npz"))
    dit = fakes.FakeDiT()
    pipe = PBRFluxPipeline(dit, fakes.FakeVAE(), device="cuda:0")
    gen = torch.Generator().manual_seed(63)
    out = pipe(prompt="[MVFLUX]", control_image=Image.fromarray(f["orch_control"]),
               dual_image=Image.fromarray(f["orch_dual"]) if with_dual else None, prompt_embeds=None,
               pooled_prompt_embeds=None, height=64, width=192, n_rows=1, n_cols=6, num_inference_steps=4,
               guidance_scale=3.5, max_sequence_length=16, generator=gen)
    torch.cuda.synchronize()
    assert len(dit.calls) == 4
    assert np.array_equal(dit.img_ids.numpy(), f["orch_%s_img_ids" % tag])
    assert dit.cond_absmax == 0.0 and dit.guidance == 3.5
    for i, (hid, t_in) in enumerate(dit.calls):
        assert np.array_equal(hid, f["orch_%s_step%d_hidden" % (tag, i)][0]), "latents at step %d" % i
        assert np.float32(t_in) == f["orch_%s_step%d_timestep" % (tag, i)][0]
    assert np.array_equal(np.asarray(out.images[0]), f["orch_%s_image" % tag])
    assert np.array_equal(torch.randn(4, generator=gen).numpy(), f["orch_%s_next_randn" % tag])


def test_full_pipeline_end_to_end_tiny(tmp_path):
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.pipeline import CustomRGBTextureFullPipeline
    from unitex_amd.texturetools import meshes
    dev = "cuda:0"
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
    # stand-in VAE (exact arithmetic): the HIP AutoencoderKL has its own parity tests (tests/test_vae_gpu.py)
    flux = PBRFluxPipeline(FluxDiT(sd, shape, device=dev), fakes.FakeVAE(), device=dev)
    flux.load_lora_weights(synthetic_lora(sd, shape, rank=16, seed=1, device=dev), adapter_name="texture")
    flux.load_lora_weights(synthetic_lora(sd, shape, rank=16, seed=2, device=dev), adapter_name="delight")
    pipe = CustomRGBTextureFullPipeline(seed=63, pipeline=flux, num_inference_steps=2, atlas_size=512, device=dev)
    verts, faces, uvs = meshes.sphere_with_faces(3000)
    mesh_path = str(tmp_path / "in.obj")
    meshes.save_obj(mesh_path, verts * 3.0 + 0.5, faces, uvs)   # un-normalised on purpose
    img_path = str(tmp_path / "ref.png")
    yy, xx = np.mgrid[0:256, 0:256]
    Image.fromarray(np.stack([xx, yy, (xx + yy) // 2], -1).astype(np.uint8)).save(img_path)
    out_dir = str(tmp_path / "out")
    png, glb = pipe(out_dir, img_path, mesh_path)
    cache = os.path.join(out_dir, "cache")
    for name in ("processed_mesh.obj", "processed_image.png", "rembg_image.png", "mv_alpha.png", "mv_ccm.png", "mv_normal.png",
                 "camera_info.pth", "mv_rgb_w_light.png", "mv_rgb.png", "wo_LTM/textured_mesh.glb", "wo_LTM/textured_mesh.mp4", "wo_LTM/visable_uv_mask.png",
                 "wo_LTM/valid_uv_mask.png", "wo_LTM/completed_uv.png", "textured_mesh.glb"):
        assert os.path.exists(os.path.join(cache, name)), name
    assert os.path.exists(png) and os.path.exists(glb) and open(glb, "rb").read(4) == b"glTF"
    assert np.asarray(Image.open(os.path.join(cache, "mv_rgb.png"))).shape == (1024, 1536, 3)
    assert np.asarray(Image.open(os.path.join(cache, "mv_normal.png"))).shape == (1024, 1536, 3)
    # ---- geometry-condition render vs oracle (coverage must be identical: same raster rule)
    pv, pf, puv, pfuv = meshes.load_obj(os.path.join(cache, "processed_mesh.obj"))
    assert abs((pv.max(0) - pv.min(0)).max() - 1.9) < 1e-5
    cam = torch.load(os.path.join(cache, "camera_info.pth"), weights_only=True)
    c2ws, intr = cam["c2ws"].numpy(), cam["intrinsics"].numpy()
    mvp = G.mvp_matrices(c2ws, intr, perspective=False)
    # the conditions are rendered from the RAW input mesh (reference pipeline.py:573), normalised by export_condition's own scale_to_bbox(0.95)
    rv_, rf_, _, _ = meshes.load_obj(mesh_path)
    rv_ = rv_.astype(np.float32)
    lo, hi = rv_.min(0), rv_.max(0)
    rv_ = ((rv_ - np.float32(0.5) * (lo + hi)) / ((hi - lo).max() / np.float32(2.0 * 0.95))).astype(np.float32)
    clip = G.transform_points(rv_, mvp)
    alpha_img = np.asarray(Image.open(os.path.join(cache, "mv_alpha.png")))
    for v in range(6):
        r, c = divmod(v, 3)
        ref = (G.rasterize(clip[v], rf_, 512, 512)[..., 3] > 0)
        got = alpha_img[r * 512:(r + 1) * 512, c * 512:(c + 1) * 512] > 0
        assert np.array_equal(got, ref), "condition alpha view %d" % v
    # ---- back-projection stage vs the CPU oracle on the same mv_rgb.png
    vv, ff, uu = meshes.unify_uv_indexing(pv, pf, puv, pfuv)
    T = 512
    img = np.asarray(Image.open(os.path.join(cache, "mv_rgb.png")).convert("RGB"), dtype=np.float32) / 255.0
    views = img.reshape(2, 512, 3, 512, 3).transpose(0, 2, 1, 3, 4).reshape(6, 512, 512, 3)
    clip = G.transform_points(vv, mvp)
    vndc = (clip[..., :2] / clip[..., 3:4]).astype(np.float32)
    alphas = np.stack([(G.rasterize(clip[v], ff, 512, 512)[..., 3] > 0).astype(np.float32) for v in range(6)])
    uvclip = np.concatenate([uu * 2 - 1, np.zeros((len(uu), 1), np.float32), np.ones((len(uu), 1), np.float32)], -1)
    rast2d = G.rasterize(uvclip, ff, T, T)
    mask2d = rast2d[..., 3] > 0
    bvh = G.BVH(vv, ff)
    col, rv, ao = G.backproject(rast2d, vv, ff, G.face_normals(vv, ff), vndc, (-c2ws[:, :3, 2]).astype(np.float32),
                                np.concatenate([views, alphas[..., None]], -1).astype(np.float32), bvh, angle_deg=100.0)
    vis = G.dilate_visibility(rv, mask2d, ao)
    valid = np.asarray(Image.open(os.path.join(cache, "wo_LTM/valid_uv_mask.png"))) > 127
    visable = np.asarray(Image.open(os.path.join(cache, "wo_LTM/visable_uv_mask.png"))) > 127
    assert np.array_equal(valid, mask2d), "UV coverage mask"
    assert np.array_equal(visable, vis.any(0)), "union visibility mask"
    atlas, seen, win, bnd = G.composite(col, vis)
    filled, _ = G.nn_fill_brute(atlas, win, rast2d, G.interpolate(vv, rast2d, ff)) if (mask2d & ~seen).sum() < 40000 else G.nn_fill(atlas, seen, mask2d, G.interpolate(vv, rast2d, ff))
    blur = G.lens_blur_collapsed(filled, G.seam_mask(bnd, mask2d))
    final = G.pull_push(blur.transpose(2, 0, 1), mask2d).transpose(1, 2, 0)
    got = np.asarray(Image.open(os.path.join(cache, "wo_LTM/completed_uv.png")).convert("RGB")).astype(np.int32)
    ref8 = (np.clip(final, 0, 1) * 255 + 0.5).astype(np.int32)
    assert (np.abs(got - ref8) > 1).mean() < 1e-3, "completed atlas vs oracle"


@pytest.mark.parametrize("n_views,view_px", [(4, 256), (8, 128)])
def test_full_pipeline_4_and_8_view_variants(tmp_path, n_views, view_px):
    """BASELINE configs[0] (four views f, r, b, l: the reference's export_condition supports them, export_nvdiffrast_video.py:931-932,
    its infer_mv does not) and configs[4] (eight views, builder-defined set): the same call surface with n_views=4 / 8 must run the
    whole chain -- condition grids of the right shape, a 1 x n control strip into the DiT, the n-view back-projection with the set's
    composite priority -- and the union visibility mask must equal the oracle's for the same cameras."""
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.pipeline import CustomRGBTextureFullPipeline
    from unitex_amd.texturetools import camera, meshes
    dev = "cuda:0"
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
    flux = PBRFluxPipeline(FluxDiT(sd, shape, device=dev), fakes.FakeVAE(), device=dev)
    flux.load_lora_weights(synthetic_lora(sd, shape, rank=16, seed=1, device=dev), adapter_name="texture")
    flux.load_lora_weights(synthetic_lora(sd, shape, rank=16, seed=2, device=dev), adapter_name="delight")
    pipe = CustomRGBTextureFullPipeline(seed=63, pipeline=flux, num_inference_steps=2, atlas_size=256, device=dev, view_size=view_px,
                                        n_views=n_views)
    verts, faces, uvs = meshes.sphere_with_faces(3000)
    mesh_path = str(tmp_path / "in.obj")
    meshes.save_obj(mesh_path, verts, faces, uvs)
    img_path = str(tmp_path / "ref.png")
    yy, xx = np.mgrid[0:256, 0:256]
    Image.fromarray(np.stack([xx, yy, (xx + yy) // 2], -1).astype(np.uint8)).save(img_path)
    out_dir = str(tmp_path / "out")
    png, glb = pipe(out_dir, img_path, mesh_path)
    cache = os.path.join(out_dir, "cache")
    lay = pipe.VIEW_LAYOUT[n_views]
    V = view_px
    assert np.asarray(Image.open(os.path.join(cache, "mv_rgb.png"))).shape == (lay["rows"] * V, lay["cols"] * V, 3)
    assert np.asarray(Image.open(os.path.join(cache, "mv_rgb_w_light.png"))).shape == (V, n_views * V, 3)
    assert open(glb, "rb").read(4) == b"glTF"
    cam = torch.load(os.path.join(cache, "camera_info.pth"), weights_only=True)
    c2ws_ref, order = camera.generate_views_c2ws(n_views, 2.8)
    assert torch.equal(cam["c2ws"], c2ws_ref) and pipe.inverse_renderer.index == list(order)
    # union visibility vs the oracle for these cameras
    pv, pf, puv, pfuv = meshes.load_obj(os.path.join(cache, "processed_mesh.obj"))
    vv, ff, uu = meshes.unify_uv_indexing(pv, pf, puv, pfuv)
    mvp = G.mvp_matrices(cam["c2ws"].numpy(), cam["intrinsics"].numpy(), perspective=False)
    clip = G.transform_points(vv, mvp)
    vndc = (clip[..., :2] / clip[..., 3:4]).astype(np.float32)
    img = np.asarray(Image.open(os.path.join(cache, "mv_rgb.png")).convert("RGB"), dtype=np.float32) / 255.0
    views = img.reshape(lay["rows"], V, lay["cols"], V, 3).transpose(0, 2, 1, 3, 4).reshape(n_views, V, V, 3)
    alphas = np.stack([(G.rasterize(clip[v], ff, V, V)[..., 3] > 0).astype(np.float32) for v in range(n_views)])
    T = 256
    uvclip = np.concatenate([uu * 2 - 1, np.zeros((len(uu), 1), np.float32), np.ones((len(uu), 1), np.float32)], -1)
    rast2d = G.rasterize(uvclip, ff, T, T)
    mask2d = rast2d[..., 3] > 0
    col, rv, ao = G.backproject(rast2d, vv, ff, G.face_normals(vv, ff), vndc, (-cam["c2ws"].numpy()[:, :3, 2]).astype(np.float32),
                                np.concatenate([views, alphas[..., None]], -1).astype(np.float32), G.BVH(vv, ff), angle_deg=100.0)
    vis = G.dilate_visibility(rv, mask2d, ao)
    visable = np.asarray(Image.open(os.path.join(cache, "wo_LTM/visable_uv_mask.png"))) > 127
    assert np.array_equal(np.asarray(Image.open(os.path.join(cache, "wo_LTM/valid_uv_mask.png"))) > 127, mask2d)
    assert np.array_equal(visable, vis.any(0)), "union visibility mask (%d views)" % n_views


def test_condition_render_uses_the_raw_input_mesh_not_the_processed_one(tmp_path):
    """/root/reference/pipeline.py:573: step_1_1 renders the geometry conditions (the DiT's control image) from `input_mesh_path`, not from
    cache/processed_mesh.obj.  A UV-less 3000-face input is subdivided by preprocess_blank_mesh (min_faces = 20 000), so the two differ: the
    alpha grid must be the oracle's coverage of the RAW mesh (bbox-normalised by export_condition itself)."""
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.pipeline import CustomRGBTextureFullPipeline
    from unitex_amd.texturetools import meshes
    dev = "cuda:0"
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    flux = PBRFluxPipeline(FluxDiT(SyntheticFluxStateDict(shape, seed=0, device=dev), shape, device=dev), fakes.FakeVAE(), device=dev)
    pipe = CustomRGBTextureFullPipeline(seed=63, pipeline=flux, num_inference_steps=2, atlas_size=256, device=dev, view_size=256)
    verts, faces, _ = meshes.sphere_with_faces(3000)
    verts = (verts * np.array([1.0, 0.6, 0.8], np.float32) + 0.25).astype(np.float32)       # an ellipsoid, off-centre
    mesh_path = str(tmp_path / "raw.obj")
    meshes.save_obj(mesh_path, verts, faces, None)
    cache = str(tmp_path / "cache"); os.makedirs(cache)
    calls = []
    real = pipe.render_geometry_images
    pipe.render_geometry_images = lambda d, p, *a, **k: (calls.append(p), real(d, p, *a, **k))[1]
    pipe.infer_mv = lambda *a, **k: None                       # the DiT is not the subject here
    pipe.preprocess_reference_image = lambda *a, **k: None
    pipe.step_1_1(cache_dir=cache, input_image_path=None, input_mesh_path=mesh_path)
    assert calls == [mesh_path]
    pv, pf, _, _ = meshes.load_obj(os.path.join(cache, "processed_mesh.obj"))
    assert len(pf) >= 20000 > len(faces)                       # the processed mesh IS a different mesh
    cam = torch.load(os.path.join(cache, "camera_info.pth"), weights_only=True)
    mvp = G.mvp_matrices(cam["c2ws"].numpy(), cam["intrinsics"].numpy(), perspective=False)
    rv_, rf_, _, _ = meshes.load_obj(mesh_path)
    rv_ = rv_.astype(np.float32)
    lo, hi = rv_.min(0), rv_.max(0)
    rv_ = ((rv_ - np.float32(0.5) * (lo + hi)) / ((hi - lo).max() / np.float32(2.0 * 0.95))).astype(np.float32)
    clip = G.transform_points(rv_, mvp)
    alpha_img = np.asarray(Image.open(os.path.join(cache, "mv_alpha.png")))
    for v in range(6):
        r, c = divmod(v, 3)
        ref = G.rasterize(clip[v], rf_, 256, 256)[..., 3] > 0
        assert np.array_equal(alpha_img[r * 256:(r + 1) * 256, c * 256:(c + 1) * 256] > 0, ref), "condition alpha of the raw mesh, view %d" % v


def test_speedup_mode_fp8_builds_an_mx_fp8_transformer_and_runs(tmp_path):
    """The reference's constructor takes `speedup_mode` and never reads it (pipeline.py:81,142-145); here "fp8" selects BASELINE configs[4]'s numerics:
    build_pipeline(speedup_mode="fp8") -> FluxDiT(fp8_weights=True) with both adapters loaded; a 2-step texture + delight pass runs end to end and
    differs from the bf16 pipeline by fp8 rounding only (same seed, same control image)."""
    from unitex_amd.flux.transformer import FluxShape
    from unitex_amd.pipeline import build_pipeline
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    yy, xx = np.mgrid[0:64, 0:192]
    ctrl = Image.fromarray(np.stack([xx % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8))
    outs = {}
    for mode in (None, "fp8"):
        pipe, wt, wd, names = build_pipeline(None, device="cuda:0", lora_rank=16, shape=shape, speedup_mode=mode)
        assert pipe.transformer.fp8_weights == (mode == "fp8")
        pipe.vae = fakes.FakeVAE()
        pipe.set_adapters(names, wt)
        img = pipe(prompt="[MVFLUX]", control_image=ctrl, height=64, width=192, num_inference_steps=2, guidance_scale=3.5, max_sequence_length=64,
                   generator=torch.Generator().manual_seed(63)).images[0]
        outs[mode] = np.asarray(img).astype(np.int32)
        del pipe
        torch.cuda.empty_cache()
    d = np.abs(outs["fp8"] - outs[None])
    assert outs[None].std() > 1.0 and d.max() > 0 and d.mean() < 8.0, "fp8 pipeline vs bf16 pipeline: mean |d| %.2f LSB, max %d" % (d.mean(), d.max())


def test_add_lora_adapters_join_both_passes(tmp_path):
    """Call surface, reference pipeline.py:112-117,142-145: `add_lora_path` / `add_lora_weights` load further adapters `add_lora_<i>` and switch them on with their
    weight in BOTH passes (weights_for_texture = [1, 0, w...], weights_for_delight = [0, 1, w...]).  Here: one extra adapter from a diffusers-spelled
    safetensors file; the transformer then runs with two switched-on adapters (rank-concatenated LoRA segment) and gives exactly what FluxDiT.set_lora gives
    for the same pair, and something else than without the extra adapter; a weight list of the wrong length is refused."""
    from safetensors.torch import save_file
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from unitex_amd.flux.transformer import FluxShape
    from unitex_amd.pipeline import build_pipeline
    dev = "cuda:0"
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
    extra = synthetic_lora(sd, shape, rank=8, seed=7, device=dev)
    flat = {}
    for mod, (A, B) in extra.items():
        if not isinstance(mod, str) or not hasattr(A, "shape"):
            continue
        flat["transformer.%s.lora_A.weight" % mod] = A.float().cpu().contiguous()
        flat["transformer.%s.lora_B.weight" % mod] = B.float().cpu().contiguous()
    path = str(tmp_path / "extra.safetensors")
    save_file(flat, path)
    with pytest.raises(ValueError):
        build_pipeline(None, device=dev, lora_rank=16, shape=shape, add_lora_path=[path], add_lora_weights=[])
    pipe, wt, wd, names = build_pipeline(None, device=dev, lora_rank=16, shape=shape, add_lora_path=[path], add_lora_weights=[0.5])
    assert names == ["texture", "delight", "add_lora_0"] and wt == [1.0, 0.0, 0.5] and wd == [0.0, 1.0, 0.5]
    pipe.vae = fakes.FakeVAE()
    yy, xx = np.mgrid[0:64, 0:192]
    ctrl = Image.fromarray(np.stack([xx % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8))

    def run(weights):
        pipe.set_adapters(names, weights)
        return np.asarray(pipe(prompt="[MVFLUX]", control_image=ctrl, height=64, width=192, num_inference_steps=2, guidance_scale=3.5, max_sequence_length=64,
                               generator=torch.Generator().manual_seed(63)).images[0]).astype(np.int32)
    with_extra = run(wt)
    active = pipe.transformer._lora_active
    assert len(active) == 2 and [s_ for _, s_ in active] == [1.0, 0.5], "texture + the extra adapter are switched on, the zero-weighted delight adapter is dropped"
    without = run([1.0, 0.0, 0.0])
    assert len(pipe.transformer._lora_active) == 1
    assert np.abs(with_extra - without).max() > 0, "the extra adapter must change the texture pass"
    delight = run(wd)
    assert [s_ for _, s_ in pipe.transformer._lora_active] == [1.0, 0.5] and np.abs(delight - with_extra).max() > 0
    # the same pair set directly on the transformer: identical image
    pipe.transformer.set_lora([(pipe._adapters["texture"], 1.0), (pipe._adapters["add_lora_0"], 0.5)])
    direct = np.asarray(pipe(prompt="[MVFLUX]", control_image=ctrl, height=64, width=192, num_inference_steps=2, guidance_scale=3.5, max_sequence_length=64,
                             generator=torch.Generator().manual_seed(63)).images[0]).astype(np.int32)
    assert np.array_equal(direct, with_extra)


def test_callback_on_step_end_and_refused_arguments():
    """flux_piplines/texturing/pipeline.py:664-671: `callback_on_step_end(pipe, i, t, {"latents": ...})` runs after every scheduler step and a returned
    "latents" replaces the running latents; arguments this loop cannot honour (custom timesteps -- the reference's own call raises for them --,
    several images per prompt, the dead redux path) are refused instead of ignored."""
    from unitex_amd.flux.transformer import FluxShape
    from unitex_amd.pipeline import build_pipeline
    shape = FluxShape(num_heads=2, num_double=1, num_single=1, joint_dim=64, pooled_dim=64)
    pipe, wt, wd, names = build_pipeline(None, device="cuda:0", lora_rank=16, shape=shape)
    pipe.vae = fakes.FakeVAE()
    pipe.set_adapters(names, wt)
    yy, xx = np.mgrid[0:64, 0:192]
    ctrl = Image.fromarray(np.stack([xx % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8))
    kw = dict(prompt="[MVFLUX]", control_image=ctrl, height=64, width=192, num_inference_steps=3, guidance_scale=3.5, max_sequence_length=64)
    seen = []

    def watch(p, i, t, tensors):
        assert p is pipe and set(tensors) == {"latents"}
        lat = tensors["latents"]
        assert lat.dim() == 3 and lat.shape[0] == 1 and lat.shape[2] == 64 and lat.shape[1] == 2 * (64 // 16) * (192 // 16)   # noise + control tokens
        seen.append((i, float(t), lat.float().abs().mean().item()))
        return {}
    base = np.asarray(pipe(generator=torch.Generator().manual_seed(63), callback_on_step_end=watch, **kw).images[0]).astype(np.int32)
    assert [s_[0] for s_ in seen] == [0, 1, 2] and seen[0][1] > seen[1][1] > seen[2][1] > 0, seen
    plain = np.asarray(pipe(generator=torch.Generator().manual_seed(63), **kw).images[0]).astype(np.int32)
    assert np.array_equal(base, plain), "an observing callback must not change the result"

    def halve(p, i, t, tensors):
        return {"latents": tensors["latents"] * 0.5} if i == 1 else {}
    changed = np.asarray(pipe(generator=torch.Generator().manual_seed(63), callback_on_step_end=halve, **kw).images[0]).astype(np.int32)
    assert np.abs(changed - plain).max() > 0, "returned latents replace the running ones"
    with pytest.raises(ValueError):
        pipe(generator=torch.Generator().manual_seed(63), timesteps=[900, 500, 100], **kw)
    with pytest.raises(NotImplementedError):
        pipe(generator=torch.Generator().manual_seed(63), num_images_per_prompt=2, **kw)
    with pytest.raises(ValueError):
        pipe(generator=torch.Generator().manual_seed(63), callback_on_step_end=watch, callback_on_step_end_tensor_inputs=["noise_pred"], **kw)


def test_build_pipeline_from_a_checkpoint_directory(tmp_path):
    """The production entry: `pretrain_models` names a directory laid out as the reference expects (pipeline.py:83-86,96-109) --
    black-forest-labs/FLUX.1-dev/{transformer,vae}/*.safetensors in diffusers key names (sharded), UniTex/{texture_gen,delight}/pytorch_lora_weights.safetensors
    in diffusers' LoRA spelling.  A tiny-shaped tree written here must load into the same pipeline as the one assembled from the tensors directly:
    identical images for the texture and the delight pass, real HIP VAE included."""
    from safetensors.torch import save_file
    from oracle import dit_ref
    from unitex_amd.flux.pipeline import PBRFluxPipeline
    from unitex_amd.flux.synthetic import synthetic_vae_state_dict
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.flux.vae_hip import AutoencoderKL
    from unitex_amd.pipeline import build_pipeline
    dev = "cuda:0"
    cfg = dit_ref.tiny_config(heads=2, double=1, single=2, joint_dim=64, pooled_dim=64)
    shape = FluxShape(num_heads=2, num_double=1, num_single=2, joint_dim=64, pooled_dim=64)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in dit_ref.make_synthetic_state_dict(cfg, seed=0).items()}
    vae_sd = {k: v.to(torch.bfloat16).contiguous() for k, v in synthetic_vae_state_dict(0).items()}
    tex = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=1)
    dlt = dit_ref.make_synthetic_lora(cfg, sd, rank=16, seed=2)
    root = tmp_path / "pretrain"
    tdir = root / "black-forest-labs" / "FLUX.1-dev" / "transformer"
    vdir = root / "black-forest-labs" / "FLUX.1-dev" / "vae"
    for d in (tdir, vdir, root / "UniTex" / "texture_gen", root / "UniTex" / "delight"):
        os.makedirs(d)
    keys = sorted(sd)
    save_file({k: sd[k] for k in keys[: len(keys) // 2]}, str(tdir / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[len(keys) // 2:]}, str(tdir / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    save_file(vae_sd, str(vdir / "diffusion_pytorch_model.safetensors"))
    for name, lo in (("texture_gen", tex), ("delight", dlt)):
        flat = {}
        for mod, (A, B) in lo.items():
            flat["transformer.%s.lora_A.weight" % mod] = A.float().contiguous()
            flat["transformer.%s.lora_B.weight" % mod] = B.float().contiguous()
        save_file(flat, str(root / "UniTex" / name / "pytorch_lora_weights.safetensors"))
    pipe, wt, wd, names = build_pipeline(str(root), device=dev, shape=shape)
    assert isinstance(pipe.vae, AutoencoderKL) and names == ["texture", "delight"]
    direct = PBRFluxPipeline(FluxDiT(sd, shape, device=dev), AutoencoderKL(vae_sd, device=dev), device=dev)
    direct.load_lora_weights(tex, adapter_name="texture")
    direct.load_lora_weights(dlt, adapter_name="delight")
    yy, xx = np.mgrid[0:64, 0:192]
    ctrl = Image.fromarray(np.stack([xx % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8))
    kw = dict(prompt="[MVFLUX]", control_image=ctrl, height=64, width=192, num_inference_steps=2, guidance_scale=3.5, max_sequence_length=64)
    for weights in (wt, wd):
        imgs = []
        for p in (pipe, direct):
            p.set_adapters(names, weights)
            imgs.append(np.asarray(p(generator=torch.Generator().manual_seed(63), **kw).images[0]))
        assert imgs[0].shape == (64, 192, 3) and imgs[0].std() > 1.0
        assert np.array_equal(imgs[0], imgs[1]), "pipeline loaded from the checkpoint tree differs from the one built from the same tensors"
    with pytest.raises(FileNotFoundError):
        os.remove(str(vdir / "diffusion_pytorch_model.safetensors"))
        build_pipeline(str(root), device=dev, shape=shape)
