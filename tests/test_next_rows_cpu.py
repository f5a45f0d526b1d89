"""CPU tests of the 'next' rows around the hot path: mesh input (GLB / UV-less meshes), the
builder-defined UV atlas, the orbit cameras of the turntable video (pinned by the reference fixture G4) and the
Motion-JPEG MP4 muxer."""
import io
import os

import numpy as np
import torch
from PIL import Image

from oracle import geom_ref as G
from unitex_amd.texturetools import camera, meshes, video

HERE = os.path.dirname(os.path.abspath(__file__))


def test_orbit_cameras_match_reference_fixture():
    f = np.load(os.path.join(HERE, "golden", "g4_cameras.npz"))
    a = camera.generate_orbit_views_c2ws(121, radius=2.8, height=0.0, theta_0=0.0, degree=True)[:120].numpy()
    assert np.array_equal(a, f["orbit_c2ws"])
    b = camera.generate_orbit_views_c2ws(9, radius=2.8, height=1.4, theta_0=30.0, degree=True).numpy()
    assert np.array_equal(b, f["orbit_c2ws_pitch"])
    # straight-down view takes the hard-coded x axis (generator.py:28-31)
    c = camera.lookat_to_matrix(torch.tensor([[0.0, 0.0, 2.8]])).numpy()[0]
    assert np.isfinite(c).all() and abs(np.linalg.det(c[:3, :3])) > 0.99


def test_glb_round_trip(tmp_path):
    v, f, uv = meshes.sphere_with_faces(1200)
    tex = (np.random.default_rng(0).random((32, 48, 3)) * 255).astype(np.uint8)
    p = str(tmp_path / "m.glb")
    meshes.save_glb(p, v, f, uv, tex)
    v2, f2, uv2, t2 = meshes.load_mesh(p)
    assert np.array_equal(v, v2) and np.array_equal(f, f2) and np.array_equal(tex, t2)
    assert np.allclose(uv, uv2, atol=1e-6)


def test_unwrap_grid_is_a_valid_atlas():
    v, f, _ = meshes.sphere_with_faces(1500)
    T = 512
    vv, fp, uu, ff = meshes.unwrap_grid(v, f, atlas=T, gutter=2.0)
    assert ff.shape == f.shape and uu.min() > 0 and uu.max() < 1
    assert np.array_equal(vv, v) and np.array_equal(fp, f)      # positions stay shared (smooth condition normals), UVs per corner
    # rasterise the atlas with the oracle rasteriser: every face owns texels, and a 1-texel dilation of any face
    # never touches another face (gutter), so bilinear fetches / dilation cannot bleed between triangles
    uvclip = np.concatenate([uu * 2 - 1, np.zeros((len(uu), 1), np.float32), np.ones((len(uu), 1), np.float32)], -1)
    ids = G.rasterize(uvclip, ff, T, T)[..., 3].astype(np.int64)
    owned = np.bincount(ids.reshape(-1), minlength=len(ff) + 1)[1:]
    assert (owned > 0).all(), "%d faces have no texel" % int((owned == 0).sum())
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = ids[max(dy, 0):, max(dx, 0):][:T - abs(dy), :T - abs(dx)]
        b = ids[:T - dy, max(-dx, 0):][:T - abs(dy), :T - abs(dx)]
        both = (a > 0) & (b > 0)
        assert (a[both] == b[both]).all(), "adjacent texels belong to different faces"


def test_prepare_blank_mesh_without_uvs(tmp_path):
    v, f, _ = meshes.sphere_with_faces(900)
    v = np.concatenate([v, v[:5] + 1e-12])                       # duplicate vertices, an unreferenced one
    f = np.concatenate([f, [[0, 0, 1]]]).astype(np.int32)        # a degenerate face
    p = str(tmp_path / "blank.obj")
    meshes.save_obj(p, v, f)
    vv, ff, uu, fu = meshes.prepare_blank_mesh(p, min_faces=3000, max_faces=20000, scale=0.95, atlas=1024, gutter=2.0)
    assert 3000 <= len(ff) <= 20000 and len(uu) == 3 * len(ff) and fu.shape == ff.shape and len(vv) < len(ff)   # shared positions
    # round trip through the OBJ the pipeline writes: the condition render sees shared positions, the inverse renderer one vertex per (v, vt)
    q = str(tmp_path / "processed_mesh.obj")
    meshes.save_obj(q, vv, ff, uu, faces_uv=fu)
    v1, f1, t1, ft1 = meshes.load_obj(q)
    assert len(v1) == len(vv) and np.array_equal(f1, ff) and np.array_equal(ft1, fu)
    v2, f2, t2, _ = meshes.load_mesh(q)
    assert len(v2) == len(t2) == 3 * len(ff) and np.allclose(v2[f2.reshape(-1)], vv[ff.reshape(-1)], atol=1e-6)
    # the reference rescales to the bbox FIRST and subdivides afterwards (uv_atlas.py:139-147, then :164): two Loop iterations pull the surface
    # inside the control mesh, so the processed mesh is slightly smaller than 2 * scale -- as the reference's is
    assert len(ff) == 16 * 896 and 1.8 < (vv.max(0) - vv.min(0)).max() <= 1.9 + 1e-5
    big_v, big_f, _ = meshes.sphere_with_faces(30000)
    dv, df = meshes.decimate_cluster(*meshes.clean_mesh(big_v, big_f), max_faces=8000)
    assert 500 < len(df) <= 8000 and df.max() < len(dv)


def test_mjpeg_mp4_round_trip(tmp_path):
    yy, xx = np.mgrid[0:64, 0:96]
    frames = [np.stack([(xx * 2 + 10 * i) % 256, yy * 3 % 256, (xx + yy) % 256], -1).astype(np.uint8) for i in range(7)]
    p = str(tmp_path / "t.mp4")
    video.write_mjpeg_mp4(p, frames, fps=15)
    blob = open(p, "rb").read()
    assert blob[4:8] == b"ftyp" and b"moov" in blob and b"jpeg" in blob
    fps, jpgs = video.read_mjpeg_mp4(p)
    assert fps == 15 and len(jpgs) == 7
    for j, fr in zip(jpgs, frames):
        im = np.asarray(Image.open(io.BytesIO(j)).convert("RGB")).astype(np.int32)
        assert im.shape == fr.shape and np.abs(im - fr).mean() < 12.0     # JPEG, smooth content
    video.write_gif(str(tmp_path / "t.gif"), frames, fps=15)
    assert Image.open(str(tmp_path / "t.gif")).n_frames == 7


def test_lora_safetensors_reader_accepts_both_spellings(tmp_path):
    """checkpoint boundary of the drop-in (reference pipeline.py:96-112 loads diffusers / peft LoRA files): module names
    relative to the transformer, peft alpha / r scaling folded into B, '.lora_A/.lora_B' and '.lora.down/.lora.up'."""
    from safetensors.torch import save_file
    from unitex_amd.flux.lora_io import load_lora_safetensors
    g = torch.Generator().manual_seed(0)
    A = torch.randn(16, 64, generator=g); B = torch.randn(96, 16, generator=g)
    A2 = torch.randn(8, 64, generator=g); B2 = torch.randn(64, 8, generator=g)
    p1 = str(tmp_path / "a.safetensors")
    save_file({"transformer.single_transformer_blocks.0.attn.to_q.lora_A.weight": A,
               "transformer.single_transformer_blocks.0.attn.to_q.lora_B.weight": B,
               "transformer.single_transformer_blocks.0.attn.to_q.alpha": torch.tensor(8.0),
               "transformer_blocks.1.ff.net.2.lora.down.weight": A2,
               "transformer_blocks.1.ff.net.2.lora.up.weight": B2}, p1)
    d = load_lora_safetensors(p1)
    assert set(d) == {"single_transformer_blocks.0.attn.to_q", "transformer_blocks.1.ff.net.2"}
    a, b = d["single_transformer_blocks.0.attn.to_q"]
    assert torch.equal(a, A) and torch.allclose(b, B * (8.0 / 16.0))
    a2, b2 = d["transformer_blocks.1.ff.net.2"]
    assert torch.equal(a2, A2) and torch.equal(b2, B2)
    p2 = str(tmp_path / "bad.safetensors")
    save_file({"transformer.x.lora_A.weight": A}, p2)
    try:
        load_lora_safetensors(p2)
        assert False, "missing lora_B must raise"
    except KeyError:
        pass


def test_lora_safetensors_reader_returns_module_copies_and_rejects_unknown_keys(tmp_path):
    """the trainer saves whole-module copies with every adapter (peft modules_to_save: x_embedder, trainer.py:297-304, written
    through get_peft_model_state_dict, trainer.py:480-490): they must come back under FULL_KEY in both spellings, and a key that
    is neither a LoRA factor, an alpha nor a module copy must raise instead of vanishing."""
    from safetensors.torch import save_file
    from unitex_amd.flux.lora_io import FULL_KEY, load_lora_safetensors
    g = torch.Generator().manual_seed(1)
    A = torch.randn(4, 64, generator=g); B = torch.randn(32, 4, generator=g)
    Wx = torch.randn(32, 64, generator=g); bx = torch.randn(32, generator=g)
    base = {"transformer.transformer_blocks.0.attn.to_out.0.lora_A.weight": A, "transformer.transformer_blocks.0.attn.to_out.0.lora_B.weight": B}
    for spelling in ("transformer.x_embedder.%s", "transformer.x_embedder.modules_to_save.default.%s", "x_embedder.modules_to_save.%s"):
        p = str(tmp_path / "m.safetensors")
        save_file(dict(base, **{spelling % "weight": Wx, spelling % "bias": bx}), p)
        d = load_lora_safetensors(p)
        assert set(d) == {"transformer_blocks.0.attn.to_out.0", FULL_KEY}
        assert set(d[FULL_KEY]) == {"x_embedder.weight", "x_embedder.bias"}
        assert torch.equal(d[FULL_KEY]["x_embedder.weight"], Wx) and torch.equal(d[FULL_KEY]["x_embedder.bias"], bx)
    p = str(tmp_path / "u.safetensors")
    save_file(dict(base, **{"transformer.x_embedder.magnitude": bx}), p)
    try:
        load_lora_safetensors(p)
        assert False, "an unrecognised tensor must raise"
    except ValueError as e:
        assert "magnitude" in str(e)


def test_vae_parameter_table_matches_oracle_module():
    """the product's diffusers-keyed VAE parameter table (flux/synthetic.py) is exactly the oracle module's state dict."""
    from oracle import vae_ref
    from unitex_amd.flux.synthetic import vae_param_shapes
    ref = {k: tuple(v.shape) for k, v in vae_ref.AutoencoderKL().state_dict().items()}
    assert ref == {k: tuple(v) for k, v in vae_param_shapes().items()}


def test_bench_workload_arithmetic_matches_baseline_md():
    """bench.py's token counts / algorithmic FLOPs are the ones BASELINE.md section 2 tabulates."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(HERE), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name, S_ref, tflop_ref, attn_ref in (("strip1024x6", 50688, 2454.0, 1800.0), ("ref512x6", 13824, 312.3, 133.9),
                                             ("view1024", 9728, 191.9, 66.3), ("view2048", 34304, 1267.0, None)):
        S = sum(bench.token_counts(name))
        assert S == S_ref
        fl, fa = bench.step_flops(S)
        assert abs(fl / 1e12 - tflop_ref) / tflop_ref < 2e-3
        if attn_ref:
            assert abs(fa / 1e12 - attn_ref) / attn_ref < 5e-3


def test_bench_counts_the_gpus_under_a_job_without_trusting_a_placeholder_uuid():
    """bench.py --gpus N refuses ranks that share a GPU; the count must not turn a driver that reports the same (all-zero) uuid for every device
    into a refusal of a valid 8-GPU job, and must still see two ranks on one device."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(HERE), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    zero = ("00000000-0000-0000-0000-000000000000", 0, 0, 0)
    same_env = ("", "", "")
    # torch.distributed.run: one device list, rank r on index r -- distinct whatever the driver says about uuids
    ids = [{"visible": same_env, "index": r, "hw": zero} for r in range(8)]
    assert bench.count_distinct_devices(ids)[0] == 8
    # fewer devices than ranks (indices wrapped): shared
    ids = [{"visible": same_env, "index": r % 4, "hw": zero} for r in range(8)]
    assert bench.count_distinct_devices(ids)[0] == 4
    # one device per process through HIP_VISIBLE_DEVICES: told apart by the hardware identity ...
    ids = [{"visible": (str(r), "", ""), "index": 0, "hw": ("GPU-%02x" % r, 0, 0x10 + r, 0)} for r in range(4)]
    assert bench.count_distinct_devices(ids) == (4, "uuid + PCI address")
    # ... two of them on the same physical GPU are seen ...
    ids[3]["hw"] = ids[2]["hw"]
    assert bench.count_distinct_devices(ids)[0] == 3
    # ... and a placeholder identity under different device lists is "unknown", not a refusal
    ids = [{"visible": (str(r), "", ""), "index": 0, "hw": zero} for r in range(4)]
    assert bench.count_distinct_devices(ids)[0] is None
    ids[1]["hw"] = None
    assert bench.count_distinct_devices(ids)[0] is None
    rec = bench.device_identity(0)      # no GPU here: the record still forms, hardware identity unknown
    assert rec["index"] == 0 and len(rec["visible"]) == 3


def test_reproject_and_query_field_forwards_variant_switches(tmp_path):
    """host logic of RGBTextureFullPipelineBase.reproject_and_query_field (reference pipeline.py:313-347): the 2 x 3 grid is
    cut into six view images in (row, col) order, and `method` / `inpainting` reach the inverse renderer as method,
    kdtree_inpainting, reproject_inpainting and filt_gradient_points with the reference's thresholds (0.15, 100 degrees)."""
    import types

    import numpy as np
    import torch
    from PIL import Image

    from unitex_amd.pipeline import RGBTextureFullPipelineBase
    HP = WP = 8
    grid = np.zeros((2 * HP, 3 * WP, 3), np.uint8)
    for r in range(2):
        for c in range(3):
            grid[r * HP:(r + 1) * HP, c * WP:(c + 1) * WP] = 40 * (3 * r + c) + 10
    Image.fromarray(grid).save(tmp_path / "mv.png")
    torch.save({"c2ws": torch.eye(4)[None].repeat(6, 1, 1), "intrinsics": torch.eye(3), "perspective": False}, tmp_path / "cam.pth")
    seen = {}

    class FakeTextured:
        def export(self, path):
            seen["export"] = path

    class FakeInverse:
        def update_from_file(self, p):
            seen["mesh"] = p

        def infer(self, mesh, **kw):
            seen["kw"] = kw
            T = kw["H2D"]
            return FakeTextured(), torch.zeros(6, T, T, 1, dtype=torch.bool), torch.ones(1, T, T, 1, dtype=torch.bool), torch.zeros(1, T, T, 3)

        def clear(self):
            seen["cleared"] = True

    fake = types.SimpleNamespace(inverse_renderer=FakeInverse(), atlas_size=16)
    fn = RGBTextureFullPipelineBase.reproject_and_query_field
    fn = getattr(fn, "__wrapped__", fn)
    for method, inp in (("reproject", False), ("kdtree", True)):
        seen.clear()
        fn(fake, str(tmp_path), "mesh.obj", str(tmp_path / "mv.png"), str(tmp_path / "cam.pth"), method=method, inpainting=inp)
        kw = seen["kw"]
        assert kw["method"] == method and kw["kdtree_inpainting"] is inp and kw["reproject_inpainting"] is inp
        assert kw["filt_gradient_points"] is inp and kw["grad_norm_threhold"] == 0.15 and kw["ray_normal_angle_threhold"] == 100
        assert kw["H"] == HP and kw["W"] == WP and kw["H2D"] == 16 and kw["perspective"] is False
        ia = kw["image_attrs"]
        assert tuple(ia.shape) == (6, HP, WP, 3)
        for v in range(6):
            assert torch.allclose(ia[v], torch.full((HP, WP, 3), (40 * v + 10) / 255.0)), "view %d of the 2 x 3 grid" % v
        assert seen["export"].endswith("textured_mesh.glb") and seen["cleared"]
        for name in ("visable_uv_mask.png", "valid_uv_mask.png", "completed_uv.png"):
            assert (tmp_path / name).exists()


def test_vertex_normal_weightings_area_and_angle():
    """VideoExporter's production default is trimesh's angle weighting (`video._vertex_normals`), the reference's own
    fallback is area weighting (mesh/structure.py:522-548, pinned by G9).  (i) On a finely tessellated unit sphere both must be
    the radial direction (1 - cos < 1e-4, outward); (ii) on a regular fan both give the axis exactly; (iii) on a skewed fan --
    one long thin triangle of large area but small corner angle -- they must differ, and the angle-weighted normal must equal
    an independent per-corner evaluation of sum(angle_i * n_i)."""
    from unitex_amd.texturetools.video import _vertex_normals
    v, f = meshes.closed_sphere(128, 64)
    for w in ("area", "angle"):
        n = _vertex_normals(torch.from_numpy(v), torch.from_numpy(f), weighting=w).double().numpy()
        d = (n * v.astype(np.float64)).sum(-1)
        assert d.min() > 1.0 - 1e-4, (w, d.min())
    # regular hexagonal fan around the apex of a cone: symmetric -> the axis for both weightings
    ring = [[np.cos(t), np.sin(t), -0.5] for t in np.arange(6) * np.pi / 3]
    fv = np.asarray([[0.0, 0.0, 0.0]] + ring, np.float32)
    ff = np.asarray([[0, 1 + i, 1 + (i + 1) % 6] for i in range(6)], np.int32)
    for w in ("area", "angle"):
        n0 = _vertex_normals(torch.from_numpy(fv), torch.from_numpy(ff), weighting=w)[0].numpy()
        assert np.allclose(n0, [0, 0, 1], atol=1e-6), (w, n0)
    # skewed fan: two triangles share vertex 0; the second is long and thin (area 10x, corner angle at vertex 0 ~ 0.1 rad)
    sv = np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0],          # triangle A in the z = 0 plane: normal +z, angle pi/2, area 0.5
                     [0, 50.0, 0.0], [0, 50.0, -5.0]], np.float32)   # triangle B in the x = 0 plane: normal -x... see below
    sf = np.asarray([[0, 1, 2], [0, 3, 4]], np.int32)
    na = _vertex_normals(torch.from_numpy(sv), torch.from_numpy(sf), weighting="area")[0].double().numpy()
    ng = _vertex_normals(torch.from_numpy(sv), torch.from_numpy(sf), weighting="angle")[0].double().numpy()
    exp = np.zeros(3)
    for tri in sf:       # independent evaluation, corner of vertex 0 only
        p0, p1, p2 = (sv[i].astype(np.float64) for i in tri)
        fn = np.cross(p1 - p0, p2 - p0); fn /= np.linalg.norm(fn)
        e1, e2 = (p1 - p0) / np.linalg.norm(p1 - p0), (p2 - p0) / np.linalg.norm(p2 - p0)
        exp += np.arccos(np.clip(e1 @ e2, -1, 1)) * fn
    exp /= np.linalg.norm(exp)
    assert np.allclose(ng, exp, atol=1e-6)
    assert np.linalg.norm(na - ng) > 0.5, "the skewed fan must separate the two weightings (area %s, angle %s)" % (na, ng)


import pytest


@pytest.mark.parametrize("n", [4, 6, 8])
def test_view_grid_strip_round_trip_for_4_6_8_views(n, tmp_path):
    """infer_mv's host permutations generalised from the reference's six views (pinned by fixture G3) to the 4-view set of
    BASELINE configs[0] (export_nvdiffrast_video.py:931-932) and the builder-defined 8-view set of configs[4]: with an echo
    pipeline, a grid of tagged tiles must come back as the same grid (strip order and the 180-degree turn of the 'down' view undone)
    and the strip the DiT sees must hold the views in the order front, left, right, back, top, down (, diagonals)."""
    import types
    from PIL import Image as PI
    from unitex_amd.pipeline import RGBTextureFullPipelineBase as B
    lay = B.VIEW_LAYOUT[n]
    V, R, Cc = 32, lay["rows"], lay["cols"]
    rng = np.random.default_rng(n)
    grid = np.zeros((R * V, Cc * V, 3), np.uint8)
    for t in range(n):
        tile = rng.integers(0, 255, (V, V, 3), dtype=np.uint8) // 2 * 2        # even values: 0.5 x + 0.5 x is exact
        tile[..., 2] = 2 * t                                                    # tag
        grid[(t // Cc) * V:(t // Cc + 1) * V, (t % Cc) * V:(t % Cc + 1) * V] = tile
    PI.fromarray(grid).save(tmp_path / "n.png"); PI.fromarray(grid).save(tmp_path / "c.png")
    PI.fromarray(np.zeros((V, V, 3), np.uint8)).save(tmp_path / "ref.png")
    seen = {}

    class Echo:
        _num_inference_steps = 2

        def set_adapters(s, adapter_names, adapter_weights):
            pass

        def __call__(s, **kw):
            seen.setdefault("strips", []).append(np.asarray(kw["control_image"]).copy())
            assert (kw["height"], kw["width"], kw["n_cols"]) == (V, n * V, n)
            return types.SimpleNamespace(images=[kw["control_image"]])
    me = types.SimpleNamespace(pipeline=Echo(), pipeline_name="texture_plus", view_size=V, n_views=n, generator=None,
                               adapter_names=["texture", "delight"], weights_for_texture=[1.0, 0.0], weights_for_delight=[0.0, 1.0])
    B.infer_mv(me, str(tmp_path), str(tmp_path / "ref.png"), str(tmp_path / "n.png"), str(tmp_path / "c.png"))
    out = np.asarray(PI.open(tmp_path / "mv_rgb.png"))
    assert np.array_equal(out, grid), "grid -> strip -> grid must be the identity for %d views" % n
    order = [int(seen["strips"][0][V // 2, i * V + V // 2, 2]) // 2 for i in range(n)]
    assert order == lay["strip"]
    names = {4: "f r b l", 6: "f r t b l d", 8: "f r t b l d x1 x2"}[n].split()
    assert [names[g] for g in order][:4] == ["f", "l", "r", "b"]


def test_obj_reader_fast_path_equals_the_line_reader_and_falls_back(tmp_path):
    """load_obj parses the all-triangle files this package writes with whole-file numpy conversions (the pipeline re-reads processed_mesh.obj in
    every stage, io/mesh_loader.py in the reference); every other spelling must go through -- and equal -- the line-by-line reader."""
    v, f, uv = meshes.make_bumpy_sphere(24, 12)[:3]
    v, f, uv = np.asarray(v, dtype=np.float32), np.asarray(f, dtype=np.int32), np.asarray(uv, dtype=np.float32)

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x is None and y is None) or (x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y))
    p = str(tmp_path / "a.obj")
    meshes.save_obj(p, v, f, uv, f)                    # v / vt / f a/b a/b a/b
    got = meshes.load_obj(p)
    same(got, meshes._load_obj_generic(p))
    assert got[0].shape == v.shape and np.array_equal(got[1], f) and np.array_equal(got[3], f) and np.allclose(got[2], uv, atol=1e-6)
    meshes.save_obj(p, v, f)                           # no uvs: f a b c
    same(meshes.load_obj(p), meshes._load_obj_generic(p))
    assert meshes.load_obj(p)[2] is None
    body = "".join("v %r %r %r\n" % tuple(float(c) for c in r) for r in v) + "".join("vt %r %r\n" % tuple(float(c) for c in r) for r in uv)
    variants = {
        "normals": body + "vn 0 0 1\n" + "".join("f %d/%d/1 %d/%d/1 %d/%d/1\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in f),
        "quad": body + "f 1/1 2/2 3/3 4/4\n" + "".join("f %d/%d %d/%d %d/%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in f),
        "negative": body + "".join("f %d/%d %d/%d %d/%d\n" % (a - len(v), a - len(uv), b - len(v), b - len(uv), c - len(v), c - len(uv)) for a, b, c in f),
        "no_vt_ref": body + "vn 0 0 1\n" + "".join("f %d//1 %d//1 %d//1\n" % (a + 1, b + 1, c + 1) for a, b, c in f),
        "vertex_colours": "".join("v %r %r %r 1 0 0\n" % tuple(float(c) for c in r) for r in v) + "".join("f %d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in f),
        "comments_and_groups": "# c\no x\n" + body + "g part\ns off\n" + "".join("f %d/%d %d/%d %d/%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in f),
    }
    for name, text in variants.items():
        q = str(tmp_path / (name + ".obj"))
        with open(q, "w") as fh:
            fh.write(text)
        same(meshes.load_obj(q), meshes._load_obj_generic(q))
    assert np.array_equal(meshes.load_obj(str(tmp_path / "negative.obj"))[1], f)
    assert len(meshes.load_obj(str(tmp_path / "quad.obj"))[1]) == len(f) + 2


def test_stage_encoders_finish_inside_the_stage_and_surface_errors(tmp_path):
    """texturetools/timer.py::Encoders: the codecs of one stage run side by side on host threads, but the stage's files are on disk when its
    `with` block ends (the stages hand over PATHS, reference pipeline.py:594-632) and an encoder's exception is raised in the stage that owns it."""
    from unitex_amd.texturetools.timer import Encoders
    imgs = [Image.fromarray(np.full((64, 64, 3), 10 * i, np.uint8)) for i in range(6)]
    with Encoders() as enc:
        for i, im in enumerate(imgs):
            enc.submit(im.save, str(tmp_path / ("a%d.png" % i)), compress_level=1)
    for i in range(6):
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / ("a%d.png" % i)))), np.asarray(imgs[i]))
    with pytest.raises(FileNotFoundError):
        with Encoders() as enc:
            enc.submit(imgs[0].save, str(tmp_path / "no_such_dir" / "x.png"))
    # level-1 hand-off PNGs are lossless: same pixels as PIL's default level
    a = (np.random.default_rng(0).random((128, 96, 3)) * 255).astype(np.uint8)
    Image.fromarray(a).save(str(tmp_path / "l1.png"), compress_level=1)
    Image.fromarray(a).save(str(tmp_path / "l6.png"))
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "l1.png"))), np.asarray(Image.open(str(tmp_path / "l6.png"))))


def _unit_sphere(n_faces):
    from unitex_amd.texturetools import meshes
    v, f, _ = meshes.sphere_with_faces(n_faces)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return meshes.clean_mesh(v, f)


def _edge_face_counts(f):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    return np.unique(e, axis=0, return_counts=True)[1]


def test_qem_decimation_hausdorff_bound_on_a_sphere():
    """utx_mesh_decimate_qem (host C++ in libunitex_hip.so; the reference: open3d simplify_quadric_decimation, uv_atlas.py:155-163 [3p]) on an
    analytic mesh: an 80 000-face unit sphere reduced to 20 000 / 5 000 faces stays a closed, outward-oriented 2-manifold whose vertices lie
    on the sphere and whose faces sag by no more than twice what ANY triangulation with that many faces must (sagitta of an equilateral
    triangulation: edge^2 / 8 with edge^2 = 4 (4 pi / F) / sqrt 3) -- the one-sided Hausdorff distance to the sphere."""
    from unitex_amd.texturetools import meshes
    v, f = _unit_sphere(80000)
    for target in (20000, 5000):
        v2, f2 = meshes.decimate_qem(v, f, target)
        assert target - 2 <= len(f2) <= target
        assert (_edge_face_counts(f2) == 2).all(), "closed 2-manifold"
        c = v2[f2].mean(1)
        n = np.cross(v2[f2[:, 1]] - v2[f2[:, 0]], v2[f2[:, 2]] - v2[f2[:, 0]])
        assert ((n * c).sum(1) > 0).all(), "orientation"
        sag = (4.0 * (4.0 * np.pi / len(f2)) / np.sqrt(3.0)) / 8.0
        assert np.abs(np.linalg.norm(v2, axis=1) - 1.0).max() <= 2.0 * sag
        assert (1.0 - np.linalg.norm(c, axis=1)).max() <= 2.0 * sag, "face centres sag %.2e, bound %.2e" % ((1.0 - np.linalg.norm(c, axis=1)).max(), 2 * sag)
    # an open surface keeps its border (boundary constraint planes): a flat 40 x 40 grid decimated 8-fold stays flat and keeps its square outline
    n_ = 41
    xx, yy = np.meshgrid(np.linspace(0, 1, n_), np.linspace(0, 1, n_))
    gv = np.stack([xx.ravel(), yy.ravel(), np.zeros(n_ * n_)], 1).astype(np.float32)
    gf = np.array([[i * n_ + j, i * n_ + j + 1, (i + 1) * n_ + j + 1] for i in range(n_ - 1) for j in range(n_ - 1)] +
                  [[i * n_ + j, (i + 1) * n_ + j + 1, (i + 1) * n_ + j] for i in range(n_ - 1) for j in range(n_ - 1)], np.int32)
    v3, f3 = meshes.decimate_qem(gv, gf, 400)
    assert len(f3) <= 400 and np.abs(v3[:, 2]).max() < 1e-6
    a = 0.5 * np.linalg.norm(np.cross(v3[f3[:, 1]] - v3[f3[:, 0]], v3[f3[:, 2]] - v3[f3[:, 0]]), axis=1).sum()
    assert abs(a - 1.0) < 1e-3 and v3[:, :2].min() > -1e-6 and v3[:, :2].max() < 1 + 1e-6


def test_loop_subdivision_and_simple_smoothing_properties():
    """meshes.subdivide_loop (open3d subdivide_loop [3p], uv_atlas.py:164-165) and meshes.smooth_simple (filter_smooth_simple, :169): 1:4 split per
    iteration; a regular flat lattice is a fixed point of the Loop masks (interior old vertices stay, everything stays planar, border vertices
    stay on the border lines); a closed surface stays closed and shrinks towards its limit surface; smoothing keeps connectivity and
    contracts a sphere uniformly."""
    from unitex_amd.texturetools import meshes
    n_ = 9
    xx, yy = np.meshgrid(np.arange(n_), np.arange(n_))
    gv = np.stack([xx.ravel(), yy.ravel(), np.zeros(n_ * n_)], 1).astype(np.float32)
    gf = np.array([[i * n_ + j, i * n_ + j + 1, (i + 1) * n_ + j + 1] for i in range(n_ - 1) for j in range(n_ - 1)] +
                  [[i * n_ + j, (i + 1) * n_ + j + 1, (i + 1) * n_ + j] for i in range(n_ - 1) for j in range(n_ - 1)], np.int32)
    lv, lf = meshes.subdivide_loop(gv, gf, 1)
    assert len(lf) == 4 * len(gf) and len(lv) == len(gv) + len(np.unique(np.sort(np.concatenate([gf[:, [0, 1]], gf[:, [1, 2]], gf[:, [2, 0]]]), 1), axis=0))
    inner = [i * n_ + j for i in range(1, n_ - 1) for j in range(1, n_ - 1)]
    assert np.abs(lv[:, 2]).max() == 0 and np.abs(lv[inner] - gv[inner]).max() < 1e-6
    assert lv[:, :2].min() >= -1e-6 and lv[:, :2].max() <= n_ - 1 + 1e-6
    sv, sf = _unit_sphere(300)
    l2, f2 = meshes.subdivide_loop(sv, sf, 2)
    assert len(f2) == 16 * len(sf) and (_edge_face_counts(f2) == 2).all()
    r = np.linalg.norm(l2, axis=1)
    assert 0.93 < r.min() and r.max() < 1.0            # towards the limit surface, inside the control mesh's circumsphere
    s3 = meshes.smooth_simple(l2, f2, 3)
    rs = np.linalg.norm(s3, axis=1)
    assert s3.shape == l2.shape and rs.max() < r.max() and rs.min() > 0.9
    # prepare_blank_mesh follows the reference's branches: below min_faces -> exactly two Loop iterations (x16 faces)


def test_projection_frames_bound_the_stretch():
    """26 projection directions: every unit normal is within 27.6 degrees of one of them -> planar projection stretches areas by <= 12.8 %, inside the
    reference's UVAtlas bound max_stretch = 1/6 (uv_atlas.py:171); the six coordinate axes (round 2) allow 73 %.  Frames are right-handed."""
    from unitex_amd.texturetools import meshes
    rng = np.random.default_rng(0)
    p = rng.normal(size=(200000, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    for n, bound in ((26, 0.1281), (6, 0.7321)):
        d, u, v = meshes.projection_frames(n)
        assert len(d) == n and np.allclose(np.cross(u, v), d, atol=1e-12) and np.allclose((u * v).sum(1), 0, atol=1e-12)
        worst = 1.0 / (p @ d.T).max(1).min() - 1.0
        assert worst <= bound, (n, worst)
    assert 1.0 / (p @ meshes.projection_frames(26)[0].T).max(1).min() - 1.0 <= 1.0 / 6.0



def _write_ply(path, v, polys, fmt, uv=None, extra_element=False):
    """a PLY writer for the reader's test only: vertex (x y z [+ a skipped uchar property] [+ s t]), optionally an element in front that must be skipped, face lists
    with a uchar count and int indices plus one scalar property behind the list"""
    import struct as st
    e = {"ascii": None, "binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
    h = ["ply", "format %s 1.0" % fmt, "comment written by the test"]
    if extra_element:
        h += ["element camera 2", "property float cx", "property short cid"]
    h += ["element vertex %d" % len(v), "property float x", "property float y", "property float z", "property uchar red"]
    if uv is not None:
        h += ["property float s", "property float t"]
    h += ["element face %d" % len(polys), "property list uchar int vertex_indices", "property float quality", "end_header"]
    body = b""
    if e is None:
        rows = []
        if extra_element:
            rows += ["0.5 7", "1.5 -3"]
        for i, p_ in enumerate(v):
            rows.append(" ".join(repr(float(x)) for x in p_) + " %d" % (i % 256) + ("" if uv is None else " %r %r" % (float(uv[i, 0]), float(uv[i, 1]))))
        for p_ in polys:
            rows.append("%d " % len(p_) + " ".join(str(int(i)) for i in p_) + " 0.25")
        body = ("\n".join(rows) + "\n").encode()
    else:
        if extra_element:
            body += st.pack(e + "fh", 0.5, 7) + st.pack(e + "fh", 1.5, -3)
        for i, p_ in enumerate(v):
            body += st.pack(e + "fffB", float(p_[0]), float(p_[1]), float(p_[2]), i % 256)
            if uv is not None:
                body += st.pack(e + "ff", float(uv[i, 0]), float(uv[i, 1]))
        for p_ in polys:
            body += st.pack(e + "B%dif" % len(p_), len(p_), *[int(i) for i in p_], 0.25)
    with open(path, "wb") as f:
        f.write(("\n".join(h) + "\n").encode() + body)


def test_interchange_mesh_formats_read_back_the_same_mesh(tmp_path):
    """load_mesh on .ply (ascii / both binary byte orders; triangles, mixed polygons, skipped properties and elements, per-vertex s t), .stl (binary / ascii: corner
    merge), .off and .gltf (data URIs and a side file) returns the mesh that was written -- the reference hands any of these to trimesh.load (io/mesh_loader.py:22-30)."""
    import base64
    import json
    import struct as st
    v, f, uv = meshes.sphere_with_faces(300)
    v = v.astype(np.float32); f = f.astype(np.int32); uv = uv.astype(np.float32)
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        p = str(tmp_path / ("tri_%s.ply" % fmt))
        _write_ply(p, v, f, fmt, uv=uv, extra_element=(fmt != "ascii"))
        v2, f2, uv2, tex = meshes.load_mesh(p)
        assert tex is None and np.array_equal(v2, v) and np.array_equal(f2, f) and np.array_equal(uv2, uv), fmt
        # quads + a pentagon + triangles in one file: the row-by-row path, fanned
        polys = [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9, 10, 11], [2, 1, 0]]
        p = str(tmp_path / ("poly_%s.ply" % fmt))
        _write_ply(p, v, polys, fmt)
        v3, f3, uv3, _ = meshes.load_mesh(p)
        assert uv3 is None and np.array_equal(v3, v)
        assert f3.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9], [7, 9, 10], [7, 10, 11], [2, 1, 0]], fmt
    # all-quad binary file: the fixed-size fast path with lists of four
    p = str(tmp_path / "quads.ply")
    _write_ply(p, v, [[0, 1, 2, 3], [4, 5, 6, 7]], "binary_little_endian")
    assert meshes.load_mesh(p)[1].tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]]
    # STL: a soup of the same triangles; corners merge back to the vertices the faces use, in first-use order
    soup = v[f].reshape(-1, 3)
    pb = str(tmp_path / "m.stl")
    with open(pb, "wb") as fh:
        fh.write(b"solid looks like ascii but is binary".ljust(80, b" ") + st.pack("<I", len(f)))
        for t in v[f]:
            fh.write(st.pack("<12fH", 0.0, 0.0, 0.0, *t.reshape(-1).tolist(), 0))
    pa = str(tmp_path / "a.stl")
    with open(pa, "w") as fh:
        fh.write("solid s\n")
        for t in v[f]:
            fh.write("facet normal 0 0 0\n outer loop\n" + "".join("  vertex %r %r %r\n" % tuple(float(x) for x in c) for c in t) + " endloop\nendfacet\n")
        fh.write("endsolid s\n")
    for p in (pb, pa):
        vs, fs, uvs, _ = meshes.load_mesh(p)
        assert uvs is None and np.array_equal(vs[fs].reshape(-1, 3), soup), p
        first_use = np.sort(np.unique(soup, axis=0, return_index=True)[1])      # the sphere's seam vertices share positions: they merge too
        assert np.array_equal(vs, soup[first_use]), p
    # OFF: counts on the header line / on the next line, a COFF with vertex colours, a face colour behind the indices, comments
    po = str(tmp_path / "m.off")
    with open(po, "w") as fh:
        fh.write("OFF\n# a comment\n%d %d 0\n" % (len(v), len(f)) + "".join("%r %r %r\n" % tuple(float(x) for x in p_) for p_ in v)
                 + "".join("3 %d %d %d 255 0 0\n" % tuple(t) for t in f))
    vo, fo, _, _ = meshes.load_mesh(po)
    assert np.array_equal(vo, v) and np.array_equal(fo, f)
    pc = str(tmp_path / "c.off")
    with open(pc, "w") as fh:
        fh.write("COFF %d 2 0\n" % len(v) + "".join("%r %r %r 10 20 30 255\n" % tuple(float(x) for x in p_) for p_ in v) + "4 0 1 2 3\n3 4 5 6\n")
    vc, fc, _, _ = meshes.load_mesh(pc)
    assert np.array_equal(vc, v) and fc.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6]]
    # glTF JSON: the GLB this package writes, re-containered -- buffer as a data URI, then as a side file with an external image
    tex = (np.random.default_rng(1).random((16, 24, 3)) * 255).astype(np.uint8)
    pg = str(tmp_path / "m.glb")
    meshes.save_glb(pg, v, f, uv, tex)
    ref = meshes.load_glb(pg)
    blob = open(pg, "rb").read()
    jl = st.unpack_from("<I", blob, 12)[0]
    js = json.loads(blob[20:20 + jl].decode())
    binc = blob[20 + jl + 8:]
    js1 = json.loads(json.dumps(js))
    js1["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(binc).decode()
    p1 = str(tmp_path / "inline.gltf")
    json.dump(js1, open(p1, "w"))
    js2 = json.loads(json.dumps(js))
    js2["buffers"][0]["uri"] = "side%20file.bin"
    open(str(tmp_path / "side file.bin"), "wb").write(binc)
    bv = js["bufferViews"][js["images"][0]["bufferView"]]
    open(str(tmp_path / "albedo.png"), "wb").write(binc[bv["byteOffset"]:bv["byteOffset"] + bv["byteLength"]])
    js2["images"][0] = {"uri": "albedo.png"}
    p2 = str(tmp_path / "side.gltf")
    json.dump(js2, open(p2, "w"))
    for p in (p1, p2):
        got = meshes.load_mesh(p)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), p
    # refusals
    import pytest
    js3 = json.loads(json.dumps(js))
    js3["buffers"][0]["uri"] = "../escape.bin"
    json.dump(js3, open(str(tmp_path / "escape.gltf"), "w"))
    with pytest.raises(ValueError, match="outside"):
        meshes.load_mesh(str(tmp_path / "escape.gltf"))
    open(str(tmp_path / "cloud.ply"), "w").write("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n")
    with pytest.raises(ValueError):
        meshes.load_mesh(str(tmp_path / "cloud.ply"))
    with pytest.raises(NotImplementedError):
        meshes.load_mesh(str(tmp_path / "m.fbx"))


def test_camera_samplers_match_reference_fixture():
    """the camera generators off the pipeline's path (camera/generator.py:129-151,187-200) against outputs captured from the reference (G12): the seeded hemisphere / sphere /
    near-front samplers name the same cameras for the same seed, the canonical Euler grid is the reference's, bit for bit; plus what any camera set must satisfy."""
    f = np.load(os.path.join(HERE, "golden", "g12_camera_samplers.npz"))
    got = {
        "canonical_888": camera.generate_canonical_views_c2ws(radius=2.8, steps=(8, 8, 8)),
        "canonical_325": camera.generate_canonical_views_c2ws(radius=1.7, steps=(3, 2, 5)),
        "hemisphere_semi_s7": camera.generate_hemisphere_views_c2ws(37, radius=2.8, seed=7, semi=True),
        "hemisphere_full_s7": camera.generate_hemisphere_views_c2ws(37, radius=2.8, seed=7, semi=False),
        "semisphere_s3": camera.generate_semisphere_views_c2ws(29, radius=2.0, seed=3, hemi=False),
        "semisphere_hemi_s3": camera.generate_semisphere_views_c2ws(29, radius=2.0, seed=3, hemi=True),
        "near_front_s5": camera.generate_near_front_views_c2ws(31, radius=2.8, scale_x=0.5, scale_y=0.25, seed=5),
        "near_front_default_s11": camera.generate_near_front_views_c2ws(8, seed=11),
    }
    assert set(got) == set(f.files)
    for k, v in got.items():
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), f[k]), k
    # what holds beyond the bits (the reference's look-at frame is orthonormal only for equatorial cameras -- its x axis is e3 x z, not normalised -- and its hemisphere / near-front
    # samplers draw per COMPONENT of the axis array, so those cameras are not on the sphere: reproduced, not repaired)
    for k, r in (("canonical_888", 2.8), ("canonical_325", 1.7)):
        R = got[k][:, :3, :3].double()
        assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand_as(R), atol=1e-5), k
        assert torch.allclose(got[k][:, :3, 3].double().norm(dim=-1), torch.full((R.shape[0],), r, dtype=torch.float64), atol=1e-5), k
    for k in ("semisphere_s3", "semisphere_hemi_s3"):
        assert torch.allclose(got[k][:, :3, 3].norm(dim=-1), torch.full((29,), 2.0), atol=1e-5), k
    assert (got["semisphere_hemi_s3"][:, 1, 3] >= 0).all() and (got["semisphere_s3"][:, 1, 3] < 0).any()      # world up = the camera frame's y row after the axis permutation
    for k, v in got.items():
        assert torch.equal(v[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(v.shape[0], 4)), k
    # unseeded calls draw from the global generator and still return well-formed cameras
    assert camera.generate_semisphere_views_c2ws(5).shape == (5, 4, 4)
