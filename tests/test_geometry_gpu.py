"""GPU parity tests (through the C ABI) of the render / back-projection kernels against the CPU oracle
(oracle/geom_ref.c + geom_ref.py).  Bar: BIT-EXACT for every integer / index / mask output and for
the float32 outputs whose expression order is fixed on both sides (raster record, interpolation, gather,
pull-push); stated tolerance only where libm (powf) is involved (lens blur)."""
import os

import numpy as np
import pytest
import torch

from oracle import geom_ref as G
from unitex_amd.texturetools.meshes import sphere_with_faces

pytestmark = pytest.mark.gpu


def _ops():
    from unitex_amd.texturetools import ops
    return ops


def _cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def _scene(n_faces=3000, T=256, HW=128, seed=0):
    verts, faces, uvs = sphere_with_faces(n_faces)
    c2ws = G.box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]  # f, r, t, b, l, d (export_nvdiffrast_video.py:926-936)
    intr = G.intrinsics(1.0, 1.0, fov=False)
    mvp = G.mvp_matrices(c2ws, intr, perspective=False)
    rng = np.random.default_rng(seed)
    return dict(verts=verts, faces=faces, uvs=uvs, c2ws=c2ws, mvp=mvp, T=T, HW=HW, rng=rng)


def test_transform_and_raster_bit_exact():
    ops = _ops()
    s = _scene()
    clip_ref = G.transform_points(s["verts"], s["mvp"])
    clip, ndc = ops.transform_points(_cu(s["verts"]), _cu(s["mvp"]))
    assert np.array_equal(clip.cpu().numpy(), clip_ref)
    assert np.array_equal(ndc.cpu().numpy(), (clip_ref[..., :2] / clip_ref[..., 3:4]).astype(np.float32))
    faces_d = _cu(s["faces"])
    for v in range(6):
        r_ref = G.rasterize(clip_ref[v], s["faces"], 128, 128)
        r = ops.rasterize(clip[v].contiguous(), faces_d, 128, 128).cpu().numpy()
        assert np.array_equal(r, r_ref), "view raster %d differs" % v
        assert (r_ref[..., 3] > 0).mean() > 0.3
    uvclip = np.concatenate([s["uvs"] * 2 - 1, np.zeros((len(s["uvs"]), 1), np.float32), np.ones((len(s["uvs"]), 1), np.float32)], -1)
    for T in (64, 256, 1024):
        r_ref = G.rasterize(uvclip, s["faces"], T, T)
        r = ops.rasterize(_cu(uvclip), faces_d, T, T)
        assert np.array_equal(r.cpu().numpy(), r_ref)
        a_ref = G.interpolate(s["verts"], r_ref, s["faces"])
        a = ops.interpolate(_cu(s["verts"]), r, faces_d).cpu().numpy()
        assert np.array_equal(a, a_ref)


def test_raster_big_triangles_and_edge_cases():
    ops = _ops()
    # two big triangles covering the viewport + a sliver + a degenerate + an off-screen one
    pos = np.array([[-1, -1, 0.5, 1], [1, -1, 0.5, 1], [1, 1, 0.5, 1], [-1, 1, 0.5, 1],
                    [-0.5, -0.5, 0.2, 1], [0.5, -0.5001, 0.2, 1], [0.5, -0.5, 0.2, 1],
                    [0.1, 0.1, 0.0, 1], [0.1, 0.1, 0.0, 1], [0.3, 0.3, 0, 1],
                    [2, 2, 0, 1], [3, 2, 0, 1], [2, 3, 0, 1],
                    [-0.3, 0.2, -0.4, 2.0], [0.4, 0.3, -0.4, 1.5], [0.0, 0.9, 0.5, 1.0]], dtype=np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12], [13, 14, 15], [2, 1, 0]], dtype=np.int32)
    for H, W in ((97, 131), (512, 512)):
        ref = G.rasterize(pos, tri, H, W)
        got = ops.rasterize(_cu(pos), _cu(tri), H, W).cpu().numpy()
        assert np.array_equal(got, ref)
        assert (ref[..., 3] > 0).all()


def test_bvh_build_and_trace_bit_exact():
    ops = _ops()
    for nf in (1, 2, 500, 20000):
        verts, faces, _ = sphere_with_faces(max(nf, 64))
        faces = faces[:nf] if nf < 64 else faces
        ref = G.BVH(verts, faces)
        b = ops.BVH(_cu(verts), _cu(faces))
        info, aabb, codes, idx = b.arrays()
        assert np.array_equal(codes.cpu().numpy().view(np.uint32), ref.codes), "sorted morton codes"
        assert np.array_equal(idx.cpu().numpy(), ref.order), "stable sort order"
        assert np.array_equal(info.cpu().numpy(), ref.info), "hierarchy"
        assert np.array_equal(aabb.cpu().numpy(), ref.aabb), "node boxes"
        rng = np.random.default_rng(nf)
        R = 20000
        ro = np.stack([rng.uniform(-1, 1, R), rng.uniform(-1, 1, R), np.full(R, 2.8)], -1).astype(np.float32)
        rd = np.tile(np.array([[0, 0, -1]], np.float32), (R, 1))
        d2 = rng.normal(size=(R, 3)).astype(np.float32)
        o2 = (-3.0 * d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32) + rng.uniform(-0.3, 0.3, (R, 3)).astype(np.float32)
        from unitex_amd import _lib
        assert 0 <= b.depth() <= 60, "these trees take the stackless packed traversal"
        for o, d in ((ro, rd), (o2, d2)):
            t_ref = ref.trace(o, d)
            t = b.trace(_cu(o), _cu(d)).cpu().numpy()          # product path: stackless walk over the packed tree
            assert np.array_equal(t, t_ref)
            try:                                                # the reference's 64-entry stack walk over the unpacked arrays: same ids
                _lib.set_option("UTX_BVH_STACK_WALK", 1)
                t_stack = b.trace(_cu(o), _cu(d)).cpu().numpy()
            finally:
                _lib.set_option("UTX_BVH_STACK_WALK", 0)
            assert np.array_equal(t_stack, t_ref)
            t_cnt, visited = b.trace_count(_cu(o), _cu(d))      # counting variant: same ids, every ray visits at least the root
            assert np.array_equal(t_cnt.cpu().numpy(), t_ref) and visited >= len(o)
        if nf >= 500:
            assert (t_ref >= 0).mean() > 0.3


def _full_case(n_faces, T, HW, seed):
    s = _scene(n_faces, T, HW, seed)
    verts, faces, uvs, mvp = s["verts"], s["faces"], s["uvs"], s["mvp"]
    rng = s["rng"]
    clip = G.transform_points(verts, mvp)
    vndc = (clip[..., :2] / clip[..., 3:4]).astype(np.float32)
    uvclip = np.concatenate([uvs * 2 - 1, np.zeros((len(uvs), 1), np.float32), np.ones((len(uvs), 1), np.float32)], -1)
    rast2d = G.rasterize(uvclip, faces, T, T)
    fn = G.face_normals(verts, faces)
    dirs = (-s["c2ws"][:, :3, 2]).astype(np.float32)
    # view images: smooth random colours, alpha = view-space coverage (mv_to_pcd alpha, renderer_inverse.py:185)
    imgs = np.zeros((6, HW, HW, 4), np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, HW), np.linspace(0, 1, HW), indexing="ij")
    for v in range(6):
        ph = rng.uniform(0, 6.28, 6)
        for c in range(3):
            imgs[v, ..., c] = 0.5 + 0.5 * np.sin(7 * xx + ph[c]) * np.cos(5 * yy + ph[c + 3])
        imgs[v, ..., 3] = (G.rasterize(clip[v], faces, HW, HW)[..., 3] > 0)
        # punch a hole in every view's alpha so that some covered texels are seen by no view (exercises NN fill)
        imgs[v, ..., 3] *= (((xx - 0.5) ** 2 + (yy - 0.5) ** 2) > 0.16 ** 2) | (xx < 0.3)
    return dict(verts=verts, faces=faces, uvs=uvs, vndc=vndc, rast2d=rast2d, fn=fn, dirs=dirs, imgs=imgs)


def test_backprojection_ray_packets_on_an_atlas_that_is_no_multiple_of_the_ray_tiles():
    """the packet walk maps a wave to an 8 x 8 texel tile and a workgroup to 16 x 16: a 250 x 250 atlas has ragged tiles on two sides"""
    from unitex_amd import _lib
    ops = _ops()
    c = _full_case(3000, 250, 96, seed=11)
    bvh_ref = G.BVH(c["verts"], c["faces"])
    col_ref, rv_ref, ao_ref = G.backproject(c["rast2d"], c["verts"], c["faces"], c["fn"], c["vndc"], c["dirs"], c["imgs"], bvh_ref)
    vd, fd = _cu(c["verts"]), _cu(c["faces"])
    bvh = ops.BVH(vd, fd)
    assert _lib.get_options()["UTX_BVH_PACKET"] == 1
    col, rv, ao = ops.backproject(_cu(c["rast2d"]), vd, fd, _cu(c["fn"]), _cu(c["vndc"]), _cu(c["dirs"]), _cu(c["imgs"]), bvh)
    assert np.array_equal(rv.cpu().numpy(), rv_ref) and np.array_equal(ao.cpu().numpy(), ao_ref) and np.array_equal(col.cpu().numpy(), col_ref)
    assert 0.05 < rv_ref.mean() < 0.9


@pytest.mark.parametrize("n_faces,T,HW", [(3000, 256, 128), (20000, 512, 256)])
def test_backprojection_chain_bit_exact(n_faces, T, HW):
    from unitex_amd import _lib
    ops = _ops()
    c = _full_case(n_faces, T, HW, seed=n_faces)
    bvh_ref = G.BVH(c["verts"], c["faces"])
    col_ref, rv_ref, ao_ref = G.backproject(c["rast2d"], c["verts"], c["faces"], c["fn"], c["vndc"], c["dirs"], c["imgs"], bvh_ref)
    vd, fd = _cu(c["verts"]), _cu(c["faces"])
    bvh = ops.BVH(vd, fd)
    rast_d = _cu(c["rast2d"])
    # the three walks of the same tree -- wave-wide packets over 8 x 8 texel tiles (round 4, the default), one thread per ray (stackless), the reference's
    # stack walk -- must give the oracle's bits
    assert _lib.get_options()["UTX_BVH_PACKET"] == 1
    try:
        for packet, stack in ((0, 1), (0, 0), (1, 0)):
            _lib.set_option("UTX_BVH_PACKET", packet); _lib.set_option("UTX_BVH_STACK_WALK", stack)
            col, rv, ao = ops.backproject(rast_d, vd, fd, _cu(c["fn"]), _cu(c["vndc"]), _cu(c["dirs"]), _cu(c["imgs"]), bvh)
            assert np.array_equal(rv.cpu().numpy(), rv_ref), "ray visibility mask (packet %d, stack walk %d)" % (packet, stack)
            assert np.array_equal(ao.cpu().numpy(), ao_ref), "alpha mask"
            assert np.array_equal(col.cpu().numpy(), col_ref), "gathered colours"
    finally:
        _lib.set_option("UTX_BVH_PACKET", 1); _lib.set_option("UTX_BVH_STACK_WALK", 0)
    assert 0.05 < rv_ref.mean() < 0.9
    # view sharding: views [2,4) only
    col2, rv2, ao2 = ops.backproject(rast_d, vd, fd, _cu(c["fn"]), _cu(c["vndc"]), _cu(c["dirs"]), _cu(c["imgs"]), bvh,
                                     view_begin=2, view_count=2)
    assert np.array_equal(rv2.cpu().numpy()[2:4], rv_ref[2:4]) and rv2.cpu().numpy()[[0, 1, 4, 5]].sum() == 0
    # hole filling + coverage + alpha
    mask2d = c["rast2d"][..., 3] > 0
    vis_ref = G.dilate_visibility(rv_ref, mask2d, ao_ref)
    vis = ops.dilate_visibility(rv, ao, rast_d)
    assert np.array_equal(vis.cpu().numpy().astype(bool), vis_ref)
    # priority composite
    atlas_ref, seen_ref, win_ref, bnd_ref = G.composite(col_ref, vis_ref)
    atlas, winner = ops.composite(col, vis, G.PRIORITY)
    assert np.array_equal(winner.cpu().numpy(), win_ref), "composite winner (texel index parity)"
    assert np.array_equal(atlas.cpu().numpy(), atlas_ref)
    # seam mask
    seam_ref = G.seam_mask(bnd_ref, mask2d)
    seam = ops.seam_mask(winner, rast_d)
    assert np.array_equal(seam.cpu().numpy().astype(bool), seam_ref)
    # NN fill (exact, brute-force oracle with the same float32 distance + tie rule)
    pos_ref = G.interpolate(c["verts"], c["rast2d"], c["faces"])
    pos = ops.interpolate(vd, rast_d, fd)
    if T <= 256:
        filled_ref, idx_ref = G.nn_fill_brute(atlas_ref, win_ref, c["rast2d"], pos_ref)
        idx = ops.nn_fill(atlas, winner, rast_d, pos, want_index=True)
        assert np.array_equal(idx.cpu().numpy().reshape(T, T), idx_ref), "nearest-seen-texel index"
        assert np.array_equal(atlas.cpu().numpy(), filled_ref)
        # independent check of the definition: scipy kd-tree agrees except on exact float ties
        f2, _ = G.nn_fill(atlas_ref, seen_ref, mask2d, pos_ref)
        assert (np.abs(f2 - filled_ref).max(-1) > 0).mean() < 1e-3
    else:
        ops.nn_fill(atlas, winner, rast_d, pos)
        filled_ref, _ = G.nn_fill(atlas_ref, seen_ref, mask2d, pos_ref)
        assert (np.abs(atlas.cpu().numpy() - filled_ref).max(-1) > 0).mean() < 1e-3
        filled_ref = atlas.cpu().numpy()
    # lens blur on the seam: powf differs between libm and the GPU -> 2e-6 abs
    blur_ref = G.lens_blur_collapsed(filled_ref, seam_ref)
    blur = ops.lens_blur_seam(atlas, seam)
    assert np.abs(blur.cpu().numpy() - blur_ref).max() < 2e-6
    sep = G.lens_blur(filled_ref.transpose(2, 0, 1)).transpose(1, 2, 0)  # the reference's separable formulation
    assert np.abs(np.where(seam_ref[..., None], sep, filled_ref) - blur.cpu().numpy()).max() < 2e-5
    # pull-push
    pp_in = blur.cpu().numpy()
    pp_ref = G.pull_push(pp_in.transpose(2, 0, 1), mask2d).transpose(1, 2, 0)
    pp = ops.pull_push(blur, _cu(mask2d.astype(np.uint8)))
    assert np.array_equal(pp.cpu().numpy(), pp_ref), "pull-push"
    # uint8 by truncation + vertical flip
    u8 = ops.to_u8(pp, flip=True).cpu().numpy()
    assert np.array_equal(u8, G.tensor_to_u8(pp_ref)[::-1])


def test_nn_fill_result_does_not_depend_on_the_cell_grid():
    """the NN fill picks its uniform grid from the atlas size; the search is exact for ANY grid (ring r+1 only holds points at least r cells away), so the
    nearest-seen-texel index must be the same for every forced grid (UTX_NN_GRID = 64 / 128 / 256), equal to the grid chosen by size, and equal to the brute-force oracle
    -- also with positions OUTSIDE [-1, 1] (clamped into edge cells: scaled copy of the same case)."""
    from unitex_amd import _lib
    ops = _ops()
    c = _full_case(3000, 250, 96, seed=5)
    T = 250
    bvh_ref = G.BVH(c["verts"], c["faces"])
    col_ref, rv_ref, ao_ref = G.backproject(c["rast2d"], c["verts"], c["faces"], c["fn"], c["vndc"], c["dirs"], c["imgs"], bvh_ref)
    mask2d = c["rast2d"][..., 3] > 0
    vis_ref = G.dilate_visibility(rv_ref, mask2d, ao_ref)
    atlas_ref, seen_ref, win_ref, bnd_ref = G.composite(col_ref, vis_ref)
    rast_d = _cu(c["rast2d"])
    for scale in (1.0, 1.7):                     # 1.7: a third of the sphere's texels lie outside the grid's cube
        pos_ref = (G.interpolate(c["verts"], c["rast2d"], c["faces"]) * np.float32(scale)).astype(np.float32)
        filled_ref, idx_ref = G.nn_fill_brute(atlas_ref, win_ref, c["rast2d"], pos_ref)
        assert ((idx_ref >= 0) & (win_ref < 0)).sum() > 100, "the case must have texels to fill"
        got = {}
        try:
            for grid in (0, 64, 128, 256):
                _lib.set_option("UTX_NN_GRID", grid)
                atlas = _cu(atlas_ref.copy())
                idx = ops.nn_fill(atlas, _cu(win_ref), rast_d, _cu(pos_ref), want_index=True)
                got[grid] = (idx.cpu().numpy().reshape(T, T), atlas.cpu().numpy())
        finally:
            _lib.set_option("UTX_NN_GRID", 0)
        for grid, (idx, atlas) in got.items():
            assert np.array_equal(idx, idx_ref), "grid %d, scale %g: nearest-seen-texel index" % (grid, scale)
            assert np.array_equal(atlas, filled_ref)


def test_orbit_video_frames_bit_exact_and_mux(tmp_path):
    """export_orbit_video (turntable of the textured mesh): perspective raster + fused UV interpolation / bilinear
    texture fetch / background composite on the GPU vs the oracle, bit for bit; container round trip."""
    import io
    from PIL import Image
    from unitex_amd.texturetools import camera, meshes, video
    from unitex_amd.texturetools.video import VideoExporter
    v, f, uv = meshes.sphere_with_faces(4000)
    rng = np.random.default_rng(5)
    tex = (rng.random((96, 128, 3)) * 255).astype(np.uint8)          # top-down image as stored in the GLB
    R, n = 160, 6
    path = str(tmp_path / "turn.mp4")
    frames = VideoExporter(device="cuda:0").export_orbit_video((v, f, uv, tex), path, n_frames=n, render_size=R, return_frames=True,
                                                               save_cover=True)
    # cameras: the host-side matrix product is pinned separately (fixture G4); feed the oracle the same fp32 MVPs so
    # that the comparison below isolates the kernels (transform, raster, shade) and can be exact
    c2ws = camera.generate_orbit_views_c2ws(n + 1, radius=2.8, height=0.0, theta_0=0.0, degree=True)[:n]
    intr = camera.generate_intrinsics(49.1, 49.1, fov=True, degree=True)
    mvp = torch.matmul(camera.intr_to_proj(intr, perspective=True), camera.c2w_to_w2c(c2ws)).numpy()
    clip = G.transform_points(v, mvp)
    texf = (tex[::-1].astype(np.float32) / np.float32(255.0))
    cover = 0
    for i in range(n):
        rast = G.rasterize(clip[i], f, R, R)
        ref = G.texture_shade(rast, uv, f, texf, bg=(1.0, 1.0, 1.0))
        assert np.array_equal(frames[i], ref), "frame %d differs in %d bytes" % (i, int((frames[i] != ref).sum()))
        cover += int((rast[..., 3] > 0).sum())
    assert cover > 0.2 * n * R * R
    fps, jpgs = video.read_mjpeg_mp4(path)
    assert fps == 15 and len(jpgs) == n
    im = np.asarray(Image.open(io.BytesIO(jpgs[0])).convert("RGB")).astype(np.int32)
    assert np.abs(im - frames[0]).mean() < 20.0
    assert os.path.exists(os.path.splitext(path)[0] + "_cover.png")


def test_bvh_on_reference_bunny_bit_exact():
    """GPU LBVH build + trace on the reference's own test mesh / ray set (fixture G8) vs the oracle, bit for bit."""
    ops = _ops()
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_bunny.npz"))
    verts, faces, ro, rd = f["verts"], f["faces"], f["rays_o"], f["rays_d"]
    ref = G.BVH(verts, faces)
    b = ops.BVH(_cu(verts), _cu(faces))
    info, aabb, codes, idx = b.arrays()
    assert np.array_equal(codes.cpu().numpy().view(np.uint32), ref.codes)
    assert np.array_equal(idx.cpu().numpy(), ref.order)
    assert np.array_equal(info.cpu().numpy(), ref.info)
    assert np.array_equal(aabb.cpu().numpy(), ref.aabb)
    t = b.trace(_cu(ro), _cu(rd)).cpu().numpy()
    assert np.array_equal(t, ref.trace(ro, rd))
    clear = f["edge_margin"] > 1e-4
    assert np.array_equal((t >= 0)[clear], f["hit"][clear])       # float64 brute-force mask stored with the fixture


def test_face_normals_bit_exact():
    ops = _ops()
    verts, faces, _ = sphere_with_faces(5000)
    faces = np.concatenate([faces, [[0, 0, 1]]]).astype(np.int32)     # a degenerate face: normalised with eps 1e-12 -> zeros
    got = ops.face_normals(_cu(verts), _cu(faces)).cpu().numpy()
    assert np.array_equal(got, G.face_normals(verts, faces))


def test_condition_render_matches_reference_fixture():
    """fixture G9 = the reference's own VideoExporter.export_condition + simple_rendering (dr seams on the oracle
    rasteriser): the HIP condition render must reproduce its alpha / world-position / normal grids.  Coverage (alpha) is
    exact; colours may differ by one uint8 step where fp32 rounding order (torch normalize / lerp vs the fused kernel)
    flips a truncation."""
    from unitex_amd.texturetools.video import VideoExporter
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_export_condition.npz"))
    # the fixture was captured with the reference's own area-weighted fallback normals (no trimesh in the harness)
    out = VideoExporter(device="cuda:0", normal_weighting="area").export_condition((f["verts"], f["faces"]), geometry_scale=0.95, n_views=6, n_rows=2, n_cols=3,
                                                          H=64, W=64, fov_deg=49.1, scale=1.0, perspective=False, orbit=False,
                                                          background="grey", return_image=True, return_camera=True)
    assert np.array_equal(np.asarray(out["alpha"]), f["alpha"])
    for key in ("ccm", "normal"):
        d = np.abs(np.asarray(out[key]).astype(np.int32) - f[key].astype(np.int32))
        assert d.max() <= 1, "%s differs by %d" % (key, int(d.max()))
        assert (d > 0).mean() < 5e-3, "%s: %.4f of the bytes differ" % (key, float((d > 0).mean()))
    assert np.array_equal(out["c2ws"].numpy(), f["c2ws"])


def test_condition_render_default_angle_weighting_on_a_sphere():
    """the production default of export_condition is trimesh's ANGLE-weighted vertex normals (fixture G9 only pins the
    area-weighted fallback).  Analytic check on a unit sphere: at every covered pixel the encoded normal must be the radial
    direction, i.e. the decoded world position / 0.95 (geometry_scale), within 2 uint8 steps + the tessellation error; and the
    per-pixel oracle (angle-weighted normals interpolated by the oracle rasteriser) must match within one truncation step."""
    from unitex_amd.texturetools.meshes import closed_sphere
    from unitex_amd.texturetools.video import VideoExporter, _vertex_normals
    v, f = closed_sphere(96, 48)
    out = VideoExporter(device="cuda:0").export_condition((v, f), geometry_scale=0.95, n_views=6, n_rows=2, n_cols=3, H=96, W=96,
                                                          fov_deg=49.1, scale=1.0, perspective=False, orbit=False, background="grey",
                                                          return_image=True, return_camera=True)
    alpha = np.asarray(out["alpha"]) > 0
    nrm = np.asarray(out["normal"]).astype(np.float64) / 255.0 * 2 - 1
    pos = np.asarray(out["ccm"]).astype(np.float64) / 255.0 * 2 - 1
    assert alpha.mean() > 0.5
    d = np.abs(nrm - pos / 0.95)[alpha]
    assert d.max() < 3.0 * 2 / 255 + 2e-3, d.max()
    # oracle of view 0 (front) with the same normals
    vs = v * 0.95
    c2ws = G.box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]
    mvp = G.mvp_matrices(c2ws, G.intrinsics(1.0, 1.0, fov=False), False)
    clip = G.transform_points(vs.astype(np.float32), mvp)
    vn = _vertex_normals(torch.from_numpy(vs.astype(np.float32)), torch.from_numpy(f), weighting="angle").numpy()
    r = G.rasterize(clip[0], f, 96, 96)
    n_i = G.interpolate(vn, r, f)
    n_i = n_i / np.maximum(np.linalg.norm(n_i, axis=-1, keepdims=True), 1e-12)
    a = (r[..., 3] > 0)
    exp = (np.clip(n_i * 0.5 + 0.5, 0, 1) * 255.0).astype(np.uint8)
    got = np.asarray(out["normal"])[:96, :96]
    assert np.array_equal(np.asarray(out["alpha"])[:96, :96] > 0, a)
    dd = np.abs(got.astype(np.int32) - exp.astype(np.int32))[a]
    assert dd.max() <= 1 and (dd > 0).mean() < 5e-3


@pytest.mark.parametrize("n_faces,n_views,HW,sample_mask", [(50000, 6, 1024, 31), (200000, 6, 1024, 31), (200000, 8, 2048, 63)])
def test_backprojection_at_baseline_config_sizes_sampled_bit_exact(n_faces, n_views, HW, sample_mask):
    """BASELINE.json configs[3] (50k-face mesh, six 1024^2 views) and configs[4] (200k-face mesh; six 1024^2 views, and the literal
    configs[4] geometry: EIGHT 2048^2 views -- camera.generate_views_c2ws(8), the builder-defined 8-view set) into a 2048^2 atlas.  The UV-space
    raster record is compared with the C oracle over the WHOLE atlas; LBVH node arrays likewise; the per-(view, texel) outputs of
    the fused gather + visibility kernel (colour, ray visibility, alpha) are compared bit for bit on a pseudo-random 1/32 sample of
    the texels -- the oracle walks only covered texels, so it is handed the raster with every other texel blanked (its result for
    a texel does not depend on any other texel)."""
    ops = _ops()
    T = 2048
    s = _scene(n_faces, T, HW, seed=n_faces)
    if n_views == 8:
        from unitex_amd.texturetools.camera import generate_views_c2ws
        s["c2ws"] = generate_views_c2ws(8, 2.8)[0].numpy().astype(np.float32)
        s["mvp"] = G.mvp_matrices(s["c2ws"], G.intrinsics(1.0, 1.0, fov=False), perspective=False)
    verts, faces, uvs, mvp = s["verts"], s["faces"], s["uvs"], s["mvp"]
    vd, fd = _cu(verts), _cu(faces)
    uvclip = np.concatenate([uvs * 2 - 1, np.zeros((len(uvs), 1), np.float32), np.ones((len(uvs), 1), np.float32)], -1)
    rast_d = ops.rasterize(_cu(uvclip), fd, T, T)
    rast_ref = G.rasterize(uvclip, faces, T, T)
    assert np.array_equal(rast_d.cpu().numpy(), rast_ref), "UV raster record (barycentrics + triangle id) at 2048^2"
    covered = rast_ref[..., 3] > 0
    assert covered.mean() > 0.5
    bvh_ref = G.BVH(verts, faces)
    bvh = ops.BVH(vd, fd)
    info, aabb, codes, idx = bvh.arrays()
    assert np.array_equal(info.cpu().numpy(), bvh_ref.info) and np.array_equal(aabb.cpu().numpy(), bvh_ref.aabb)
    assert np.array_equal(idx.cpu().numpy(), bvh_ref.order)
    clip, ndc = ops.transform_points(vd, _cu(mvp))
    dirs = (-s["c2ws"][:, :3, 2]).astype(np.float32)
    from unitex_amd.texturetools.benchmarks import smooth_views
    imgs = np.zeros((n_views, HW, HW, 4), np.float32)
    imgs[..., :3] = smooth_views(n_views, HW, HW)
    for v in range(n_views):
        imgs[v, ..., 3] = (ops.rasterize(clip[v].contiguous(), fd, HW, HW)[..., 3] > 0).float().cpu().numpy()
    fn = G.face_normals(verts, faces)
    col, rv, ao = ops.backproject(rast_d, vd, fd, _cu(fn), ndc.contiguous(), _cu(dirs), _cu(imgs), bvh, angle_deg=100.0)
    yy, xx = np.mgrid[0:T, 0:T]
    sample = (((yy * 2654435761 + xx * 40503) >> 7) & sample_mask) == 0
    rast_s = rast_ref.copy()
    rast_s[~sample] = 0.0
    col_ref, rv_ref, ao_ref = G.backproject(rast_s, verts, faces, fn, ndc.cpu().numpy(), dirs, imgs, bvh_ref, angle_deg=100.0)
    pick = sample & covered
    assert pick.sum() > 25000
    assert rv.shape[0] == n_views
    assert np.array_equal(rv.cpu().numpy()[:, pick], rv_ref[:, pick]), "ray visibility"
    assert np.array_equal(ao.cpu().numpy()[:, pick], ao_ref[:, pick]), "alpha mask"
    assert np.array_equal(col.cpu().numpy()[:, pick], col_ref[:, pick]), "gathered colours"
    assert 0.05 < rv_ref[:, pick].mean() < 0.9


def _helicoid(turns=2.0, n_r=12, n_t=160, pitch=0.25):
    """spiral ramp: two turns lie on top of each other in the xy projection while every normal faces +z -> ONE connected
    same-bucket component that folds over itself."""
    r = np.linspace(0.3, 1.0, n_r); t = np.linspace(0.0, 2 * np.pi * turns, n_t)
    R, Tt = np.meshgrid(r, t, indexing="ij")
    v = np.stack([R * np.cos(Tt), R * np.sin(Tt), pitch * Tt / (2 * np.pi)], -1).reshape(-1, 3).astype(np.float32)
    f = []
    for i in range(n_r - 1):
        for j in range(n_t - 1):
            a, b, c, d = i * n_t + j, i * n_t + j + 1, (i + 1) * n_t + j, (i + 1) * n_t + j + 1
            f += [[a, c, b], [b, c, d]]
    return v, np.asarray(f, np.int32)


def test_chart_unwrap_labels_and_bijectivity():
    """the chart unwrap of UV-less meshes (meshes.unwrap_charts; reference: UVAtlas through open3d, uv_atlas.py:171-175 [3p]):
    (i) the GPU chart labelling (utx_chart_flood) equals connected components computed on the CPU, label = smallest face index;
    (ii) rasterise-and-count with the ORACLE rasteriser: no face of visible size loses its texels to another face, also on a
    spiral ramp whose single same-bucket component folds over itself in projection (split into layers by the unwrap);
    (iii) few charts: the bumpy sphere unwraps into a handful, not one chart per triangle."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from unitex_amd.texturetools import meshes
    ops = _ops()
    for name, (v, f) in {"sphere": meshes.closed_sphere(96, 48), "bumpy": sphere_with_faces(8000)[:2], "helicoid": _helicoid()}.items():
        adj = meshes.face_adjacency(f)
        dirs = meshes.projection_frames(26)[0]
        bucket = meshes.chart_buckets(v, f, adj, dirs=dirs)
        chart = ops.chart_flood(_cu(adj), _cu(bucket)).cpu().numpy()
        rows, cols = np.nonzero(adj >= 0)
        nb = adj[rows, cols]
        keep = bucket[rows] == bucket[nb]
        ncomp, lab = connected_components(coo_matrix((np.ones(keep.sum()), (rows[keep], nb[keep])), shape=(len(f), len(f))), directed=False)
        mins = np.full(ncomp, len(f)); np.minimum.at(mins, lab, np.arange(len(f)))
        assert np.array_equal(chart, mins[lab]), "%s: chart labels" % name
        T = 512
        vv, ff, uu, fu = meshes.unwrap_charts(v, f, atlas=T, gutter=3.0)
        assert np.array_equal(vv, v.astype(np.float32)) and np.array_equal(ff, f) and uu.min() >= 0 and uu.max() <= 1
        uvclip = np.concatenate([uu * 2 - 1, np.zeros((len(uu), 1), np.float32), np.ones((len(uu), 1), np.float32)], -1)
        # no texel centre is claimed by two faces: the ORACLE rasteriser with the faces in order and reversed (all triangles at z = 0: the smaller id
        # wins a shared texel, the other one in the reversed pass) must name the same face everywhere
        ids = G.rasterize(uvclip, fu, T, T)[..., 3].astype(np.int64)
        idr = G.rasterize(uvclip, np.ascontiguousarray(fu[::-1]), T, T)[..., 3].astype(np.int64)
        idr = np.where(idr > 0, len(f) + 1 - idr, idr)
        assert np.array_equal(ids, idr), "%s: %d texels are claimed by two faces" % (name, int((ids != idr).sum()))
        owned = np.bincount(ids.reshape(-1), minlength=len(f) + 1)[1:]
        t = uu.reshape(-1, 3, 2).astype(np.float64)
        area = np.abs((t[:, 1, 0] - t[:, 0, 0]) * (t[:, 2, 1] - t[:, 0, 1]) - (t[:, 2, 0] - t[:, 0, 0]) * (t[:, 1, 1] - t[:, 0, 1])) * 0.5 * T * T
        big = area >= 4.0
        assert big.mean() > 0.4, name
        assert abs(owned.sum() - area.sum()) < 0.02 * area.sum(), "%s: texels owned %.0f vs UV area %.0f" % (name, owned.sum(), area.sum())
        # UV triangles keep their orientation (no mirrored charts)
        sgn = (t[:, 1, 0] - t[:, 0, 0]) * (t[:, 2, 1] - t[:, 0, 1]) - (t[:, 2, 0] - t[:, 0, 0]) * (t[:, 1, 1] - t[:, 0, 1])
        assert (sgn[big] > 0).all(), name
        if name == "sphere":
            assert len(np.unique(chart)) <= 60          # 26 projection directions -> 26 caps / bands / corners (+ a few islands), not one chart per triangle
        # STRETCH BOUND (the reference's UVAtlas max_stretch = 0.1667, uv_atlas.py:171): one texel density for the whole atlas, so a face's 3-D area per
        # UV area relative to the least stretched face is its area stretch under the planar projection -- <= 1/6 for every face of visible size
        p3 = v.astype(np.float64)[f]
        a3 = 0.5 * np.linalg.norm(np.cross(p3[:, 1] - p3[:, 0], p3[:, 2] - p3[:, 0]), axis=1)
        ratio = a3[big] / area[big]
        stretch = ratio / ratio.min() - 1.0
        assert stretch.max() <= 1.0 / 6.0 + 1e-6, "%s: area stretch %.3f above the 1/6 bound" % (name, stretch.max())


def test_blank_mesh_chart_unwrap_feeds_the_inverse_renderer(tmp_path):
    """UV-less input through the pipeline's mesh stage (prepare_blank_mesh(unwrap='charts') -> processed_mesh.obj -> load_mesh) into
    NVDiffRendererInverse.infer: runs, the seam mask (reference renderer_inverse.py:603-605) stays a small fraction of the covered
    texels, and the charts fill much more of the atlas than the per-triangle grid of round 1 (which spends most texels on gutters)."""
    from unitex_amd.texturetools import camera, meshes
    from unitex_amd.texturetools.benchmarks import smooth_views
    from unitex_amd.texturetools.renderer_inverse import NVDiffRendererInverse
    v, f = meshes.closed_sphere(64, 32)
    src = str(tmp_path / "blank.obj")
    meshes.save_obj(src, v * 2.0, f)
    frac = {}
    for how in ("charts", "grid"):
        vv, ff, uu, fu = meshes.prepare_blank_mesh(src, min_faces=3000, max_faces=20000, scale=0.95, atlas=512, gutter=3.0, unwrap=how)
        out = str(tmp_path / ("processed_%s.obj" % how))
        meshes.save_obj(out, vv, ff, uu, faces_uv=fu)
        inv = NVDiffRendererInverse(device="cuda:0").update_from_file(out)
        c2ws, order = camera.generate_views_c2ws(6, 2.8)
        res = inv.infer(None, c2ws=c2ws, intrinsics=camera.generate_intrinsics(1.0, 1.0, fov=False),
                        image_attrs=torch.from_numpy(smooth_views(6, 128, 128)).cuda(), perspective=False, H=128, W=128, H2D=512, W2D=512,
                        filt_gradient_points=False, ray_normal_angle_threhold=100.0)
        covered = res[2][0, ..., 0]
        frac[how] = ((inv.last["seam"].bool() & covered).float().sum().item() / covered.float().sum().item(), covered.float().mean().item())
        assert res[0].texture.shape == (512, 512, 3)
    # seams (texels whose composite winner differs from a neighbour's) stay a small fraction; the charts use the atlas area far better
    # than one gutter-separated slot per triangle (texel density = texture resolution on the surface)
    assert frac["charts"][0] < 0.25 and frac["charts"][1] > 0.3 and frac["charts"][1] > 1.3 * frac["grid"][1], frac
