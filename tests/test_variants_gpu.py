"""GPU parity tests (through the C ABI) of the non-default back-projection variants:
exact k-NN gather, the gradient / facing filter of the view masks, infer(method='kdtree') in its three flavours, and the
query-field (LTM) hook.  Checker: oracle/geom_ref.py (pinned on these variants by fixture G11, tests/test_golden_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import geom_ref as G
from unitex_amd.texturetools.meshes import sphere_with_faces

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ops():
    from unitex_amd.texturetools import ops
    return ops


def _cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


@pytest.mark.parametrize("N,M,k", [(20000, 4000, 1), (20000, 4000, 4), (6000, 3000, 32), (20, 50, 32)])
def test_knn_gather_bit_exact_neighbours(N, M, k):
    """neighbour ids (float32 distance, index tie rule) identical to the brute-force oracle; means to 1 ulp-ish."""
    ops = _ops()
    rng = np.random.default_rng(N + k)
    src = rng.uniform(-0.95, 0.95, (N, 3)).astype(np.float32)
    src[: N // 4] *= 0.1                                   # a dense cluster: many points per grid cell
    dst = rng.uniform(-1.0, 1.0, (M, 3)).astype(np.float32)
    attr = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    smask = rng.uniform(size=N) < 0.7
    dmask = rng.uniform(size=M) < 0.8
    idx_ref, d2_ref = G.knn_brute(src, dst, k, smask)
    out_ref = G.knn_gather(src, attr, dst, k, src_mask=smask, dst_mask=dmask, out=np.full((M, 3), -7.0, np.float32))
    out = torch.full((M, 3), -7.0, dtype=torch.float32, device="cuda")
    out, idx, d2 = ops.knn_gather(_cu(src), _cu(dst), k, src_attr=_cu(attr), src_mask=_cu(smask.astype(np.uint8)),
                                  dst_mask=_cu(dmask.astype(np.uint8)), out=out, want_index=True)
    idx, d2, out = idx.cpu().numpy(), d2.cpu().numpy(), out.cpu().numpy()
    assert np.array_equal(idx[dmask], idx_ref[dmask]), "neighbour ids"
    assert np.array_equal(d2[dmask], d2_ref[dmask]), "squared distances"
    assert np.array_equal(out[~dmask], out_ref[~dmask]), "untouched rows"
    assert np.abs(out[dmask] - out_ref[dmask]).max() < 1e-6


def test_knn_mvpaint_weighting_matches_oracle():
    ops = _ops()
    rng = np.random.default_rng(5)
    N, M, k = 8000, 3000, 8
    src = rng.uniform(-0.9, 0.9, (N, 3)).astype(np.float32); dst = rng.uniform(-0.9, 0.9, (M, 3)).astype(np.float32)
    attr = rng.uniform(0, 1, (N, 9)).astype(np.float32)
    sn = rng.normal(size=(N, 3)).astype(np.float32); dn = rng.normal(size=(M, 3)).astype(np.float32)
    ref = G.knn_gather(src, attr, dst, k, mode="mvpaint", src_nrm=sn, dst_nrm=dn)
    out = ops.knn_gather(_cu(src), _cu(dst), k, src_attr=_cu(attr), mode="mvpaint", src_nrm=_cu(sn), dst_nrm=_cu(dn)).cpu().numpy()
    # sum(w) can be close to zero (cosines of both signs): compare where the oracle's result is well conditioned
    ok = np.abs(ref).max(-1) < 50
    assert ok.mean() > 0.9
    assert np.abs(out[ok] - ref[ok]).max() < 2e-3 and np.median(np.abs(out[ok] - ref[ok])) < 1e-6


def test_view_visibility_gradient_filter_bit_exact():
    ops = _ops()
    verts, faces, uvs = sphere_with_faces(3000)
    c2ws = G.box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]
    mvp = G.mvp_matrices(c2ws, G.intrinsics(1.0, 1.0, fov=False), perspective=False)
    HW = 96
    clip = G.transform_points(verts, mvp)
    fn = G.face_normals(verts, faces)
    va = np.concatenate([verts, G.vertex_normals_area(verts, faces)], -1).astype(np.float32)
    rast = np.stack([G.rasterize(clip[v], faces, HW, HW) for v in range(6)])
    attr = np.stack([G.interpolate(va, rast[v], faces) for v in range(6)])
    dirs = (-c2ws[:, :3, 2]).astype(np.float32)
    dirs = (dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32)
    for thr, ang in ((0.20, 115.0), (0.15, 100.0), (0.5, 95.0)):
        ref = G.view_visibility(attr, rast, fn, dirs, grad_thr=thr, angle_deg=ang)
        vis, alpha = ops.view_visibility(_cu(attr), _cu(rast), _cu(fn), _cu(dirs), grad_thr=thr, angle_deg=ang)
        assert np.array_equal(vis.cpu().numpy().astype(bool), ref), "filtered view mask (thr %g, angle %g)" % (thr, ang)
        assert np.array_equal(alpha.cpu().numpy(), ref.astype(np.float32))
        assert 0.02 < ref.mean() < 0.9


def _g11():
    f = np.load(os.path.join(GOLD, "g11_kdtree_and_filter.npz"))
    return f


def _renderer(f):
    from unitex_amd.texturetools.renderer_inverse import NVDiffRendererInverse
    inv = NVDiffRendererInverse(device="cuda")
    inv.update_from_arrays(f["verts"], f["faces"], f["uvs"])
    return inv


@pytest.mark.parametrize("name,kw", [("order_mean", dict(kdtree_method="order_mean", kdtree_n_neighbors_visiable=1, kdtree_n_neighbors_invisiable=4)),
                                     ("mean", dict(kdtree_method="mean", kdtree_n_neighbors=4)),
                                     ("mvpaint", dict(kdtree_method="mvpaint", kdtree_n_neighbors=4))])
def test_infer_kdtree_variants_match_reference_fixture(name, kw):
    """product infer(method='kdtree', filt_gradient_points=True) on the inputs of fixture G11 against the atlases the
    reference itself produced (its rasteriser / ray tracer / kd-tree seams stubbed by the oracle when the fixture was made)."""
    f = _g11()
    inv = _renderer(f)
    T = 96
    imgs = torch.from_numpy(f["images"].astype(np.float32))
    out = inv.infer(None, c2ws=torch.from_numpy(f["c2ws"]), intrinsics=torch.from_numpy(f["intr"]), image_attrs=imgs, perspective=False,
                    H=96, W=96, H2D=T, W2D=T, method="kdtree", grad_norm_threhold=0.20, ray_normal_angle_threhold=115.0,
                    filt_gradient_points=True, **kw)
    n = imgs.shape[0]
    ref_vis = np.unpackbits(f["mask_visiable"])[: n * 96 * 96].reshape(n, 96, 96).astype(bool)
    got_vis = inv.last["view_mask"].cpu().numpy().astype(bool)
    assert int((got_vis != ref_vis).sum()) <= 8, "filtered view masks vs the reference"
    ref_v2d = np.unpackbits(f["mask_2d_visiable"])[: n * T * T].reshape(n, T, T).astype(bool)
    assert int((out[1][..., 0].cpu().numpy() != ref_v2d).sum()) <= 24, "texel visibility vs the reference"
    err = np.abs(out[3][0].cpu().numpy() - f["color_2d_" + name][0])
    # a handful of knife-edge mask pixels move a handful of texels to another source; everything else agrees to float noise
    assert (err > 1e-3).mean() < 2e-2 and np.median(err) < 1e-6, "%s: max %g frac %g" % (name, err.max(), (err > 1e-3).mean())


def test_query_field_hook_replaces_the_nearest_neighbour_fill():
    """reproject_inpainting=True hands (seen positions, seen colours, unseen positions) to the registered field; a field that
    answers with the exact nearest neighbour must reproduce the default path bit for bit."""
    f = _g11()
    inv = _renderer(f)
    imgs = torch.from_numpy(f["images"].astype(np.float32))
    kw = dict(c2ws=torch.from_numpy(f["c2ws"]), intrinsics=torch.from_numpy(f["intr"]), image_attrs=imgs, perspective=False,
              H=96, W=96, H2D=96, W2D=96, method="reproject", ray_normal_angle_threhold=100.0, filt_gradient_points=True)
    base = inv.infer(None, **kw)[3].cpu().numpy()
    calls = []

    def field(pv, cv, pi):
        calls.append((pv.shape, cv.shape, pi.shape))
        idx, _ = G.knn_brute(pv.cpu().numpy(), pi.cpu().numpy(), 1)
        return torch.from_numpy(cv.cpu().numpy()[idx[:, 0]]).to(pi.device)
    with pytest.raises(NotImplementedError):
        inv.infer(None, reproject_inpainting=True, **kw)
    inv.register_query_field(field)
    hooked = inv.infer(None, reproject_inpainting=True, **kw)[3].cpu().numpy()
    assert len(calls) == 1 and calls[0][0][0] == calls[0][1][0] and calls[0][2][0] > 0
    assert np.array_equal(hooked, base)
    inv.clear()
    assert inv.query_field_function is None
