"""Edge cases and error behaviour through the C ABI (GPU): invalid descriptors must come back as a negative return
code + utx_last_error message (the Python shim raises RuntimeError), never as a fault; degenerate sizes must work."""
import numpy as np
import pytest
import torch

from oracle import dit_ref
from oracle import geom_ref as G

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_gemm_rejects_invalid_descriptors():
    from unitex_amd.flux import ops
    A = torch.zeros(128, 96, dtype=BF, device="cuda")       # K = 96 is not a multiple of 64
    B = torch.zeros(128, 96, dtype=BF, device="cuda")
    with pytest.raises(RuntimeError, match="utx_gemm_bf16"):
        ops.gemm(A, B)
    A = torch.zeros(128, 128, dtype=BF, device="cuda")
    B = torch.zeros(60, 128, dtype=BF, device="cuda")       # N = 60 is not a multiple of 8
    with pytest.raises(RuntimeError, match="utx_gemm_bf16"):
        ops.gemm(A, B)
    with pytest.raises(RuntimeError, match="utx_gemm_bf16"):   # gate without residual
        d = ops.make_gemm_desc(A, A, torch.empty(128, 128, dtype=BF, device="cuda"), gate=torch.ones(128, dtype=BF, device="cuda"), res=A)
        d.res = None
        ctx = ops.get_ctx(0)
        import ctypes as C
        ctx.check(ctx.lib.utx_gemm_bf16(ctx.handle, C.byref(d), ctx.stream()))


def test_attention_degenerate_sizes_and_errors():
    from unitex_amd.flux import ops
    g = torch.Generator().manual_seed(0)
    for S in (1, 33, 64, 65):
        q = torch.randn(2, S, 128, generator=g).to(BF)
        k = torch.randn(2, S, 128, generator=g).to(BF)
        v = torch.randn(2, S, 128, generator=g).to(BF)
        S_pad = (S + 63) // 64 * 64
        Qh = torch.zeros(2, S_pad, 128, dtype=BF, device="cuda"); Qh[:, :S] = q.cuda()
        Kh = torch.zeros(2, S_pad, 128, dtype=BF, device="cuda"); Kh[:, :S] = k.cuda()
        Vt = torch.zeros(2, 128, S_pad, dtype=BF, device="cuda"); Vt[:, :, :S] = v.cuda().transpose(1, 2)
        out = ops.attention(Qh, Kh, Vt, S=S).float().cpu().view(S, 2, 128).permute(1, 0, 2)
        ref = dit_ref.sdpa(q.float(), k.float(), v.float(), em=False)
        assert (out - ref).abs().max().item() < 3e-2, S
    with pytest.raises(RuntimeError, match="utx_attn_fwd_bf16"):
        ops.attention(Qh, Kh, Vt, S=0)


def test_vae_primitives_reject_bad_shapes():
    from unitex_amd.flux import ops
    from unitex_amd._lib import ptr
    ctx = ops.get_ctx(0)
    x = torch.zeros(64, 96, dtype=BF, device="cuda")
    w = torch.zeros(int(ctx.lib.utx_group_norm_workspace_bytes()), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="utx_group_norm"):      # C = 96 is not a multiple of 128
        ctx.check(ctx.lib.utx_group_norm(ctx.handle, ptr(x), 64, 96, ptr(x), ptr(x), 1e-6, 0, ptr(x), ptr(w), ctx.stream()))
    with pytest.raises(RuntimeError, match="utx_softmax_rows"):    # ncol not a multiple of 8
        ctx.check(ctx.lib.utx_softmax_rows(ctx.handle, ptr(x), 4, 96, 90, ctx.stream()))
    with pytest.raises(RuntimeError, match="utx_conv3x3_thin"):    # Cin = 5 has no thin kernel
        ctx.check(ctx.lib.utx_conv3x3_thin(ctx.handle, ptr(x), 4, 4, 5, ptr(x), ptr(x), 8, ptr(x), ctx.stream()))


def test_raster_degenerate_geometry():
    from unitex_amd.texturetools import ops as gops
    # one zero-area triangle, one triangle fully outside the frustum, one behind the camera (w < 0), one valid
    pos = np.array([[0.1, 0.1, 0, 1], [0.1, 0.1, 0, 1], [0.1, 0.1, 0, 1],
                    [3, 3, 0, 1], [4, 3, 0, 1], [3, 4, 0, 1],
                    [-0.5, -0.5, 0, -1], [0.5, -0.5, 0, -1], [0, 0.5, 0, -1],
                    [-0.5, -0.5, 0.2, 1], [0.5, -0.5, 0.2, 1], [0, 0.5, 0.2, 1]], dtype=np.float32)
    tri = np.arange(12, dtype=np.int32).reshape(4, 3)
    r = gops.rasterize(torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), 32, 32).cpu().numpy()
    ref = G.rasterize(pos, tri, 32, 32)
    assert np.array_equal(r, ref)
    ids = np.unique(r[..., 3])
    assert set(ids.tolist()) == {0.0, 4.0}


def test_backprojection_with_nothing_visible_still_fills():
    """all six views see nothing (alpha 0 everywhere): every texel is unseen -> the atlas stays black, no fault."""
    from unitex_amd.texturetools import camera, meshes
    from unitex_amd.texturetools.renderer_inverse import NVDiffRendererInverse
    v, f, uv = meshes.sphere_with_faces(800)
    inv = NVDiffRendererInverse(device="cuda:0").update_from_arrays(v * 1e-3 + 5.0, f, uv)     # far off-screen, tiny
    c2ws = camera.generate_box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]
    intr = camera.generate_intrinsics(1.0, 1.0, fov=False)
    imgs = torch.rand(6, 64, 64, 3, device="cuda:0")
    out = inv.infer(None, c2ws=c2ws, intrinsics=intr, image_attrs=imgs, perspective=False, H=64, W=64, H2D=128, W2D=128,
                    filt_gradient_points=False, ray_normal_angle_threhold=100.0)
    tex = out[0].texture
    assert tex.shape == (128, 128, 3) and int(tex.max()) == 0
    assert not bool(out[1].any())
