"""Functional wrappers over the geometry half of the C ABI (tensor in / tensor out, all on the GPU).
Used by the parity tests and by NVDiffRendererInverse / VideoExporter."""
import ctypes as C
import math

import numpy as np
import torch

from .._lib import BackprojectDesc, ptr
from ..flux.ops import get_ctx

F32, I32, U8 = torch.float32, torch.int32, torch.uint8


def _f(t):
    assert t.is_cuda and t.dtype == F32 and t.is_contiguous(), "expected a contiguous CUDA float32 tensor"
    return t


def _i(t):
    assert t.is_cuda and t.dtype == I32 and t.is_contiguous(), "expected a contiguous CUDA int32 tensor"
    return t


def transform_points(verts, mvp, want_ndc=True):
    """verts [V,3], mvp [n,4,4] -> clip [n,V,4], ndc [n,V,2]"""
    ctx = get_ctx(verts.device.index)
    V, n = verts.shape[0], mvp.shape[0]
    clip = torch.empty(n, V, 4, dtype=F32, device=verts.device)
    ndc = torch.empty(n, V, 2, dtype=F32, device=verts.device) if want_ndc else None
    ctx.check(ctx.lib.utx_transform_points(ctx.handle, ptr(_f(verts)), V, ptr(_f(mvp)), n, ptr(clip), ptr(ndc), ctx.stream()))
    return clip, ndc


def rasterize(pos_clip, tri, H, W):
    """pos_clip [V,4], tri [F,3] int32 -> rast [H,W,4] = (u, v, z/w, id+1)"""
    ctx = get_ctx(pos_clip.device.index)
    F = tri.shape[0]
    rast = torch.empty(H, W, 4, dtype=F32, device=pos_clip.device)
    wb = ctx.lib.utx_rasterize_workspace_bytes(F, H, W)
    work = torch.empty(wb, dtype=U8, device=pos_clip.device)
    ctx.check(ctx.lib.utx_rasterize(ctx.handle, ptr(_f(pos_clip)), ptr(_i(tri)), F, H, W, ptr(rast), ptr(work), ctx.stream()))
    return rast


def chart_flood(adj, bucket):
    """adj [F,3] int32 (face across each edge, -1 = border), bucket [F] int32 -> chart [F] int32: smallest face index of the
    connected same-bucket component (utx_chart_flood; synchronises the stream)."""
    ctx = get_ctx(adj.device.index)
    F = adj.shape[0]
    chart = torch.empty(F, dtype=torch.int32, device=adj.device)
    flag = torch.zeros(1, dtype=torch.int32, device=adj.device)
    rc = ctx.lib.utx_chart_flood(ctx.handle, ptr(_i(adj)), ptr(_i(bucket)), F, ptr(chart), ptr(flag), ctx.stream())
    if rc < 0:
        ctx.check(rc)
    return chart


def interpolate(attr, rast, tri):
    ctx = get_ctx(attr.device.index)
    H, W = rast.shape[:2]
    Cc = attr.shape[1]
    out = torch.empty(H, W, Cc, dtype=F32, device=attr.device)
    ctx.check(ctx.lib.utx_interpolate(ctx.handle, ptr(_f(attr)), Cc, ptr(_f(rast)), ptr(_i(tri)), H * W, ptr(out), ctx.stream()))
    return out


def condition_shade(rast, nrm, pos, bg):
    """rast [N,H,W,4], nrm/pos [N,H,W,3] -> uint8 normal, ccm [N,H,W,3], alpha [N,H,W]"""
    ctx = get_ctx(rast.device.index)
    npix = rast.numel() // 4
    on = torch.empty(rast.shape[:-1] + (3,), dtype=U8, device=rast.device)
    oc = torch.empty_like(on)
    oa = torch.empty(rast.shape[:-1], dtype=U8, device=rast.device)
    arr = (C.c_float * 3)(*[float(x) for x in bg])
    ctx.check(ctx.lib.utx_condition_shade(ctx.handle, ptr(_f(rast)), ptr(_f(nrm)), ptr(_f(pos)), arr, npix, ptr(on), ptr(oc), ptr(oa), ctx.stream()))
    return on, oc, oa


def face_normals(verts, faces):
    ctx = get_ctx(verts.device.index)
    out = torch.empty(faces.shape[0], 3, dtype=F32, device=verts.device)
    ctx.check(ctx.lib.utx_face_normals(ctx.handle, ptr(_f(verts)), ptr(_i(faces)), faces.shape[0], ptr(out), ctx.stream()))
    return out


def texture_shade(rast, uv01, tri, tex, bg=(1.0, 1.0, 1.0)):
    """rast [H,W,4], uv01 [V,2], tex [Ht,Wt,3] fp32 (UV-raster orientation) -> uint8 RGB [H,W,3]."""
    ctx = get_ctx(rast.device.index)
    H, W = rast.shape[:2]
    out = torch.empty(H, W, 3, dtype=U8, device=rast.device)
    bgv = (C.c_float * 3)(*[float(b) for b in bg])
    ctx.check(ctx.lib.utx_texture_shade(ctx.handle, ptr(_f(rast)), ptr(_f(uv01)), ptr(_i(tri)), ptr(_f(tex)), tex.shape[0], tex.shape[1],
                                        bgv, H * W, ptr(out), ctx.stream()))
    return out


class BVH:
    """utx_bvh handle (RayTracing / APRMISRayTracing of the reference)."""

    def __init__(self, verts, faces):
        self.ctx = get_ctx(verts.device.index)
        self.verts, self.faces = _f(verts), _i(faces)   # kept alive: the handle borrows them
        h = C.c_void_p()
        # every array of the tree in ONE torch allocation (caching allocator: no hipMalloc / hipFree on the path), the build itself enqueued without a host wait (the wait moved one launch later: the first launch that needs the tree's depth blocks on the build's event once per tree)
        nbytes = int(self.ctx.lib.utx_bvh_workspace_bytes(faces.shape[0]))
        self.work = torch.empty(nbytes + 256, dtype=torch.uint8, device=verts.device)
        base = (self.work.data_ptr() + 255) & ~255
        self.ctx.check(self.ctx.lib.utx_bvh_build_ws(self.ctx.handle, ptr(self.verts), verts.shape[0], ptr(self.faces), faces.shape[0],
                                                     C.c_void_p(base), C.c_size_t(nbytes), C.byref(h), self.ctx.stream()))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.utx_bvh_free(self.handle)      # the handle only: the arrays are self.work, which torch releases stream-ordered like any other tensor
                self.handle = None
        except Exception:
            pass

    def arrays(self):
        """copies of (info [2F-1,3], aabb [2F-1,6], sorted codes [F], sorted ids [F]) as torch tensors"""
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        F = self.ctx.lib.utx_bvh_arrays(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        torch.cuda.synchronize()
        dev = self.verts.device

        def view(p, n, dtype):
            t = torch.empty(n, dtype=dtype, device=dev)
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), p, ctypes.c_size_t(n * t.element_size()), 3)
            return t
        info = view(a, (2 * F - 1) * 3, I32).view(-1, 3)
        aabb = view(b, (2 * F - 1) * 6, F32).view(-1, 6)
        codes = view(c, F, I32)
        idx = view(d, F, I32)
        return info, aabb, codes, idx

    def trace(self, rays_o, rays_d):
        ro, rd = _f(rays_o.reshape(-1, 3)), _f(rays_d.reshape(-1, 3))
        tid = torch.empty(ro.shape[0], dtype=I32, device=ro.device)
        self.ctx.check(self.ctx.lib.utx_bvh_trace(self.ctx.handle, self.handle, ptr(ro), ptr(rd), ro.shape[0], ptr(tid), self.ctx.stream()))
        return tid

    def trace_count(self, rays_o, rays_d):
        """(tid, number of tree nodes the rays visited in total) -- the measurement behind bench.py's nodes_visited_per_ray"""
        ro, rd = _f(rays_o.reshape(-1, 3)), _f(rays_d.reshape(-1, 3))
        tid = torch.empty(ro.shape[0], dtype=I32, device=ro.device)
        visited = torch.zeros(1, dtype=torch.int64, device=ro.device)
        self.ctx.check(self.ctx.lib.utx_bvh_trace_count(self.ctx.handle, self.handle, ptr(ro), ptr(rd), ro.shape[0], ptr(tid), ptr(visited),
                                                        self.ctx.stream()))
        return tid, int(visited.item())

    def depth(self):
        """longest root-to-leaf path of the tree (<= 60: the stackless packed traversal is in use)"""
        return int(self.ctx.lib.utx_bvh_depth(self.handle))


def backproject(rast2d, verts, faces, fnormal, vndc, dirs, images, bvh, angle_deg=100.0, view_begin=0, view_count=None,
                out=None):
    """fused per-(view, texel) gather + visibility.  Returns color [n,Th,Tw,3], rayvis [n,Th,Tw] u8, alphaok u8."""
    ctx = get_ctx(rast2d.device.index)
    Th, Tw = rast2d.shape[:2]
    n, H, W = images.shape[:3]
    dev = rast2d.device
    if out is None:
        color = torch.zeros(n, Th, Tw, 3, dtype=F32, device=dev)
        rayvis = torch.zeros(n, Th, Tw, dtype=U8, device=dev)
        alphaok = torch.zeros(n, Th, Tw, dtype=U8, device=dev)
    else:
        color, rayvis, alphaok = out
    d = BackprojectDesc()
    d.rast2d, d.verts, d.faces, d.fnormal = ptr(_f(rast2d)), ptr(_f(verts)), ptr(_i(faces)), ptr(_f(fnormal))
    d.vndc, d.dirs, d.images = ptr(_f(vndc)), ptr(_f(dirs)), ptr(_f(images))
    d.color, d.rayvis, d.alphaok = ptr(color), ptr(rayvis), ptr(alphaok)
    d.T_h, d.T_w, d.V, d.n_views, d.H, d.W = Th, Tw, verts.shape[0], n, H, W
    d.view_begin, d.view_count = view_begin, (n - view_begin if view_count is None else view_count)
    d.cos_thresh = float(np.float32(math.cos(math.radians(angle_deg))))
    d.two_sqrt3 = float(np.float32(2.0 * math.sqrt(3.0)))
    ctx.check(ctx.lib.utx_backproject(ctx.handle, C.byref(d), bvh.handle, ctx.stream()))
    return color, rayvis, alphaok


def dilate_visibility(rayvis, alphaok, rast2d):
    ctx = get_ctx(rayvis.device.index)
    n, H, W = rayvis.shape
    tmp = torch.empty_like(rayvis)
    out = torch.empty_like(rayvis)
    ctx.check(ctx.lib.utx_dilate_visibility(ctx.handle, ptr(rayvis), ptr(alphaok), ptr(_f(rast2d)), n, H, W, ptr(tmp), ptr(out), ctx.stream()))
    return out


def composite(colors, vis, order):
    ctx = get_ctx(colors.device.index)
    n, H, W = vis.shape
    atlas = torch.empty(H, W, 3, dtype=F32, device=colors.device)
    winner = torch.empty(H, W, dtype=torch.int8, device=colors.device)
    arr = (C.c_int * len(order))(*order)
    ctx.check(ctx.lib.utx_composite(ctx.handle, ptr(_f(colors)), ptr(vis), arr, len(order), H * W, ptr(atlas), ptr(winner), ctx.stream()))
    return atlas, winner


def seam_mask(winner, rast2d):
    ctx = get_ctx(winner.device.index)
    H, W = winner.shape
    tmp = torch.empty(H, W, dtype=U8, device=winner.device)
    seam = torch.empty(H, W, dtype=U8, device=winner.device)
    ctx.check(ctx.lib.utx_seam_mask(ctx.handle, ptr(winner), ptr(_f(rast2d)), H, W, ptr(tmp), ptr(seam), ctx.stream()))
    return seam


def view_visibility(attr6, rast, fnormal, dirs, grad_thr=0.20, angle_deg=115.0, radius=15):
    """mv_to_pcd's filt_gradient_points=True visibility (renderer_inverse.py:189-209): attr6 [n,H,W,6] interpolated
    (position, vertex normal), rast [n,H,W,4], dirs [n,3] ray directions -> (vis u8 [n,H,W], alpha f32 [n,H,W])."""
    ctx = get_ctx(rast.device.index)
    n, H, W, _ = rast.shape
    tmp = torch.empty(2 * n * H * W, dtype=U8, device=rast.device)
    vis = torch.empty(n, H, W, dtype=U8, device=rast.device)
    alpha = torch.empty(n, H, W, dtype=F32, device=rast.device)
    cos_thr = float(np.float32(math.cos(math.radians(angle_deg))))
    ctx.check(ctx.lib.utx_view_visibility(ctx.handle, ptr(_f(attr6)), ptr(_f(rast)), ptr(_f(fnormal)), ptr(_f(dirs)), n, H, W,
                                          float(grad_thr), cos_thr, int(radius), ptr(tmp), ptr(vis), ptr(alpha), ctx.stream()))
    return vis, alpha


def knn_gather(src_pos, dst_pos, k, src_attr=None, src_mask=None, dst_mask=None, out=None, mode="mean", src_nrm=None, dst_nrm=None,
               want_index=False):
    """exact k-NN in 3-D over dense, byte-masked point sets (bake_mv_to_uv_kdtree, renderer_inverse.py:367-433).
    src_pos [N,3], dst_pos [M,3]; src_attr [N,C] -> out [M,C] (written only where dst_mask): mean of the k neighbours'
    attributes, or the MVPaint weighting (mode='mvpaint').  want_index: also return (idx [M,k] i32, d2 [M,k] f32)."""
    from .._lib import KnnDesc
    ctx = get_ctx(src_pos.device.index)
    dev = src_pos.device
    src_pos, dst_pos = _f(src_pos).reshape(-1, 3), _f(dst_pos).reshape(-1, 3)
    N, M = src_pos.shape[0], dst_pos.shape[0]
    d = KnnDesc()
    keep = [src_pos, dst_pos]
    d.src_pos, d.dst_pos, d.N, d.M, d.k = ptr(src_pos), ptr(dst_pos), N, M, int(k)
    d.mode = {"mean": 0, "mvpaint": 1}[mode]
    if src_attr is not None:
        src_attr = _f(src_attr).reshape(N, -1)
        d.C = src_attr.shape[1]
        if out is None:
            out = torch.zeros(M, d.C, dtype=F32, device=dev)
        assert out.is_contiguous() and out.dtype == F32 and out.numel() == M * d.C
        d.src_attr, d.out_attr = ptr(src_attr), ptr(out)
        keep += [src_attr, out]
    for name, t in (("src_mask", src_mask), ("dst_mask", dst_mask)):
        if t is not None:
            t = t.reshape(-1).to(U8).contiguous()
            keep.append(t)
            setattr(d, name, ptr(t))
    for name, t in (("src_nrm", src_nrm), ("dst_nrm", dst_nrm)):
        if t is not None:
            t = _f(t).reshape(-1, 3)
            keep.append(t)
            setattr(d, name, ptr(t))
    idx = d2 = None
    if want_index:
        idx = torch.empty(M, int(k), dtype=I32, device=dev)
        d2 = torch.empty(M, int(k), dtype=F32, device=dev)
        d.out_idx, d.out_d2 = ptr(idx), ptr(d2)
    wb = ctx.lib.utx_knn_workspace_bytes(N)
    work = torch.empty(wb, dtype=U8, device=dev)
    ctx.check(ctx.lib.utx_knn(ctx.handle, C.byref(d), ptr(work), wb, ctx.stream()))
    if want_index:
        return out, idx, d2
    return out


def nn_fill(atlas, winner, rast2d, pos, want_index=False):
    """in place on atlas [H,W,3]"""
    ctx = get_ctx(atlas.device.index)
    H, W = winner.shape
    T = H * W
    wb = ctx.lib.utx_nn_fill_workspace_bytes(T)
    work = torch.empty(wb, dtype=U8, device=atlas.device)
    idx = torch.empty(T, dtype=I32, device=atlas.device) if want_index else None
    ctx.check(ctx.lib.utx_nn_fill(ctx.handle, ptr(_f(pos)), ptr(winner), ptr(_f(rast2d)), T, ptr(_f(atlas)), ptr(idx), ptr(work), wb, ctx.stream()))
    return idx


def lens_blur_kernel49(radius=3.0):
    """Collapse the 5 separable complex components of lens_blur_torch (image/lens_blur.py:62-112,190-212)
    into one real 7x7 kernel: K = sum_c A_c Re(k_c (x) k_c) + B_c Im(k_c (x) k_c)  (host, float32)."""
    params = [[4.892608, 1.685979, -22.356787, 85.91246], [4.71187, 4.998496, 35.918936, -28.875618],
              [4.052795, 8.244168, -13.212253, -1.578428], [2.929212, 11.900859, 0.507991, 1.816328],
              [1.512961, 16.116382, 0.138051, -0.01]]
    scale = 1.2
    kr = int(math.ceil(radius))
    n = 2 * kr + 1
    ax = np.linspace(-radius, radius, n, dtype=np.float32) * np.float32(scale) * np.float32(1 / radius)
    ks = []
    for a, b, A, B in params:
        re = (np.exp(-a * ax ** 2) * np.cos(b * ax ** 2)).astype(np.float32)
        im = (np.exp(-a * ax ** 2) * np.sin(b * ax ** 2)).astype(np.float32)
        ks.append((re, im, A, B))
    total = 0.0
    for re, im, A, B in ks:
        total += float(np.sum(A * (np.outer(re, re) - np.outer(im, im)) + B * (np.outer(re, im) + np.outer(im, re))))
    K = np.zeros((n, n), dtype=np.float64)
    for re, im, A, B in ks:
        re64, im64 = re.astype(np.float64), im.astype(np.float64)
        K += A * (np.outer(re64, re64) - np.outer(im64, im64)) + B * (np.outer(re64, im64) + np.outer(im64, re64))
    return (K / total).astype(np.float32)


def lens_blur_seam(src, seam, k49=None):
    ctx = get_ctx(src.device.index)
    H, W = seam.shape
    if k49 is None:
        k49 = lens_blur_kernel49()
    arr = (C.c_float * 49)(*[float(x) for x in np.asarray(k49, dtype=np.float32).reshape(-1)])
    dst = torch.empty_like(src)
    ctx.check(ctx.lib.utx_lens_blur_seam(ctx.handle, ptr(_f(src)), ptr(seam), H, W, arr, ptr(dst), ctx.stream()))
    return dst


def pull_push(kd, mask):
    """kd [H,W,3] f32, mask [H,W] u8 -> [H,W,3]"""
    ctx = get_ctx(kd.device.index)
    H, W = mask.shape
    wb = ctx.lib.utx_pull_push_workspace_bytes(H, W)
    work = torch.empty(wb, dtype=U8, device=kd.device)
    out = torch.empty_like(kd)
    ctx.check(ctx.lib.utx_pull_push(ctx.handle, ptr(_f(kd)), ptr(mask), H, W, ptr(out), ptr(work), ctx.stream()))
    return out


def to_u8(img, flip=False):
    ctx = get_ctx(img.device.index)
    rows = img.shape[0]
    row_elems = img.numel() // rows
    out = torch.empty(img.shape, dtype=U8, device=img.device)
    ctx.check(ctx.lib.utx_to_u8(ctx.handle, ptr(_f(img)), rows, row_elems, int(flip), ptr(out), ctx.stream()))
    return out
