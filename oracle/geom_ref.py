"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the TextureTools render / UV back-projection path.

Python face of oracle/geom_ref.c (rasteriser, interpolation, LBVH, ray visibility, per-view gather) plus
numpy restatements of the atlas post-processing.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.

Reference lines restated (relative to /root/reference/TextureTools/texturetools):
  camera/conversion.py:8-28,50-57; camera/generator.py:93-114,153-185               (pinned: G4)
  mesh/structure_v2.py:49-50 (face cross products / normals)
  render/nvdiffrast/renderer_inverse.py:262-298,316-343 (uv_to_pcd), :435-444 (boundary mask),
      :574-633 (bake_mv_to_uv_reproject_blur)                                        (pinned: G5-G7)
  image/lens_blur.py:260-280 + kernel construction :62-112                           (pinned: G5)
  texture/stitching/mip.py:9-95                                                      (pinned: G5)
  pcd/knn/__init__.py:103-113 -> torch_kdtree [3p]: exact 1-NN (ties: lowest index here; unpinned)
nvdiffrast's coverage rule is [3p] and unpinned: see the header of geom_ref.c.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgeom_ref.so")
_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class BpCfg(C.Structure):
    _fields_ = [("T_h", C.c_int), ("T_w", C.c_int), ("n_views", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("cos_thresh", C.c_float)]


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "geom_ref.c")
        so = os.environ.get("UTX_ORACLE_SO") or _SO      # UTX_ORACLE_SO: another build of the same source (the sanitizer build, tests/test_host_sanitizers_cpu.py)
        if so == _SO and ((not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = C.CDLL(so)
        L.utxref_rasterize.argtypes = [f32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, f32p]
        L.utxref_interpolate.argtypes = [f32p, C.c_int, f32p, i32p, C.c_long, f32p]
        L.utxref_bvh_build.argtypes = [f32p, C.c_int, i32p, C.c_int, i32p, f32p, u32p, i32p]
        L.utxref_bvh_trace.argtypes = [i32p, f32p, f32p, i32p, f32p, f32p, C.c_long, i32p, C.POINTER(C.c_long)]
        L.utxref_brute_trace.argtypes = [f32p, i32p, C.c_int, f32p, f32p, C.c_long, i32p, f32p]
        L.utxref_backproject.argtypes = [C.POINTER(BpCfg), f32p, f32p, i32p, f32p, f32p, C.c_int, f32p, f32p, i32p,
                                         f32p, f32p, u8p, u8p]
        L.utxref_nn_fill_brute.argtypes = [f32p, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS"), f32p, C.c_long, f32p, i32p]
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------------------------------------
# cameras (pure restatement; pinned by fixture G4)
# ------------------------------------------------------------------------------------------------
def box_views_c2ws(radius=2.8):
    """generate_box_views_c2ws (camera/generator.py:153-185): front, right, back, left, top, down."""
    r = radius
    m = np.zeros((6, 4, 4), dtype=np.float32)
    rows = [
        ([1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, r]),
        ([0, 0, 1, r], [0, 1, 0, 0], [-1, 0, 0, 0]),
        ([-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, -r]),
        ([0, 0, -1, -r], [0, 1, 0, 0], [1, 0, 0, 0]),
        ([1, 0, 0, 0], [0, 0, 1, r], [0, -1, 0, 0]),
        ([-1, 0, 0, 0], [0, 0, -1, -r], [0, -1, 0, 0]),
    ]
    for i, (a, b, c) in enumerate(rows):
        m[i, 0], m[i, 1], m[i, 2], m[i, 3] = a, b, c, [0, 0, 0, 1]
    return m


def intrinsics(f_x, f_y, fov=True, degree=False):
    """generate_intrinsics (camera/generator.py:93-114)."""
    if fov:
        if degree:
            f_x, f_y = math.radians(f_x), math.radians(f_y)
        fx, fy = 1 / (2 * math.tan(f_x / 2)), 1 / (2 * math.tan(f_y / 2))
    else:
        fx, fy = f_x, f_y
    return np.array([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], dtype=np.float32)


def intr_to_proj(intr, near=0.01, far=1000.0, perspective=True):
    """camera/conversion.py:8-28, including the row-1 sign flip 'for nvdiffrast'."""
    intr = np.asarray(intr, dtype=np.float32)
    p = np.zeros(intr.shape[:-2] + (4, 4), dtype=np.float32)
    if perspective:
        p[..., 0, 0] = 2 * intr[..., 0, 0]
        p[..., 1, 1] = 2 * intr[..., 1, 1]
        p[..., 2, 2] = -(far + near) / (far - near)
        p[..., 0, 2] = 2 * intr[..., 0, 2] - 1
        p[..., 1, 2] = 2 * intr[..., 1, 2] - 1
        p[..., 3, 2] = -1.0
        p[..., 2, 3] = -2.0 * far * near / (far - near)
    else:
        p[..., 0, 0] = intr[..., 0, 0]
        p[..., 1, 1] = intr[..., 1, 1]
        p[..., 2, 2] = -2.0 / (far - near)
        p[..., 3, 3] = 1.0
        p[..., 0, 3] = -(2 * intr[..., 0, 2] - 1)
        p[..., 1, 3] = -(2 * intr[..., 1, 2] - 1)
        p[..., 2, 3] = -(far + near) / (far - near)
    p[..., 1, :] = -p[..., 1, :]
    return p


def c2w_to_w2c(c2w):
    """camera/conversion.py:50-57 (rigid inverse)."""
    c2w = np.asarray(c2w, dtype=np.float32)
    w2c = np.zeros_like(c2w)
    R = c2w[..., :3, :3]
    w2c[..., :3, :3] = np.swapaxes(R, -1, -2)
    w2c[..., :3, 3:] = -np.swapaxes(R, -1, -2) @ c2w[..., :3, 3:]
    w2c[..., 3, 3] = 1.0
    return w2c


def mvp_matrices(c2ws, intr, perspective):
    """proj @ w2c per view (renderer_inverse.py:264)."""
    proj = intr_to_proj(intr, perspective=perspective)
    return (proj @ c2w_to_w2c(c2ws)).astype(np.float32)


def transform_points(verts, mvp):
    """clip = [x,y,z,1] @ mvp^T with the fixed accumulation order ((x*m0 + y*m1) + z*m2) + m3 (float32),
    the order the HIP kernel uses (torch.matmul's order is unspecified: differences ~1 ulp, unpinned)."""
    v = _f(verts)
    out = np.empty((mvp.shape[0], v.shape[0], 4), dtype=np.float32)
    for n in range(mvp.shape[0]):
        m = mvp[n].astype(np.float32)
        for r in range(4):
            out[n, :, r] = ((v[:, 0] * m[r, 0] + v[:, 1] * m[r, 1]) + v[:, 2] * m[r, 2]) + m[r, 3]
    return out


def face_normals(verts, faces):
    """structure_v2.py:49-50: cross(v1-v0, v2-v0), F.normalize(eps 1e-12).  float32, fixed op order."""
    v, f = _f(verts), _i(faces)
    a = v[f[:, 1]] - v[f[:, 0]]
    b = v[f[:, 2]] - v[f[:, 0]]
    c = np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                  a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=-1).astype(np.float32)
    n = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]).astype(np.float32)
    n = np.maximum(n, np.float32(1e-12))
    return (c / n[:, None]).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# C kernels
# ------------------------------------------------------------------------------------------------
def rasterize(pos_clip, tri, H, W):
    pos_clip, tri = _f(pos_clip), _i(tri)
    out = np.zeros((H, W, 4), dtype=np.float32)
    lib().utxref_rasterize(pos_clip, pos_clip.shape[0], tri, tri.shape[0], H, W, out)
    return out


def interpolate(attr, rast, tri):
    attr, tri, rast = _f(attr), _i(tri), _f(rast)
    H, W = rast.shape[:2]
    out = np.zeros((H, W, attr.shape[1]), dtype=np.float32)
    lib().utxref_interpolate(attr, attr.shape[1], rast, tri, H * W, out)
    return out


class BVH:
    def __init__(self, verts, faces):
        self.verts, self.faces = _f(verts), _i(faces)
        F = self.faces.shape[0]
        self.info = np.zeros((2 * F - 1, 3), dtype=np.int32)
        self.aabb = np.zeros((2 * F - 1, 6), dtype=np.float32)
        self.codes = np.zeros(F, dtype=np.uint32)
        self.order = np.zeros(F, dtype=np.int32)
        lib().utxref_bvh_build(self.verts, self.verts.shape[0], self.faces, F, self.info, self.aabb, self.codes, self.order)

    def trace(self, rays_o, rays_d):
        ro, rd = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
        tid = np.full(ro.shape[0], -1, dtype=np.int32)
        nodes = C.c_long(0)
        lib().utxref_bvh_trace(self.info, self.aabb, self.verts, self.faces, ro, rd, ro.shape[0], tid, C.byref(nodes))
        self.nodes_visited = nodes.value
        return tid

    def brute(self, rays_o, rays_d):
        ro, rd = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
        tid = np.full(ro.shape[0], -1, dtype=np.int32)
        t = np.zeros(ro.shape[0], dtype=np.float32)
        lib().utxref_brute_trace(self.verts, self.faces, self.faces.shape[0], ro, rd, ro.shape[0], tid, t)
        return tid, t


def backproject(rast2d, verts, faces, fnormal, vndc, dirs, images, bvh, angle_deg=100.0):
    """per-(view, texel) colour gather + ray visibility (renderer_inverse.py:277-298,316-325).
    rast2d [Th,Tw,4]; vndc [n,V,2]; dirs [n,3]; images [n,H,W,4] (rgb + alpha).
    Returns color [n,Th,Tw,3] f32, rayvis [n,Th,Tw] u8, alphaok [n,Th,Tw] u8."""
    rast2d, verts, faces = _f(rast2d), _f(verts), _i(faces)
    Th, Tw = rast2d.shape[:2]
    n, H, W = images.shape[:3]
    cfg = BpCfg(Th, Tw, n, H, W, np.float32(math.cos(math.radians(angle_deg))))
    color = np.zeros((n, Th, Tw, 3), dtype=np.float32)
    rayvis = np.zeros((n, Th, Tw), dtype=np.uint8)
    alphaok = np.zeros((n, Th, Tw), dtype=np.uint8)
    lib().utxref_backproject(C.byref(cfg), rast2d, verts, faces, _f(fnormal), _f(vndc), verts.shape[0], _f(dirs),
                             _f(images), bvh.info, bvh.aabb, color, rayvis, alphaok)
    return color, rayvis, alphaok


# ------------------------------------------------------------------------------------------------
# atlas post-processing (numpy restatements)
# ------------------------------------------------------------------------------------------------
def _shift_sum(mask_u8, offsets):
    """sum of zero-padded shifted copies of a [..., H, W] uint8 array."""
    H, W = mask_u8.shape[-2:]
    pad = max(max(abs(dy), abs(dx)) for dy, dx in offsets)
    p = np.zeros(mask_u8.shape[:-2] + (H + 2 * pad, W + 2 * pad), dtype=np.int32)
    p[..., pad:pad + H, pad:pad + W] = mask_u8
    out = np.zeros(mask_u8.shape, dtype=np.int32)
    for dy, dx in offsets:
        out += p[..., pad + dy:pad + dy + H, pad + dx:pad + dx + W]
    return out


def dilate_visibility(rayvis, mask2d, alphaok):
    """renderer_inverse.py:326-343 with kernel_mode=7 (A13): k=3 then k=5 hole filling, worked out as
    integer neighbour counts:  k=3: set if any of the 8 neighbours is set;
                               k=5: 25*rim - core >= 135  (rim = 16 border cells, core = inner 3x3).
    then AND coverage, AND (cos < 100.0) [always true, A14], AND alpha > 0.999."""
    m = rayvis.astype(np.uint8).copy()
    n8 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
    # k = 3: kernel = 9 on the rim (8 neighbours), -1 at the centre; threshold ((3-1)^2-1)*(3-2)^2 = 3
    s = 9 * _shift_sum(m, n8) - m.astype(np.int32)
    m = (m.astype(bool) | (s >= 3)).astype(np.uint8)
    rim = [(dy, dx) for dy in range(-2, 3) for dx in range(-2, 3) if max(abs(dy), abs(dx)) == 2]
    core = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    s = 25 * _shift_sum(m, rim) - _shift_sum(m, core)
    m = (m.astype(bool) | (s >= 135)).astype(np.uint8)
    return m.astype(bool) & mask2d.astype(bool)[None] & alphaok.astype(bool)


def boundary_mask(mask):
    """get_boundary_mask (renderer_inverse.py:435-444), kernel 3: inner = set & some 3x3 neighbour unset
    (out-of-image counts as set for the inner test: max_pool pads with -inf on 1-alpha);
    outer = unset & some 3x3 neighbour set."""
    m = mask.astype(bool)
    H, W = m.shape[-2:]
    n9 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    cnt_set = _shift_sum(m.astype(np.uint8), n9)
    inside = _shift_sum(np.ones_like(m, dtype=np.uint8), n9)  # in-image neighbours (incl. self)
    any_unset = cnt_set < inside
    any_set = cnt_set > 0
    return (m & any_unset) | (~m & any_set)


def max_pool(mask, k):
    r = k // 2
    offs = [(dy, dx) for dy in range(-r, r + 1) for dx in range(-r, r + 1)]
    return _shift_sum(mask.astype(np.uint8), offs) > 0


def min_pool(mask, k):
    """1 - max_pool(1 - m): out-of-image neighbours are ignored (max_pool2d pads with -inf)."""
    r = k // 2
    offs = [(dy, dx) for dy in range(-r, r + 1) for dx in range(-r, r + 1)]
    return _shift_sum((~mask.astype(bool)).astype(np.uint8), offs) == 0


PRIORITY = [0, 3, 4, 1, 2, 5]  # renderer_inverse.py:44  frtbld -> f, b, l, r, t, d


def composite(colors, vis, order=PRIORITY):
    """first-come-wins priority composite (renderer_inverse.py:595-602) + seam mask source.
    colors [n,H,W,3], vis [n,H,W] bool -> atlas [H,W,3], seen [H,W], winner [H,W] int8 (-1 none),
    boundary_union [H,W] (OR over views of boundary_mask(newly claimed region))."""
    H, W = vis.shape[1:]
    atlas = np.zeros((H, W, 3), dtype=np.float32)
    seen = np.zeros((H, W), dtype=bool)
    winner = np.full((H, W), -1, dtype=np.int8)
    bnd = np.zeros((H, W), dtype=bool)
    for i in order:
        extra = vis[i] & ~seen
        atlas[extra] = colors[i][extra]
        winner[extra] = i
        seen |= extra
        bnd |= boundary_mask(extra)
    return atlas, seen, winner, bnd


def seam_mask(bnd, mask2d):
    """renderer_inverse.py:603-604: 3x3 max-pool of the boundary union, AND 7x7-eroded coverage."""
    return max_pool(bnd, 3) & min_pool(mask2d, 7)


def nn_fill(atlas, seen, mask2d, pos):
    """renderer_inverse.py:606-615: unseen covered texels take the colour of the nearest seen texel in
    3-D (k=1).  Exact NN; ties -> lowest index in the row-major compacted list of seen texels."""
    from scipy.spatial import cKDTree
    out = atlas.copy()
    src = seen & mask2d
    dst = (~seen) & mask2d
    if dst.sum() == 0 or src.sum() == 0:
        return out, np.zeros(0, dtype=np.int64)
    sp = pos[src].astype(np.float64)
    dp = pos[dst].astype(np.float64)
    # k=2 neighbours to resolve exact ties deterministically (lowest index)
    kk = min(4, sp.shape[0])
    dist, idx = cKDTree(sp).query(dp, k=kk)
    if kk == 1:
        dist, idx = dist[:, None], idx[:, None]
    best = idx[:, 0].copy()
    for j in range(1, kk):
        tie = (dist[:, j] == dist[:, 0]) & (idx[:, j] < best)
        best[tie] = idx[tie, j]
    out[dst] = atlas[src][best]
    return out, best


def nn_fill_brute(atlas, winner, rast2d, pos):
    """exact 1-NN fill with the float32 distance expression and tie rule of the HIP kernel (brute force)."""
    H, W = winner.shape
    out = _f(atlas).copy()
    idx = np.full(H * W, -1, dtype=np.int32)
    lib().utxref_nn_fill_brute(_f(pos).reshape(-1, 3), np.ascontiguousarray(winner, dtype=np.int8).reshape(-1),
                               _f(rast2d).reshape(-1, 4), H * W, out.reshape(-1, 3), idx)
    return out, idx.reshape(H, W)


# ---- lens blur (image/lens_blur.py) -----------------------------------------------------------
_LB5 = [[4.892608, 1.685979, -22.356787, 85.91246], [4.71187, 4.998496, 35.918936, -28.875618],
        [4.052795, 8.244168, -13.212253, -1.578428], [2.929212, 11.900859, 0.507991, 1.816328],
        [1.512961, 16.116382, 0.138051, -0.01]]
_LB5_SCALE = 1.2


def lens_blur_kernels(radius=3.0):
    """5-component complex 1-D kernels, jointly normalised so that sum_c A*Re(k (x) k) + B*Im(k (x) k) = 1
    (lens_blur.py:62-112).  Returns list of (real[7], imag[7], A, B) in float32."""
    kr = int(math.ceil(radius))
    n = 2 * kr + 1
    ax = np.linspace(-radius, radius, n, dtype=np.float32) * np.float32(_LB5_SCALE) * np.float32(1 / radius)
    ks = []
    for a, b, A, B in _LB5:
        re = (np.exp(-a * ax ** 2) * np.cos(b * ax ** 2)).astype(np.float32)
        im = (np.exp(-a * ax ** 2) * np.sin(b * ax ** 2)).astype(np.float32)
        ks.append((re, im, A, B))
    total = np.float32(0.0)
    for re, im, A, B in ks:
        for i in range(n):
            for j in range(n):
                total = np.float32(total + np.float32(A * (re[i] * re[j] - im[i] * im[j]) + B * (re[i] * im[j] + im[i] * re[j])))
    total = np.float32(math.sqrt(total))
    return [((re / total).astype(np.float32), (im / total).astype(np.float32), np.float32(A), np.float32(B)) for re, im, A, B in ks]


def _conv1d(img, k, axis):
    """zero-padded correlation of [C,H,W] float32 with a 1-D kernel along axis (1 = y, 2 = x)."""
    r = len(k) // 2
    out = np.zeros_like(img)
    pad = [(0, 0), (0, 0), (0, 0)]
    pad[axis] = (r, r)
    p = np.pad(img, pad)
    for i, w in enumerate(k):
        sl = [slice(None)] * 3
        sl[axis] = slice(i, i + img.shape[axis])
        out = out + p[tuple(sl)] * np.float32(w)
    return out.astype(np.float32)


def lens_blur(img, radius=3.0, gamma=5.0):
    """lens_blur_torch (lens_blur.py:260-280) on [C,H,W] float32: x^g -> sum_c (A Re + B Im) of the separable
    complex convolution (horizontal 1x7 then vertical 7x1, zero padding) -> clamp>=0 -> ^(1/g) -> clamp[0,1]."""
    x = np.power(img.astype(np.float32), np.float32(gamma)).astype(np.float32)
    acc = np.zeros_like(x)
    for re, im, A, B in lens_blur_kernels(radius):
        ir = _conv1d(x, re, 2)
        ii = _conv1d(x, im, 2)
        f1 = _conv1d(ir, re, 1)
        f2 = _conv1d(ir, im, 1)
        f3 = _conv1d(ii, re, 1)
        f4 = _conv1d(ii, im, 1)
        acc = acc + ((f1 - f4) * A + (f2 + f3) * B)
    out = np.power(np.maximum(acc, 0).astype(np.float32), np.float32(1.0 / gamma))
    return np.clip(out, 0, 1).astype(np.float32)


def lens_blur_kernel49(radius=3.0):
    """the 5 complex separable components collapsed into one real 7x7 kernel (float64 accumulate -> f32)."""
    kr = int(math.ceil(radius))
    n = 2 * kr + 1
    ax = np.linspace(-radius, radius, n, dtype=np.float32) * np.float32(_LB5_SCALE) * np.float32(1 / radius)
    K = np.zeros((n, n), dtype=np.float64)
    for a, b, A, B in _LB5:
        re = (np.exp(-a * ax ** 2) * np.cos(b * ax ** 2)).astype(np.float32).astype(np.float64)
        im = (np.exp(-a * ax ** 2) * np.sin(b * ax ** 2)).astype(np.float32).astype(np.float64)
        K += A * (np.outer(re, re) - np.outer(im, im)) + B * (np.outer(re, im) + np.outer(im, re))
    return (K / K.sum()).astype(np.float32)


def lens_blur_collapsed(img_hwc, seam, k49=None):
    """the HIP kernel's formulation (utx_lens_blur_seam): one real 7x7 correlation of x^5 (zero padding,
    row-major float32 accumulation), ^(1/5), clamp; evaluated on seam texels only."""
    K = lens_blur_kernel49() if k49 is None else np.asarray(k49, dtype=np.float32)
    x = img_hwc.astype(np.float32)
    x5 = ((x * x) * (x * x)) * x
    H, W, _ = x.shape
    p = np.pad(x5, [(3, 3), (3, 3), (0, 0)])
    acc = np.zeros_like(x)
    for dy in range(7):
        for dx in range(7):
            acc = acc + np.float32(K[dy, dx]) * p[dy:dy + H, dx:dx + W]
    out = np.clip(np.power(np.maximum(acc, 0).astype(np.float32), np.float32(0.2)), 0, 1).astype(np.float32)
    return np.where(seam.astype(bool)[..., None], out, x).astype(np.float32)


# ---- pull-push (texture/stitching/mip.py) -------------------------------------------------------
def _pull(kd, mask):
    """pull_push_mip (mip.py:9-24): 2x2 average of colour and alpha; partially covered cells are
    renormalised by their alpha; mask_mip = alpha > 0."""
    C, H, W = kd.shape
    a = mask.astype(np.float32).reshape(H // 2, 2, W // 2, 2)
    a = ((a[:, 0, :, 0] + a[:, 0, :, 1]) + a[:, 1, :, 0] + a[:, 1, :, 1]) * np.float32(0.25)
    k = kd.reshape(C, H // 2, 2, W // 2, 2)
    k = ((k[:, :, 0, :, 0] + k[:, :, 0, :, 1]) + k[:, :, 1, :, 0] + k[:, :, 1, :, 1]) * np.float32(0.25)
    part = (a > 0) & (a < 1)
    k = np.where(part[None], k / np.where(part, a, 1)[None], k).astype(np.float32)
    return k, a > 0


def _push(kd, mask, kd_mip, mask_mip):
    """pull_push_fill (mip.py:27-48): bilinear 2x upsample of the (replicate-padded) mip colour with weights
    9/16, 3/16, 3/16, 1/16; only texels outside `mask` are replaced."""
    C, H, W = kd.shape
    p = np.pad(kd_mip, [(0, 0), (1, 1), (1, 1)], mode="edge")
    up = np.zeros((C, H, W), dtype=np.float32)
    w9, w3, w1 = np.float32(0.5625), np.float32(0.1875), np.float32(0.0625)
    for py in (0, 1):
        for px in (0, 1):
            # fine texel (2i+py, 2j+px): near coarse cell (i, j); far neighbours i-1/i+1, j-1/j+1
            ys, xs = (-1 if py == 0 else 1), (-1 if px == 0 else 1)
            c = p[:, 1:-1, 1:-1]
            cy = p[:, 1 + ys:p.shape[1] - 1 + ys, 1:-1]
            cx = p[:, 1:-1, 1 + xs:p.shape[2] - 1 + xs]
            cxy = p[:, 1 + ys:p.shape[1] - 1 + ys, 1 + xs:p.shape[2] - 1 + xs]
            up[:, py::2, px::2] = _push_order(c, cx, cy, cxy, py, px, w9, w3, w1)
    return np.where(mask[None], kd, up).astype(np.float32), mask


def _push_order(c, cx, cy, cxy, py, px, w9, w3, w1):
    """conv2d accumulation order of the reference's 2x2 kernels over the padded mip (row-major taps)."""
    # taps in row-major order of the 2x2 window; which tap is the near cell depends on the phase
    if py == 0 and px == 0:   # window rows (i-1, i), cols (j-1, j); kernel [[1,3],[3,9]]/16
        return ((cxy * w1 + cy * w3) + cx * w3) + c * w9
    if py == 0 and px == 1:   # cols (j, j+1): kernel [[3,1],[9,3]]/16
        return ((cy * w3 + cxy * w1) + c * w9) + cx * w3
    if py == 1 and px == 0:   # rows (i, i+1): kernel [[3,9],[1,3]]/16
        return ((cx * w3 + c * w9) + cxy * w1) + cy * w3
    return ((c * w9 + cx * w3) + cy * w3) + cxy * w1


def pull_push(kd, mask):
    """pull_push (mip.py:51-95) on [C,H,W] float32 + [H,W] bool."""
    C, H, W = kd.shape
    n = max(min(int(math.log2(H)), int(math.log2(W))) - 2, 0)
    if n == 0:
        return kd.copy()
    kd = np.where(mask[None], kd, 0).astype(np.float32)
    ks, ms = [], []
    k, m = kd, mask
    for _ in range(n):
        k, m = _pull(k, m)
        ks.append(k)
        ms.append(m)
    k, m = ks[-1], ms[-1]
    for lvl in range(n - 1, 0, -1):
        k, m = _push(ks[lvl - 1], ms[lvl - 1], k, m)
    out, _ = _push(kd, mask, k, m)
    return out


def tensor_to_u8(img):
    """tensor_to_image (renderer_utils.py:62-83): clamp(0,1)*255 -> uint8 by TRUNCATION (A4)."""
    return (np.clip(img, 0.0, 1.0).astype(np.float32) * np.float32(255.0)).astype(np.uint8)


def texture_shade(rast, uv01, tri, tex, bg=(1.0, 1.0, 1.0)):
    """NVDiffRendererBase.uv_rendering restricted to what export_orbit_video uses (renderer_base.py:289-336):
    UV interpolation -> dr.texture(filter 'linear', wrap) -> lerp with the background by the coverage mask ->
    clamp * 255 -> uint8 (truncation).  float32 arithmetic in the order the kernel uses."""
    rast, uv01, tri, tex = _f(rast), _f(uv01), _i(tri), _f(tex)
    H, W = rast.shape[:2]
    Ht, Wt = tex.shape[:2]
    f32 = np.float32
    idx = rast[..., 3].astype(np.int32) - 1
    cov = idx >= 0
    t = tri[np.where(cov, idx, 0)]
    u, v = rast[..., 0], rast[..., 1]
    w = (f32(1.0) - u) - v
    a0, a1, a2 = uv01[t[..., 0]], uv01[t[..., 1]], uv01[t[..., 2]]
    tu = (a0[..., 0] * u + a1[..., 0] * v) + a2[..., 0] * w
    tv = (a0[..., 1] * u + a1[..., 1] * v) + a2[..., 1] * w
    x = tu * f32(Wt) - f32(0.5)
    y = tv * f32(Ht) - f32(0.5)
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = (x - x0).astype(f32), (y - y0).astype(f32)
    ix0 = np.mod(x0.astype(np.int64), Wt); ix1 = np.mod(x0.astype(np.int64) + 1, Wt)
    iy0 = np.mod(y0.astype(np.int64), Ht); iy1 = np.mod(y0.astype(np.int64) + 1, Ht)
    one = f32(1.0)
    top = tex[iy0, ix0] * (one - fx)[..., None] + tex[iy0, ix1] * fx[..., None]
    bot = tex[iy1, ix0] * (one - fx)[..., None] + tex[iy1, ix1] * fx[..., None]
    c = top * (one - fy)[..., None] + bot * fy[..., None]
    c = np.where(cov[..., None], c, np.asarray(bg, dtype=f32)[None, None, :]).astype(f32)
    return (np.clip(c, f32(0.0), f32(1.0)) * f32(255.0)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------------------------
# non-default back-projection variants
# ------------------------------------------------------------------------------------------------------------------
def vertex_normals_area(verts, faces):
    """PBRMesh.vertex_normals (mesh/structure_v2.py:64-71): face cross products scattered to the three corners, summed,
    normalised (the mean over the three corner slots only scales the sum)."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    c = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], c)
    zero = (n * n).sum(-1, keepdims=True) <= 1e-20
    n = np.where(zero, np.array([0.0, 0.0, 1.0]), n)
    return (n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)).astype(np.float32)


def view_visibility(attr6, rast, fnormal, dirs, grad_thr=0.20, angle_deg=115.0, radius=15):
    """mv_to_pcd with filt_gradient_points=True (renderer_inverse.py:189-209), float32, same operation order as the kernel.
    attr6 [n,H,W,6], rast [n,H,W,4], fnormal [F,3], dirs [n,3] (ray direction per view) -> vis bool [n,H,W].
    torch.gradient: central differences inside, one-sided at the border.  The reference's MaxPool2d(31, 1, 15) runs on a
    [n,H,W,1] tensor, i.e. with H as channels: it pools along W only -- a 31-wide erosion of each image ROW."""
    a = np.asarray(attr6, np.float32)
    n, H, W, _ = a.shape
    xm = np.concatenate([a[:, :, :1], a[:, :, :-1]], 2); xp = np.concatenate([a[:, :, 1:], a[:, :, -1:]], 2)
    ym = np.concatenate([a[:, :1], a[:, :-1]], 1); yp = np.concatenate([a[:, 1:], a[:, -1:]], 1)
    sx = np.full((1, 1, W, 1), 2.0, np.float32); sx[0, 0, 0, 0] = sx[0, 0, -1, 0] = 1.0
    sy = np.full((1, H, 1, 1), 2.0, np.float32); sy[0, 0, 0, 0] = sy[0, -1, 0, 0] = 1.0
    dx = ((xp - xm) / sx).astype(np.float32); dy = ((yp - ym) / sy).astype(np.float32)
    acc = np.zeros((n, H, W), np.float32)
    for c in range(6):
        acc = (acc + (dx[..., c] * dx[..., c] + dy[..., c] * dy[..., c]).astype(np.float32)).astype(np.float32)
    smooth = np.sqrt(acc).astype(np.float32) < np.float32(grad_thr)
    tid = np.maximum(rast[..., 3].astype(np.int64) - 1, 0)
    fn = np.asarray(fnormal, np.float32)[tid]
    d = np.asarray(dirs, np.float32)[:, None, None, :]
    nd = np.maximum(np.sqrt(((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)), np.float32(1e-8))
    nn = np.maximum(np.sqrt(((fn[..., 0] * fn[..., 0] + fn[..., 1] * fn[..., 1]) + fn[..., 2] * fn[..., 2]).astype(np.float32)), np.float32(1e-8))
    cs = (((d[..., 0] * fn[..., 0] + d[..., 1] * fn[..., 1]) + d[..., 2] * fn[..., 2]).astype(np.float32) / (nd * nn).astype(np.float32)).astype(np.float32)
    facing = cs < np.float32(math.cos(math.radians(angle_deg)))
    er = np.ones_like(smooth)
    for o in range(-radius, radius + 1):          # erosion along W, out-of-range ignored
        sh = np.ones_like(smooth)
        if o < 0:
            sh[:, :, -o:] = smooth[:, :, :o]
        elif o > 0:
            sh[:, :, :-o] = smooth[:, :, o:]
        else:
            sh = smooth
        er &= sh
    return (rast[..., 3] > 0) & facing & er


def knn_brute(src_pos, dst_pos, k, src_mask=None, chunk=2048):
    """exact k nearest sources per query: d2 = (dx*dx + dy*dy) + dz*dz in float32, ascending, ties -> lower source index.
    Returns idx [M,k] (into the unmasked source array, -1 when fewer than k exist) and d2 [M,k]."""
    s = np.asarray(src_pos, np.float32).reshape(-1, 3)
    q = np.asarray(dst_pos, np.float32).reshape(-1, 3)
    ids = np.arange(len(s)) if src_mask is None else np.flatnonzero(np.asarray(src_mask).reshape(-1))
    sv = s[ids]
    M = len(q)
    kk = min(k, len(ids))
    idx = np.full((M, k), -1, np.int64)
    d2o = np.full((M, k), np.inf, np.float32)
    for a in range(0, M, chunk):
        qq = q[a:a + chunk]
        dx = sv[None, :, 0] - qq[:, None, 0]; dy = sv[None, :, 1] - qq[:, None, 1]; dz = sv[None, :, 2] - qq[:, None, 2]
        d2 = ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)
        order = np.lexsort((np.broadcast_to(ids[None, :], d2.shape), d2), axis=1)[:, :kk]      # by d2, then by index
        idx[a:a + chunk, :kk] = ids[order]
        d2o[a:a + chunk, :kk] = np.take_along_axis(d2, order, 1)
    return idx, d2o


def knn_gather(src_pos, src_attr, dst_pos, k, src_mask=None, dst_mask=None, out=None, mode="mean", src_nrm=None, dst_nrm=None):
    """mean (or MVPaint-weighted mean, renderer_inverse.py:389-399) of the k nearest sources' attributes, written to
    out[dst_mask]; float32, neighbour order = ascending distance."""
    sa = np.asarray(src_attr, np.float32).reshape(len(np.asarray(src_pos).reshape(-1, 3)), -1)
    q = np.asarray(dst_pos, np.float32).reshape(-1, 3)
    C = sa.shape[1]
    if out is None:
        out = np.zeros((len(q), C), np.float32)
    sel = np.arange(len(q)) if dst_mask is None else np.flatnonzero(np.asarray(dst_mask).reshape(-1))
    if len(sel) == 0:
        return out
    idx, d2 = knn_brute(src_pos, q[sel], k, src_mask)
    have = (idx >= 0).sum(1)
    res = np.zeros((len(sel), C), np.float32)
    if mode == "mean":
        acc = np.zeros((len(sel), C), np.float32)
        for j in range(idx.shape[1]):
            ok = idx[:, j] >= 0
            acc[ok] = (acc[ok] + sa[idx[ok, j]]).astype(np.float32)
        res = (acc / np.maximum(have, 1)[:, None].astype(np.float32)).astype(np.float32)
    else:
        sn = np.asarray(src_nrm, np.float32).reshape(-1, 3); dn = np.asarray(dst_nrm, np.float32).reshape(-1, 3)[sel]
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = (np.float32(1.0) / d2).astype(np.float32)
        inv = np.where(np.isnan(inv) | (idx < 0), np.float32(0), inv)
        l1 = np.zeros(len(sel), np.float32)
        for j in range(idx.shape[1]):
            l1 = (l1 + np.abs(inv[:, j])).astype(np.float32)
        ndn = np.maximum(np.sqrt(((dn[:, 0] * dn[:, 0] + dn[:, 1] * dn[:, 1]) + dn[:, 2] * dn[:, 2]).astype(np.float32)), np.float32(1e-8))
        wsum = np.zeros(len(sel), np.float32)
        acc = np.zeros((len(sel), C), np.float32)
        ws = []
        for j in range(idx.shape[1]):
            nj = sn[np.maximum(idx[:, j], 0)]
            nsn = np.maximum(np.sqrt(((nj[:, 0] * nj[:, 0] + nj[:, 1] * nj[:, 1]) + nj[:, 2] * nj[:, 2]).astype(np.float32)), np.float32(1e-8))
            cs = (((nj[:, 0] * dn[:, 0] + nj[:, 1] * dn[:, 1]) + nj[:, 2] * dn[:, 2]).astype(np.float32) / (nsn * ndn).astype(np.float32)).astype(np.float32)
            w = ((inv[:, j] / np.maximum(l1, np.float32(1e-12))).astype(np.float32) * cs).astype(np.float32)
            w = np.where(idx[:, j] >= 0, w, np.float32(0))
            ws.append(w)
            wsum = (wsum + w).astype(np.float32)
        for j in range(idx.shape[1]):
            acc = (acc + (sa[np.maximum(idx[:, j], 0)] * ws[j][:, None]).astype(np.float32)).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            res = (acc / wsum[:, None]).astype(np.float32)
        res = np.where(np.isfinite(res), res, np.float32(0)).astype(np.float32)
    out[sel[have > 0]] = res[have > 0]
    return out


def bake_kdtree(view_pos, view_mask, images, vis2d, mask2d, pos2d, nrm2d=None, view_fnormal=None, method="order_mean",
                k_all=32, k_vis=1, k_inv=32, order=PRIORITY):
    """bake_mv_to_uv_kdtree (renderer_inverse.py:367-433) on dense layers, before pull_push.
    view_pos [n,H,W,3], view_mask [n,H,W] bool (mask_visiable), images [n,H,W,C], vis2d [n,T,T] bool, mask2d [T,T] bool,
    pos2d / nrm2d [T,T,3]; view_fnormal [n,H,W,3] (face normal per view pixel, 'mvpaint')."""
    n, H, W, C = images.shape
    T0, T1 = mask2d.shape
    atlas = np.zeros((T0 * T1, C), np.float32)
    if method in ("mean", "mvpaint"):
        kw = {} if method == "mean" else dict(mode="mvpaint", src_nrm=view_fnormal, dst_nrm=nrm2d)
        knn_gather(view_pos, images, pos2d, k_all, src_mask=view_mask, dst_mask=mask2d, out=atlas, **kw)
    else:
        cur = np.zeros((T0, T1), bool)
        for i in order:
            extra = (~cur) & vis2d[i]
            knn_gather(view_pos[i], images[i], pos2d, k_vis, src_mask=view_mask[i], dst_mask=extra, out=atlas)
            cur |= extra
        seen = cur & mask2d
        knn_gather(pos2d, atlas.copy(), pos2d, k_inv, src_mask=seen, dst_mask=mask2d & ~seen, out=atlas)
    return atlas.reshape(T0, T1, C)
