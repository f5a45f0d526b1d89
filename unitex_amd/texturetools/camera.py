"""Camera helpers with the reference's names and conventions (host-side torch, tiny matrices).

Mirrors TextureTools/texturetools/camera/conversion.py:8-28,50-57 and camera/generator.py:93-114,153-185
of the reference (pinned by tests/golden/g4_cameras.npz); the generators off the pipeline's path (hemisphere / sphere / near-front samplers, the canonical Euler grid:
camera/generator.py:42-90,129-151,187-200) are pinned by tests/golden/g12_camera_samplers.npz."""
import math

import torch


def intr_to_proj(intr_mtx: torch.Tensor, near=0.01, far=1000.0, perspective=True):
    """Normalised intrinsics [...,3,3] -> GL-style projection [...,4,4]; row 1 negated (y flip) exactly as
    the reference does 'for nvdiffrast', so image row 0 is the top of the view."""
    fx, fy = intr_mtx[..., 0, 0], intr_mtx[..., 1, 1]
    cx, cy = intr_mtx[..., 0, 2], intr_mtx[..., 1, 2]
    P = torch.zeros((*intr_mtx.shape[:-2], 4, 4), dtype=intr_mtx.dtype, device=intr_mtx.device)
    if perspective:
        P[..., 0, 0], P[..., 1, 1] = 2 * fx, 2 * fy
        P[..., 0, 2], P[..., 1, 2] = 2 * cx - 1, 2 * cy - 1
        P[..., 2, 2] = -(far + near) / (far - near)
        P[..., 2, 3] = -2.0 * far * near / (far - near)
        P[..., 3, 2] = -1.0
    else:
        P[..., 0, 0], P[..., 1, 1] = fx, fy
        P[..., 0, 3], P[..., 1, 3] = -(2 * cx - 1), -(2 * cy - 1)
        P[..., 2, 2] = -2.0 / (far - near)
        P[..., 2, 3] = -(far + near) / (far - near)
        P[..., 3, 3] = 1.0
    P[..., 1, :] = -P[..., 1, :]
    return P


def c2w_to_w2c(c2w: torch.Tensor):
    """rigid inverse: R^T, -R^T t"""
    Rt = c2w[..., :3, :3].transpose(-1, -2)
    w2c = torch.zeros_like(c2w)
    w2c[..., :3, :3] = Rt
    w2c[..., :3, 3:] = -Rt @ c2w[..., :3, 3:]
    w2c[..., 3, 3] = 1.0
    return w2c


def generate_intrinsics(f_x: float, f_y: float, fov=True, degree=False):
    """focal/size (or fov) -> normalised [3,3]; for orthographic cameras f is the scale."""
    if fov:
        if degree:
            f_x, f_y = math.radians(f_x), math.radians(f_y)
        f_x, f_y = 1 / (2 * math.tan(f_x / 2)), 1 / (2 * math.tan(f_y / 2))
    return torch.tensor([[f_x, 0.0, 0.5], [0.0, f_y, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32)


# axis (right, up, back) per box view: front, right, back, left, top, down -- camera centre = radius * back
_BOX_AXES = [
    ((1, 0, 0), (0, 1, 0), (0, 0, 1)),
    ((0, 0, -1), (0, 1, 0), (1, 0, 0)),
    ((-1, 0, 0), (0, 1, 0), (0, 0, -1)),
    ((0, 0, 1), (0, 1, 0), (-1, 0, 0)),
    ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
    ((-1, 0, 0), (0, 0, -1), (0, -1, 0)),
]


def generate_box_views_c2ws(radius=2.8):
    """the six hard-coded axis views of the reference (generator.py:153-185), order f, r, b, l, t, d."""
    out = torch.zeros(6, 4, 4, dtype=torch.float32)
    for i, (rx, up, bk) in enumerate(_BOX_AXES):
        out[i, :3, 0] = torch.tensor(rx, dtype=torch.float32)
        out[i, :3, 1] = torch.tensor(up, dtype=torch.float32)
        out[i, :3, 2] = torch.tensor(bk, dtype=torch.float32)
        out[i, :3, 3] = radius * torch.tensor(bk, dtype=torch.float32)
        out[i, 3, 3] = 1.0
    return out


def _axes_from_back(back):
    """(right, up, back) of a camera at radius * back looking at the origin with world +y up -- the construction that yields
    the four side views of _BOX_AXES when `back` is an axis."""
    b = torch.nn.functional.normalize(torch.tensor(back, dtype=torch.float64), dim=0)
    r = torch.nn.functional.normalize(torch.stack([b[2], torch.zeros((), dtype=torch.float64), -b[0]]), dim=0)   # (0,1,0) x back
    u = torch.linalg.cross(b, r)
    return r.float(), u.float(), b.float()


# the two extra views of the 8-view set: upper diagonals over the front-right and the back-left corners (elevation 35.26 deg)
_DIAGONAL_BACKS = [(1.0, 1.0, 1.0), (-1.0, 1.0, -1.0)]


def generate_views_c2ws(n_views: int, radius=2.8):
    """View sets of the texture path beyond the reference's hard-wired six, in the INVERSE RENDERER's view order, with the
    composite priority (indices into that order, first = wins):
      6 (the reference, pipeline.py:206, renderer_inverse.py:44): f, r, t, b, l, d          priority f, b, l, r, t, d
      4 (export_nvdiffrast_video.py:931-932; BASELINE configs[0]):  f, r, b, l              priority f, b, l, r
      8 (BASELINE configs[4]; builder-defined, the reference has no 8-view set): the six above + two upper diagonal views over
        the (+x,+z) and (-x,-z) corners, orthographic like the others; they come LAST in priority, i.e. they only fill texels
        that no axis view sees (the 54.7-degree grazing band around the cube corners and concavities).
    Returns (c2ws [n,4,4], priority list)."""
    box = generate_box_views_c2ws(radius)
    if n_views == 6:
        return box[[0, 1, 4, 2, 3, 5]], [0, 3, 4, 1, 2, 5]
    if n_views == 4:
        return box[[0, 1, 2, 3]], [0, 2, 3, 1]
    if n_views == 8:
        extra = torch.zeros(2, 4, 4, dtype=torch.float32)
        for i, bk in enumerate(_DIAGONAL_BACKS):
            r, u, b = _axes_from_back(bk)
            extra[i, :3, 0], extra[i, :3, 1], extra[i, :3, 2], extra[i, :3, 3] = r, u, b, radius * b
            extra[i, 3, 3] = 1.0
        return torch.cat([box[[0, 1, 4, 2, 3, 5]], extra]), [0, 3, 4, 1, 2, 5, 6, 7]
    raise ValueError("view sets exist for 4, 6 and 8 views")


def lookat_to_matrix(lookat: torch.Tensor) -> torch.Tensor:
    """camera positions looking at the origin -> c2w (camera/generator.py:8-40).  World x forward / y right / z up is
    re-expressed as z forward / x right / y up by the fixed axis permutation applied on the left."""
    lookat = torch.as_tensor(lookat, dtype=torch.float32)
    e2 = torch.tensor([0.0, 1.0, 0.0]); e3 = torch.tensor([0.0, 0.0, 1.0])
    z_axis = torch.nn.functional.normalize(lookat, dim=-1)
    x_axis = torch.linalg.cross(e3.expand_as(z_axis), z_axis, dim=-1)
    degenerate = (x_axis == 0).all(dim=-1, keepdim=True)          # looking straight down / up: x is hard-coded
    x_axis = torch.where(degenerate, e2, x_axis)
    y_axis = torch.linalg.cross(z_axis, x_axis, dim=-1)
    rots = torch.stack([x_axis, y_axis, z_axis], dim=-1)
    top = torch.cat([rots, lookat.unsqueeze(-1)], dim=-1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(lookat.shape[:-1] + (1, 4))
    c2ws = torch.cat([top, bottom], dim=-2)
    perm = torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    return torch.matmul(perm, c2ws)


def generate_orbit_views_c2ws(num_views: int, radius: float = 1.0, height: float = 0.0, theta_0: float = 0.0, degree=False):
    """cameras on a horizontal circle at `height` on the sphere of `radius` (camera/generator.py:117-127)."""
    if degree:
        theta_0 = math.radians(theta_0)
    projected_radius = math.sqrt(radius ** 2 - height ** 2)
    theta = torch.linspace(theta_0, 2.0 * math.pi + theta_0, num_views, dtype=torch.float32)
    xyz = torch.stack([projected_radius * torch.cos(theta), projected_radius * torch.sin(theta),
                       torch.full((num_views,), fill_value=height, dtype=torch.float32)], dim=-1)
    return lookat_to_matrix(xyz)


# ---- the other camera generators of camera/generator.py (not on the texture pipeline's path; export_orbit_video(enhance_mode='canonical') and the reference's data tools use
# them).  Seeded samplers: the random draws are made in the reference's order so that a seed names the same cameras (tests/golden/g12_camera_samplers.npz).

def _gen(seed):
    return torch.Generator().manual_seed(seed) if seed is not None else None


def _tangent_frame(n: torch.Tensor):
    """an orthonormal pair (t, b) perpendicular to the unit vectors n [..., 3]: t along (1,1,1) x n, or along (-1,1,1) x n where n is parallel to (1,1,1)
    (Blender Cycles' make_orthonormals, which camera/generator.py:42-52 follows)"""
    nx, ny, nz = n[..., 0:1], n[..., 1:2], n[..., 2:3]
    generic = torch.cat([nz - ny, nx - nz, ny - nx], dim=-1)
    diagonal = torch.cat([nz - ny, nx + nz, -ny - nx], dim=-1)
    t = torch.nn.functional.normalize(torch.where((nx != ny) | (nx != nz), generic, diagonal), dim=-1)
    return t, torch.linalg.cross(n, t)


def _concentric_disk(shape, generator=None):
    """Shirley-Chiu concentric map of two uniform draws on [-1, 1]^2 to the unit disk (camera/generator.py:54-64): draw order a, then b"""
    a = 2.0 * torch.rand(shape, dtype=torch.float32, generator=generator) - 1.0
    b = 2.0 * torch.rand(shape, dtype=torch.float32, generator=generator) - 1.0
    a_leads = a ** 2 > b ** 2
    r = torch.where(a_leads, a, b)
    phi = torch.where(a_leads, (torch.pi / 4) * (b / a), (torch.pi / 2) - (torch.pi / 4) * (a / b))
    return r * torch.cos(phi), r * torch.sin(phi)


def _about(n, x, y, z):
    t, b = _tangent_frame(n)
    return x * t + y * b + z * n


def generate_hemisphere_views_c2ws(num_views: int, radius: float = 1.0, seed=None, semi: bool = True):
    """cameras scattered over the upper half space about +z (semi) or over both by a coin (camera/generator.py:66-79,129-134).  Bug-compatible on purpose: the reference draws its
    disk sample PER COMPONENT of the [num_views, 3] axis array (three samples blended through the tangent frame), so the positions are NOT on the sphere of `radius`; a seed
    names the reference's cameras only if that is reproduced (tests/golden/g12_camera_samplers.npz)."""
    up = torch.tensor([0.0, 0.0, 1.0]).unsqueeze(0).repeat(num_views, 1)
    g = _gen(seed)
    x, y = _concentric_disk(up.shape, generator=g)
    z = 1 - (x ** 2 + y ** 2)
    x, y = x * torch.sqrt(z + 1.0), y * torch.sqrt(z + 1.0)
    if not semi:
        z = z * torch.where(torch.randn(up.shape, generator=g) > 0.0, 1.0, -1.0)
    return lookat_to_matrix(radius * _about(up, x, y, z))


def generate_semisphere_views_c2ws(num_views: int, radius: float = 1.0, seed=None, hemi: bool = False):
    """cameras at normalised Gaussian directions -- the whole sphere, or its upper half with hemi (camera/generator.py:136-144)"""
    d = torch.nn.functional.normalize(torch.randn((num_views, 3), dtype=torch.float32, generator=_gen(seed)), dim=-1)
    if hemi:
        d[:, 2] = torch.abs(d[:, 2])
    return lookat_to_matrix(radius * d)


def generate_near_front_views_c2ws(num_views: int, radius: float = 1.0, scale_x: float = 1.0, scale_y: float = 1.0, seed=None):
    """cameras scattered about the front view +x: a Gaussian offset (scale_x, scale_y) in the tangent plane, pulled back towards the sphere (camera/generator.py:81-90,146-151);
    drawn per component of the axis array like the hemisphere sampler above (bug-compatible)"""
    front = torch.tensor([1.0, 0.0, 0.0]).unsqueeze(0).repeat(num_views, 1)
    g = _gen(seed)
    x = torch.randn(front.shape, dtype=torch.float32, generator=g)
    y = torch.randn(front.shape, dtype=torch.float32, generator=g)
    r = torch.sqrt((scale_x ** 2) * (x ** 2) + (scale_y ** 2) * (y ** 2) + 1)
    return lookat_to_matrix(radius * _about(front, scale_x * x / r, scale_y * y / r, 1 / r))


def _axis_rotation(axis: int, angle: torch.Tensor) -> torch.Tensor:
    """rotation matrices [..., 3, 3] about coordinate axis 0 / 1 / 2 (right-handed, column vectors)"""
    c, s = torch.cos(angle), torch.sin(angle)
    i, j = (axis + 1) % 3, (axis + 2) % 3
    m = torch.zeros(angle.shape + (3, 3), dtype=angle.dtype)
    m[..., axis, axis] = 1.0
    m[..., i, i], m[..., i, j], m[..., j, i], m[..., j, j] = c, -s, s, c
    return m


def euler_xyz_to_matrix(angles: torch.Tensor) -> torch.Tensor:
    """extrinsic-order product Rx(a0) Ry(a1) Rz(a2) of Euler angles [..., 3] in radians (camera/rotation.py:199-225 with convention 'XYZ')"""
    rx, ry, rz = (_axis_rotation(k, angles[..., k]) for k in range(3))
    return torch.matmul(torch.matmul(rx, ry), rz)


def generate_canonical_views_c2ws(radius=2.8, steps=(8, 8, 8)):
    """the Euler grid of export_orbit_video(enhance_mode='canonical'): yaw fastest, then pitch, then roll, each over [0, 360) degrees; the camera sits at R (0, 0, radius)
    (camera/generator.py:187-200)"""
    grid = lambda n: [360.0 * i / n for i in range(n)]
    eulers = torch.tensor([[yaw, pitch, roll] for roll in grid(steps[2]) for pitch in grid(steps[1]) for yaw in grid(steps[0])], dtype=torch.float64).to(torch.float32)
    rots = euler_xyz_to_matrix(torch.deg2rad(eulers))
    c2ws = torch.eye(4).repeat(rots.shape[0], 1, 1)
    c2ws[:, :3, :3] = rots
    c2ws[:, :3, 3] = torch.matmul(torch.tensor([[0.0, 0.0, float(radius)]]), rots.transpose(1, 2))[:, 0]
    return c2ws


def parse_color(color):
    """'grey' -> (128,128,128)/255 etc. (utils/parse_color.py:5-19 via PIL's colour map)."""
    if color is None:
        return None
    if isinstance(color, str):
        from PIL import ImageColor
        rgb = ImageColor.getrgb(color)[:3]
        return torch.tensor([c / 255.0 for c in rgb], dtype=torch.float32)
    if isinstance(color, (int, float)):
        return torch.tensor([float(color)] * 3, dtype=torch.float32)
    return torch.tensor(list(color), dtype=torch.float32)
