"""VideoExporter.export_condition -- geometry-condition render (alpha / world-position 'ccm' / world-normal
grids) that feeds the DiT (reference: TextureTools/texturetools/video/export_nvdiffrast_video.py:900-999 on
top of NVDiffRendererBase.simple_rendering, render/nvdiffrast/renderer_base.py:101-200).

HIP path: clip transform -> rasterise per view -> interpolate vertex normals / positions -> fused shade +
uint8 conversion kernel.  The orbit-video export (export_orbit_video) is a 'next' row."""
import numpy as np
import torch
from PIL import Image

from . import camera, meshes, ops


def _vertex_normals(verts, faces):
    """area-weighted vertex normals (sum of face cross products, normalised).  The reference's forward render
    takes trimesh's vertex normals (mesh/structure.py:355-356) [3p, unpinned]."""
    v, f = verts.double(), faces.long()
    c = torch.linalg.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=-1)
    n = torch.zeros_like(v)
    for k in range(3):
        n.index_add_(0, f[:, k], c)
    return torch.nn.functional.normalize(n, dim=-1).float().contiguous()


class VideoExporter:
    def __init__(self, device="cuda"):
        self.device = torch.device(device if device != "cuda" else "cuda:%d" % torch.cuda.current_device())

    def export_condition(self, mesh_path, geometry_scale=1.0, n_views=4, n_rows=2, n_cols=2, H=512, W=512, scale=0.85,
                         fov_deg=49.1, perspective=False, orbit=True, background=None, return_info=False,
                         return_image=True, return_mesh=False, return_camera=False):
        assert n_views == n_rows * n_cols, "Value Error: (n_views, n_rows, n_cols)=%s" % ((n_views, n_rows, n_cols),)
        assert not orbit and not perspective, "the texture pipeline renders orthographic box views (pipeline.py:200-214)"
        if isinstance(mesh_path, str):
            verts, faces, _, _ = meshes.load_obj(mesh_path)
        else:
            verts, faces = mesh_path
        verts = torch.as_tensor(verts, dtype=torch.float32)
        faces = torch.as_tensor(faces, dtype=torch.int32)
        # scale_to_bbox(scale=geometry_scale): largest bbox side -> 2*scale, centred (mesh/structure.py:290-303, A7)
        lo, hi = verts.min(0).values, verts.max(0).values
        s = (hi - lo).max() / (2.0 * geometry_scale)
        verts = ((verts - 0.5 * (lo + hi)) / s).contiguous()
        c2ws = camera.generate_box_views_c2ws(radius=2.8)
        sel = {1: [0], 2: [0, 2], 4: [0, 1, 2, 3], 6: [0, 1, 4, 2, 3, 5] if (n_rows, n_cols) == (2, 3) else list(range(6))}[n_views]
        c2ws = c2ws[sel]
        intrinsics = camera.generate_intrinsics(scale, scale, fov=False, degree=False)
        bg = camera.parse_color(background)
        dev = self.device
        vd, fd = verts.to(dev), faces.to(dev).contiguous()
        nrm = _vertex_normals(verts, faces).to(dev)
        mvp = torch.matmul(camera.intr_to_proj(intrinsics, perspective=False), camera.c2w_to_w2c(c2ws)).to(dev).contiguous()
        clip, _ = ops.transform_points(vd, mvp, want_ndc=False)
        rasts, ns, ps = [], [], []
        for v in range(n_views):
            r = ops.rasterize(clip[v].contiguous(), fd, H, W)
            rasts.append(r)
            ns.append(ops.interpolate(nrm, r, fd))
            ps.append(ops.interpolate(vd, r, fd))
        rast, ni, pi = torch.stack(rasts), torch.stack(ns), torch.stack(ps)
        bgv = bg.tolist() if bg is not None else [0.0, 0.0, 0.0]
        normal_u8, ccm_u8, alpha_u8 = ops.condition_shade(rast, ni, pi, bgv)

        def grid(t):
            a = t.cpu().numpy()
            if a.ndim == 3:
                return a.reshape(n_rows, n_cols, H, W).transpose(0, 2, 1, 3).reshape(n_rows * H, n_cols * W)
            return a.reshape(n_rows, n_cols, H, W, 3).transpose(0, 2, 1, 3, 4).reshape(n_rows * H, n_cols * W, 3)
        results = {"alpha": Image.fromarray(grid(alpha_u8), mode="L"), "ccm": Image.fromarray(grid(ccm_u8), mode="RGB"),
                   "normal": Image.fromarray(grid(normal_u8), mode="RGB")}
        if return_camera:
            results.update({"c2ws": c2ws, "intrinsics": intrinsics, "perspective": perspective})
        return results
