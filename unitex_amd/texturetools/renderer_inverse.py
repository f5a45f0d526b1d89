"""NVDiffRendererInverse -- MI355X-native drop-in for the reference's multi-view -> UV back-projection
(TextureTools/texturetools/render/nvdiffrast/renderer_inverse.py).

Kept call surface (as the reference orchestrator uses it, /root/reference/pipeline.py:330-359):
    inv = NVDiffRendererInverse(device='cuda')
    inv.update_from_file(mesh_path)
    textured_mesh, mask_2d_visiable, mask_2d, color_2d = inv.infer(mesh_path, c2ws=..., intrinsics=...,
        image_attrs=[6,H,W,3], perspective=False, H=, W=, H2D=, W2D=, method='reproject', ...)
    inv.clear();  inv.register_query_field(fn)

What runs where: every per-texel stage is a HIP kernel behind the C ABI (ops.py): clip transform,
UV-space raster, fused gather + LBVH visibility, hole filling, priority composite, seam mask, exact 3-D
NN fill, seam lens blur, pull-push, uint8 conversion.  No [6,2048,2048,3] intermediates other than the
per-view colour layers that the composite (and, multi-GPU, the all-gather) consumes.

Multi-GPU: views are sharded over ranks (`view_shard=(rank, world)`); each rank fills its
views' colour/visibility layers, ONE all_gather (RCCL over xGMI, or gloo in the CPU tests) assembles the
layers, and the composite + post-processing run replicated on every rank.
"""
import contextlib
import os
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import camera, meshes, ops

PRIORITY = [0, 3, 4, 1, 2, 5]  # frtbld -> f, b, l, r, t, d  (reference renderer_inverse.py:44)


class DeviceMesh:
    """PBRMesh of the reference (mesh/structure_v2.py:25-77) reduced to what the inverse renderer reads:
    vertices, faces, per-face normals, UVs mapped to [-1, 1], and the lazily built LBVH ('optix')."""

    def __init__(self, verts, faces, uvs, device):
        self.device = torch.device(device)
        self.vertices = torch.as_tensor(verts, dtype=torch.float32).to(self.device).contiguous()
        self.faces = torch.as_tensor(faces, dtype=torch.int32).to(self.device).contiguous()
        self.uvs01 = np.asarray(uvs, dtype=np.float32)
        self.uvs_2d = (torch.as_tensor(uvs, dtype=torch.float32) * 2.0 - 1.0).to(self.device).contiguous()  # structure_v2.py:287
        self.normals = ops.face_normals(self.vertices, self.faces)      # HIP, bit-exact vs the oracle's op order
        self._bvh = None

    @property
    def optix(self):
        if self._bvh is None:
            self._bvh = ops.BVH(self.vertices, self.faces)
        return self._bvh


def load_device_mesh(path, device):
    verts, faces, uvs, _ = meshes.load_mesh(path)        # .obj / .glb
    if uvs is None:
        raise ValueError("mesh %s has no UVs: run it through meshes.prepare_blank_mesh (pipeline.preprocess_blank_mesh) first" % path)
    return DeviceMesh(verts, faces, uvs, device)


class TexturedMesh:
    """what infer() returns in place of a trimesh.Trimesh: exposes .export(path) for .glb / .obj."""

    def __init__(self, verts, faces, uvs01, texture_u8_top_down):
        self.vertices, self.faces, self.uv, self.texture = verts, faces, uvs01, texture_u8_top_down

    def export(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if path.lower().endswith(".glb"):
            meshes.save_glb(path, self.vertices, self.faces, self.uv, self.texture)
        else:
            from PIL import Image
            base = os.path.splitext(path)[0]
            Image.fromarray(self.texture).save(base + ".png")
            with open(base + ".mtl", "w") as f:
                f.write("newmtl material_0\nmap_Kd %s\n" % os.path.basename(base + ".png"))
            meshes.save_obj(path, self.vertices, self.faces, self.uv, mtl=os.path.basename(base + ".mtl"))
        return path


class NVDiffRendererInverse:
    def __init__(self, device="cuda", pbr_mesh: Optional[DeviceMesh] = None, view_shard: Tuple[int, int] = (0, 1),
                 process_group=None):
        self.device = torch.device(device if device != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.pbr_mesh = pbr_mesh
        self.index = list(PRIORITY)
        self.query_field_function = None
        self.view_shard = view_shard
        self.process_group = process_group
        self.last = {}
        self.stage_events = None   # set to [] to collect (stage, start_event, end_event) per infer() stage

    @contextlib.contextmanager
    def _stage(self, name):
        """optional HIP-event bracket per stage (bench.py / tools/bench_backproject.py); no sync, no cost when off."""
        if self.stage_events is None:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.stage_events.append((name, a, b))

    # ---- reference surface
    def clear(self):
        self.pbr_mesh = None
        self.query_field_function = None

    def update_from_file(self, pbr_mesh_path=None):
        assert pbr_mesh_path is not None
        self.pbr_mesh = load_device_mesh(pbr_mesh_path, self.device)
        return self

    def update_from_arrays(self, verts, faces, uvs):
        self.pbr_mesh = DeviceMesh(verts, faces, uvs, self.device)
        return self

    def register_query_field(self, query_field: Optional[Callable] = None):
        self.query_field_function = query_field
        return self

    # ---- stages
    def _mvp(self, c2ws, intrinsics, perspective):
        c2ws = torch.as_tensor(c2ws, dtype=torch.float32).cpu()
        intr = torch.as_tensor(intrinsics, dtype=torch.float32).cpu()
        mvp = torch.matmul(camera.intr_to_proj(intr, perspective=perspective), camera.c2w_to_w2c(c2ws))
        return mvp.to(self.device).contiguous(), c2ws

    def mv_to_pcd(self, c2ws, intrinsics, render_size, perspective=True):
        """view-space coverage alpha [n,H,W] (float 0/1) -- all that the 'reproject' path consumes of the
        reference's mv_to_pcd when filt_gradient_points=False (renderer_inverse.py:183-185,211-213)."""
        H, W = (render_size, render_size) if isinstance(render_size, int) else render_size
        m = self.pbr_mesh
        mvp, _ = self._mvp(c2ws, intrinsics, perspective)
        clip, ndc = ops.transform_points(m.vertices, mvp)
        alpha = torch.empty(mvp.shape[0], H, W, dtype=torch.float32, device=self.device)
        for v in range(mvp.shape[0]):
            rast = ops.rasterize(clip[v].contiguous(), m.faces, H, W)
            alpha[v] = (rast[..., 3] > 0).float()
        return {"alpha": alpha, "clip": clip, "ndc": ndc}

    def infer(self, blank_mesh, c2ws, intrinsics, image_attrs, H=512, W=512, H2D=2048, W2D=2048, perspective=True,
              grad_norm_threhold=0.20, ray_normal_angle_threhold=115.0, grid_interpolate_mode="torch", method="reproject",
              reproject_method="lens", reproject_inpainting=False, filt_gradient_points=True, return_layers=False, **unused):
        assert method == "reproject", "only the default 'reproject' path is built (kdtree / blending variants: SURVEY 8f rank 4)"
        assert not perspective, "the reference's texture path is orthographic (pipeline.py:208-210)"
        assert not filt_gradient_points and not reproject_inpainting, "LTM / inpainting branch is unreleased in the reference"
        assert len(self.index) == image_attrs.shape[0] == torch.as_tensor(c2ws).shape[0]
        m = self.pbr_mesh
        n = image_attrs.shape[0]
        dev = self.device
        with self._stage("view_raster"):
            mv = self.mv_to_pcd(c2ws, intrinsics, (H, W), perspective=perspective)
        images = torch.cat([torch.as_tensor(image_attrs, dtype=torch.float32).to(dev), mv["alpha"][..., None]], dim=-1).contiguous()
        _, c2ws_cpu = self._mvp(c2ws, intrinsics, perspective)
        dirs = (-c2ws_cpu[:, :3, 2]).contiguous().to(dev)
        # UV-space raster: uv in [-1,1] used directly as clip xy, z = 0, w = 1 (renderer_inverse.py:268-274)
        uvclip = torch.cat([m.uvs_2d, torch.zeros_like(m.uvs_2d[:, :1]), torch.ones_like(m.uvs_2d[:, :1])], dim=-1).contiguous()
        with self._stage("uv_raster"):
            rast2d = ops.rasterize(uvclip, m.faces, H2D, W2D)
        from .distributed import view_range
        rank, world = self.view_shard
        v0, v1, per = view_range(rank, world, n)
        color = torch.zeros(n, H2D, W2D, 3, dtype=torch.float32, device=dev)
        rayvis = torch.zeros(n, H2D, W2D, dtype=torch.uint8, device=dev)
        alphaok = torch.zeros(n, H2D, W2D, dtype=torch.uint8, device=dev)
        with self._stage("bvh_build"):
            bvh = m.optix
        if v1 > v0:
            with self._stage("backproject"):
                ops.backproject(rast2d, m.vertices, m.faces, m.normals, mv["ndc"].contiguous(), dirs, images, bvh,
                                angle_deg=ray_normal_angle_threhold, view_begin=v0, view_count=v1 - v0, out=(color, rayvis, alphaok))
        with self._stage("dilate_visibility"):
            vis = ops.dilate_visibility(rayvis, alphaok, rast2d)
        if world > 1:
            color, vis = self._gather_layers(color, vis, per, n)
        with self._stage("composite"):
            atlas, winner = ops.composite(color, vis, self.index)
        with self._stage("seam_mask"):
            seam = ops.seam_mask(winner, rast2d)
        with self._stage("nn_fill"):
            pos = ops.interpolate(m.vertices, rast2d, m.faces)
            ops.nn_fill(atlas, winner, rast2d, pos)
        with self._stage("lens_blur_seam"):
            blurred = ops.lens_blur_seam(atlas, seam)
        mask_u8 = (rast2d[..., 3] > 0).to(torch.uint8).contiguous()
        with self._stage("pull_push"):
            color_2d = ops.pull_push(blurred, mask_u8)
        with self._stage("to_u8"):
            tex = ops.to_u8(color_2d, flip=True)  # tensor_to_image + FLIP_TOP_BOTTOM (link_pbr_to_mesh.py:17)
        textured = TexturedMesh(m.vertices.cpu().numpy(), m.faces.cpu().numpy(), m.uvs01, tex.cpu().numpy())
        self.last = {"rast2d": rast2d, "winner": winner, "seam": seam, "atlas_prefill": atlas}
        out = (textured, vis.bool()[..., None], (rast2d[..., 3] > 0)[None, ..., None], color_2d[None])
        if return_layers:
            return out + (color, vis)
        return out

    def _gather_layers(self, color, vis, per, n):
        """ONE all-gather of the per-view layers (texturetools/distributed.py)."""
        from .distributed import gather_view_layers
        rank, world = self.view_shard
        return gather_view_layers(color, vis, rank, world, group=self.process_group)
