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
import time
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
        self._vertex_normals = None

    @property
    def vertex_normals(self):
        """area-weighted vertex normals (mesh/structure_v2.py:64-71: face cross products scattered to the corners, normalised)."""
        if self._vertex_normals is None:
            from .video import _vertex_normals
            self._vertex_normals = _vertex_normals(self.vertices.cpu(), self.faces.cpu(), weighting="area").to(self.device)
        return self._vertex_normals

    @property
    def optix(self):
        if self._bvh is None:
            self._bvh = ops.BVH(self.vertices, self.faces)
        return self._bvh


def load_device_mesh(path, device):
    verts, faces, uvs, _ = meshes.load_mesh(path)        # .obj / .glb / .gltf / .ply / .stl / .off
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
            Image.fromarray(self.texture).save(base + ".png", compress_level=int(os.environ.get("UTX_PNG_LEVEL", "1")))
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

    def mv_to_pcd(self, c2ws, intrinsics, render_size, perspective=True, grad_norm_threhold=0.20, ray_normal_angle_threhold=115.0,
                  filt_gradient_points=False, want_points=False):
        """view-space pass of the reference's mv_to_pcd (renderer_inverse.py:159-241), dense instead of compacted:
          alpha [n,H,W] f32 = mask_visiable as float: plain coverage, or with filt_gradient_points=True coverage AND
          back-facing-ray test AND the (row-wise, 31 wide) eroded gradient test (:189-209);
          want_points: also pos [n,H,W,3] (interpolated vertex positions = the point cloud of :224-232) and the raster."""
        H, W = (render_size, render_size) if isinstance(render_size, int) else render_size
        m = self.pbr_mesh
        mvp, c2ws_cpu = self._mvp(c2ws, intrinsics, perspective)
        clip, ndc = ops.transform_points(m.vertices, mvp)
        n = mvp.shape[0]
        rast = torch.empty(n, H, W, 4, dtype=torch.float32, device=self.device)
        for v in range(n):
            rast[v] = ops.rasterize(clip[v].contiguous(), m.faces, H, W)
        out = {"clip": clip, "ndc": ndc}
        attrs = None
        if filt_gradient_points or want_points:
            va = torch.cat([m.vertices, m.vertex_normals], dim=-1).contiguous()
            attrs = ops.interpolate(va, rast.view(n * H, W, 4), m.faces).view(n, H, W, 6)
        if filt_gradient_points:
            assert not perspective
            dirs = torch.nn.functional.normalize(-c2ws_cpu[:, :3, 2], dim=-1).contiguous().to(self.device)
            vis, alpha = ops.view_visibility(attrs, rast, m.normals, dirs, grad_thr=grad_norm_threhold, angle_deg=ray_normal_angle_threhold)
        else:
            alpha = (rast[..., 3] > 0).float()
            vis = (rast[..., 3] > 0).to(torch.uint8)
        out["alpha"], out["mask_visiable"] = alpha, vis
        if want_points:
            out["pos"] = attrs[..., :3].contiguous()
            out["rast"] = rast
        return out

    def query_field(self, vertices_visiable, colors_visiable, vertices_invisiable):
        """the LTM hook (renderer_inverse.py:143-157): colours for unseen points from the seen ones."""
        if self.query_field_function is None:
            raise NotImplementedError("using register_query_field before query")
        return self.query_field_function(vertices_visiable, colors_visiable, vertices_invisiable)

    def _fill_unseen(self, atlas, seen_u8, covered, pos, inpainting, k=1):
        """colours for covered-but-unseen texels, in place on atlas [H2D,W2D,C]: the k nearest seen texels in 3-D (mean),
        or the registered query field (renderer_inverse.py:420-426 / :606-615)."""
        unseen = covered & (seen_u8 == 0)
        if inpainting:
            seen_b = seen_u8.bool()
            col = self.query_field(pos[seen_b], atlas[seen_b], pos[unseen])      # compaction only at this (user) boundary
            atlas[unseen] = torch.as_tensor(col, dtype=torch.float32, device=atlas.device)
        else:
            ops.knn_gather(pos, pos, k, src_attr=atlas, src_mask=seen_u8, dst_mask=unseen.to(torch.uint8), out=atlas.view(-1, atlas.shape[-1]))
        return unseen

    def _bake_kdtree(self, mv, images, vis, rast2d, pos2d, method, k_all, k_vis, k_inv, inpainting):
        """bake_mv_to_uv_kdtree (renderer_inverse.py:367-433) on dense layers.
        mv: view pass (pos [n,H,W,3], mask_visiable [n,H,W]); images [n,H,W,C]; vis [n,H2D,W2D] u8; pos2d [H2D,W2D,3]."""
        n, Hh, Ww, C = images.shape
        H2D, W2D = rast2d.shape[:2]
        covered = rast2d[..., 3] > 0
        cov_u8 = covered.to(torch.uint8)
        atlas = torch.zeros(H2D, W2D, C, dtype=torch.float32, device=self.device)
        if method in ("mean", "mvpaint"):
            if method == "mean" and inpainting:
                b = mv["mask_visiable"].bool()
                col = self.query_field(mv["pos"][b], images[b], pos2d[covered])
                atlas[covered] = torch.as_tensor(col, dtype=torch.float32, device=self.device)
            else:
                kw = {}
                if method == "mvpaint":   # both point clouds carry FACE normals (renderer_inverse.py:226-233, 350)
                    tid = (mv["rast"][..., 3].long() - 1).clamp_(min=0)
                    tid2 = (rast2d[..., 3].long() - 1).clamp_(min=0)
                    kw = dict(mode="mvpaint", src_nrm=self.pbr_mesh.normals[tid.view(-1)], dst_nrm=self.pbr_mesh.normals[tid2.view(-1)])
                ops.knn_gather(mv["pos"], pos2d, k_all, src_attr=images, src_mask=mv["mask_visiable"], dst_mask=cov_u8,
                               out=atlas.view(-1, C), **kw)
        elif method == "order_mean":
            current = torch.zeros(H2D, W2D, dtype=torch.bool, device=self.device)
            for i in self.index:
                extra = (~current) & vis[i].bool()
                ops.knn_gather(mv["pos"][i], pos2d, k_vis, src_attr=images[i], src_mask=mv["mask_visiable"][i],
                               dst_mask=extra.to(torch.uint8), out=atlas.view(-1, C))
                current |= extra
            self._fill_unseen(atlas, (current & covered).to(torch.uint8), covered, pos2d, inpainting, k=k_inv)
        else:
            raise NotImplementedError("method %s is not supported" % method)
        return atlas

    def infer(self, blank_mesh, c2ws, intrinsics, image_attrs, H=512, W=512, H2D=2048, W2D=2048, perspective=True,
              grad_norm_threhold=0.20, ray_normal_angle_threhold=115.0, grid_interpolate_mode="torch", method="reproject",
              kdtree_n_neighbors=32, kdtree_n_neighbors_visiable=1, kdtree_n_neighbors_invisiable=32, kdtree_method="order_mean",
              kdtree_inpainting=False, reproject_method="lens", reproject_inpainting=False, filt_gradient_points=True,
              return_layers=False, **unused):
        """renderer_inverse.py:635-726.  method='reproject' is bake_mv_to_uv_reproject_blur (the pipeline's path),
        method='kdtree' bake_mv_to_uv_kdtree ('order_mean' | 'mean' | 'mvpaint'); *_inpainting=True routes the unseen texels
        through the registered query field (the LTM hook) instead of the nearest-neighbour fill; filt_gradient_points adds
        the gradient / facing filter to the view masks.  Colours: 3 channels (rgb) or 9 (PBR stack) for 'kdtree', 3 for 'reproject'."""
        assert method in ("kdtree", "reproject")
        assert not perspective, "the reference's texture path is orthographic (pipeline.py:208-210)"
        t_host0 = time.perf_counter()
        # keyword arguments of the reference's signature (renderer_inverse.py:635-659) that this build fixes at the values the pipeline uses: anything else is
        # refused, not dropped
        fixed = dict(reproject_kernel_size_boundary=3, reproject_kernel_size_boundary_blur=3, reproject_kernel_size_blur=5, return_mv_reproject_uv=False)
        for k_, v_ in unused.items():
            if k_ not in fixed:
                raise TypeError("infer() got an unexpected keyword argument %r" % k_)
            if v_ != fixed[k_]:
                raise NotImplementedError("infer(%s=%r): only %r (the pipeline's value) is built" % (k_, v_, fixed[k_]))
        if grid_interpolate_mode not in ("torch", "pytorch"):
            raise NotImplementedError("grid_interpolate_mode %r: the bilinear grid_sample form ('torch') is the one the pipeline uses and the one built" % (grid_interpolate_mode,))
        assert reproject_method == "lens"
        assert len(self.index) == image_attrs.shape[0] == torch.as_tensor(c2ws).shape[0]
        m = self.pbr_mesh
        n = image_attrs.shape[0]
        dev = self.device
        image_attrs = torch.as_tensor(image_attrs, dtype=torch.float32).to(dev)
        with self._stage("view_raster"):
            mv = self.mv_to_pcd(c2ws, intrinsics, (H, W), perspective=perspective, grad_norm_threhold=grad_norm_threhold,
                                ray_normal_angle_threhold=ray_normal_angle_threhold, filt_gradient_points=filt_gradient_points,
                                want_points=(method == "kdtree"))
        # the alpha channel the texels sample is mask_visiable (uv_to_pcd(alpha_attrs=alpha_visiable), :661-670)
        images = torch.cat([image_attrs[..., :3], mv["alpha"][..., None]], dim=-1).contiguous()
        _, c2ws_cpu = self._mvp(c2ws, intrinsics, perspective)
        dirs = (-c2ws_cpu[:, :3, 2]).contiguous().to(dev)
        # UV-space raster: uv in [-1,1] used directly as clip xy, z = 0, w = 1 (renderer_inverse.py:268-274)
        uvclip = torch.cat([m.uvs_2d, torch.zeros_like(m.uvs_2d[:, :1]), torch.ones_like(m.uvs_2d[:, :1])], dim=-1).contiguous()
        with self._stage("uv_raster"):
            rast2d = ops.rasterize(uvclip, m.faces, H2D, W2D)
        from .distributed import view_range
        rank, world = self.view_shard
        v0, v1, per = view_range(rank, world, n)
        # the back-projection kernel writes EVERY texel of the views it is given; only a view shard (world > 1) leaves layers to others, and those start as zeros
        alloc = torch.zeros if (world > 1 or v1 - v0 < n) else torch.empty
        color = alloc(n, H2D, W2D, 3, dtype=torch.float32, device=dev)
        rayvis = alloc(n, H2D, W2D, dtype=torch.uint8, device=dev)
        alphaok = alloc(n, H2D, W2D, dtype=torch.uint8, device=dev)
        with self._stage("bvh_build"):
            bvh = m.optix
        if v1 > v0:
            with self._stage("backproject"):
                ops.backproject(rast2d, m.vertices, m.faces, m.normals, mv["ndc"].contiguous(), dirs, images, bvh,
                                angle_deg=ray_normal_angle_threhold, view_begin=v0, view_count=v1 - v0, out=(color, rayvis, alphaok))
        with self._stage("dilate_visibility"):
            vis = ops.dilate_visibility(rayvis, alphaok, rast2d)
        if world > 1:
            with self._stage("all_gather"):          # the one exchange step of the path
                color, vis = self._gather_layers(color, vis, per, n)
        mask_u8 = (rast2d[..., 3] > 0).to(torch.uint8).contiguous()
        winner = seam = None
        if method == "reproject":
            assert image_attrs.shape[-1] == 3
            with self._stage("composite"):
                atlas, winner = ops.composite(color, vis, self.index)
            with self._stage("seam_mask"):
                seam = ops.seam_mask(winner, rast2d)
            with self._stage("nn_fill"):
                pos = ops.interpolate(m.vertices, rast2d, m.faces)
                if reproject_inpainting:
                    self._fill_unseen(atlas, (winner >= 0).to(torch.uint8), rast2d[..., 3] > 0, pos, True)
                else:
                    ops.nn_fill(atlas, winner, rast2d, pos)
            with self._stage("lens_blur_seam"):
                baked = ops.lens_blur_seam(atlas, seam)
        else:
            with self._stage("kdtree_bake"):
                pos2d = ops.interpolate(m.vertices, rast2d, m.faces)
                baked = self._bake_kdtree(mv, image_attrs.contiguous(), vis, rast2d, pos2d,
                                          kdtree_method, kdtree_n_neighbors, kdtree_n_neighbors_visiable, kdtree_n_neighbors_invisiable,
                                          kdtree_inpainting)
        with self._stage("pull_push"):
            if baked.shape[-1] == 3:
                color_2d = ops.pull_push(baked, mask_u8)
            else:   # PBR stack: three rgb groups through the same 3-channel kernel
                color_2d = torch.cat([ops.pull_push(baked[..., c:c + 3].contiguous(), mask_u8) for c in range(0, baked.shape[-1], 3)], dim=-1)
        with self._stage("to_u8"):
            tex = ops.to_u8(color_2d[..., :3].contiguous(), flip=True)  # tensor_to_image + FLIP_TOP_BOTTOM (link_pbr_to_mesh.py:17)
        self.host_enqueue_ms = (time.perf_counter() - t_host0) * 1e3      # the host's wall time up to the last enqueue.  NOT free of device waits when the tree is fresh: the first utx_backproject behind utx_bvh_build_ws blocks in
        # hipEventSynchronize(depth_ready) until the build -- and everything queued in front of it -- has run (the depth picks the traversal on the host), so on a new mesh this figure
        # contains that GPU time; on a cached tree (DeviceMesh.optix, the benchmark's later iterations) it is enqueue time only.  The copies below wait for the GPU
        textured = TexturedMesh(m.vertices.cpu().numpy(), m.faces.cpu().numpy(), m.uvs01, tex.cpu().numpy())
        self.last = {"rast2d": rast2d, "winner": winner, "seam": seam, "atlas_prefill": baked, "view_mask": mv["mask_visiable"]}
        out = (textured, vis.bool()[..., None], (rast2d[..., 3] > 0)[None, ..., None], color_2d[None])
        if return_layers:
            return out + (color, vis)
        return out

    def _gather_layers(self, color, vis, per, n):
        """ONE all-gather of the per-view layers (texturetools/distributed.py)."""
        from .distributed import gather_view_layers
        rank, world = self.view_shard
        return gather_view_layers(color, vis, rank, world, group=self.process_group)
