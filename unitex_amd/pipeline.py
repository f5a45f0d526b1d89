"""CustomRGBTextureFullPipeline -- the reference's call surface (/root/reference/pipeline.py:141-632) on MI355X.

    pipe = CustomRGBTextureFullPipeline(pretrain_models=..., super_resolutions=False, seed=63)
    rembg_png, glb = pipe(save_dir, input_image_path, input_mesh_path, clear_cache=False)

Stage sequence, artefact names and the uint8-PNG hand-off between stages (A5) are the reference's.  The DiT
is FluxDiT + PBRFluxPipeline (HIP kernels), render / back-projection are VideoExporter /
NVDiffRendererInverse (HIP kernels); meshes without UVs are cleaned and unwrapped on the host (texturetools/meshes.py),
the orbit video is rendered by VideoExporter.export_orbit_video.  Not built: RMBG-2.0 background removal (an existing
alpha channel is honoured instead) and super-resolution (TSD_SR, off by default in the reference)."""
import os
import shutil
from typing import Tuple

import numpy as np
import torch
from PIL import Image

from .texturetools.timer import CPUTimer, Encoders

# zlib level of the stage hand-off PNGs.  PNG is lossless: the level changes the file size (+15 % at 1 against PIL's default 6), never a pixel --
# and the encoder's time: a 2048^2 RGB atlas costs 0.3-0.5 s at level 6 on the GPU box's host, several times the whole back-projection chain.
PNG_LEVEL = int(os.environ.get("UTX_PNG_LEVEL", "1"))


def build_pipeline(pretrain_models=None, pipeline_name="texture_plus", device="cuda:0", seed=0, lora_rank=64, shape=None,
                   sequence_parallel=False, process_group=None, speedup_mode=None, add_lora_path=None, add_lora_weights=None):
    """FluxDiT + VAE + adapters.  speedup_mode: the reference's constructor takes the argument and never reads it (pipeline.py:81,142-145); here
    "fp8" runs the big linears on OCP MX fp8 operands (FluxDiT(fp8_weights=True): BASELINE configs[4] numerics, NOT the default), "fp8-attn" additionally runs
    QK^T / PV on the fp8 matrix pipe (FluxDiT(fp8_attention=True), csrc/attention_fp8.hip; single GPU), anything else = bf16.  With a `pretrain_models` directory holding diffusers-format safetensors the real
    weights are loaded; otherwise (no checkpoints exist here) FLUX.1-dev-shaped synthetic weights are generated.
    sequence_parallel: ONE job over the ranks of `process_group` (flux/ulysses.py).
    add_lora_path / add_lora_weights (reference pipeline.py:112-117): further adapters `add_lora_<i>` (safetensors paths, or already loaded dicts), switched on
    with weight add_lora_weights[i] in BOTH passes beside the texture / delight adapter -- they join the rank-concatenated LoRA segment of the GEMMs (bf16) or
    are merged into the fp8 weights like the others."""
    from .flux.pipeline import PBRFluxPipeline
    from .flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from .flux.transformer import FluxDiT, FluxShape
    from .flux.vae_hip import AutoencoderKL
    shape = shape or FluxShape()
    ckpt = os.path.join(pretrain_models, "black-forest-labs", "FLUX.1-dev") if pretrain_models else None
    if ckpt and os.path.isdir(os.path.join(ckpt, "transformer")):
        from .flux.lora_io import load_flux_transformer_state_dict, load_lora_safetensors, load_vae
        sd = load_flux_transformer_state_dict(os.path.join(ckpt, "transformer"))
        vae = load_vae(os.path.join(ckpt, "vae"), device)
        tex = load_lora_safetensors(os.path.join(pretrain_models, "UniTex", "texture_gen", "pytorch_lora_weights.safetensors"))
        dlt = load_lora_safetensors(os.path.join(pretrain_models, "UniTex", "delight", "pytorch_lora_weights.safetensors"))
    else:
        sd = SyntheticFluxStateDict(shape, seed=0, device=device)
        vae = AutoencoderKL.synthetic(seed=0, device=device)
        tex = synthetic_lora(sd, shape, rank=lora_rank, seed=1, device=device)
        dlt = synthetic_lora(sd, shape, rank=lora_rank, seed=2, device=device)
    pipe = PBRFluxPipeline(FluxDiT(sd, shape, device=device, sequence_parallel=sequence_parallel, sp_group=process_group,
                                   fp8_weights=(speedup_mode in ("fp8", "fp8-attn")), fp8_attention=(speedup_mode == "fp8-attn")), vae, device=device)
    pipe.load_lora_weights(tex, adapter_name="texture")
    pipe.load_lora_weights(dlt, adapter_name="delight")
    weights_for_texture, weights_for_delight, adapter_names = [1.0, 0.0], [0.0, 1.0], ["texture", "delight"]
    if add_lora_path is not None:
        if add_lora_weights is None or len(add_lora_weights) != len(add_lora_path):
            raise ValueError("add_lora_weights must give one weight per entry of add_lora_path")
        for i, src in enumerate(add_lora_path):
            pipe.load_lora_weights(src, adapter_name="add_lora_%d" % i)
            adapter_names.append("add_lora_%d" % i)
            weights_for_texture.append(float(add_lora_weights[i]))
            weights_for_delight.append(float(add_lora_weights[i]))
    pipe._num_inference_steps = 28
    return pipe, weights_for_texture, weights_for_delight, adapter_names


class RGBTextureFullPipelineBase:
    def __init__(self, pretrain_models=None, pipeline_name="texture_plus", super_resolutions=False, seed=0, speedup_mode=None,
                 add_lora_path=None, add_lora_weights=None, enable_rembg=False, device="cuda:0", pipeline=None,
                 num_inference_steps=None, atlas_size=2048, view_size=512, multi_gpu=None, process_group=None, n_views=6):
        """multi_gpu (beyond the reference, which is single-GPU): None = automatic -- when torch.distributed is initialised with
        more than one rank, the ranks work on ONE mesh together: the DiT runs sequence-parallel (two all-to-alls per layer), the
        geometry-condition render and the back-projection are sharded by view (`view_shard=(rank, world)`, ONE all-gather each;
        SURVEY 8e).  Every rank executes the same orchestration and ends with the same atlas; rank 0 owns `save_dir`, the other
        ranks keep their intermediate artefacts in `cache.rank<r>`.  False = each process is an independent single-GPU job."""
        import torch.distributed as dist
        from .texturetools.renderer_inverse import NVDiffRendererInverse
        from .texturetools.video import VideoExporter
        if super_resolutions:
            raise NotImplementedError("TSD_SR super-resolution is off by default in the reference (run.py:4) and out of scope")
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if multi_gpu is None:
            multi_gpu = world > 1
        if multi_gpu and world < 2:
            raise ValueError("multi_gpu=True needs an initialised torch.distributed group with more than one rank")
        self.rank, self.world = (dist.get_rank(process_group), world) if multi_gpu else (0, 1)
        self.process_group = process_group if multi_gpu else None
        shard = (self.rank, self.world)
        if pipeline is None:
            pipeline, wt, wd, names = build_pipeline(pretrain_models, pipeline_name, device=device, sequence_parallel=bool(multi_gpu),
                                                     process_group=self.process_group, speedup_mode=speedup_mode, add_lora_path=add_lora_path,
                                                     add_lora_weights=add_lora_weights)
        else:
            wt, wd, names = [1.0, 0.0], [0.0, 1.0], ["texture", "delight"]
            if add_lora_path is not None:       # a caller-built pipeline: the extra adapters are registered on it the same way
                if add_lora_weights is None or len(add_lora_weights) != len(add_lora_path):
                    raise ValueError("add_lora_weights must give one weight per entry of add_lora_path")
                for i, src in enumerate(add_lora_path):
                    pipeline.load_lora_weights(src, adapter_name="add_lora_%d" % i)
                    names.append("add_lora_%d" % i); wt.append(float(add_lora_weights[i])); wd.append(float(add_lora_weights[i]))
        if num_inference_steps is not None:
            pipeline._num_inference_steps = num_inference_steps
        self.weights_for_texture, self.weights_for_delight, self.adapter_names = wt, wd, names
        self.pipeline_name = pipeline_name
        self.pipeline = pipeline
        self.video_exporter = VideoExporter(device=device, view_shard=shard, process_group=self.process_group)
        self.inverse_renderer = NVDiffRendererInverse(device=device, view_shard=shard, process_group=self.process_group)
        self.generator = torch.Generator().manual_seed(seed)   # ONE CPU generator shared by all draws (A19)
        self.super_resolutions = super_resolutions
        self.atlas_size = atlas_size
        # per-view resolution: 512 is the reference's hard-wired operating point (pipeline.py:239-255); 1024 is
        # BASELINE.json's configs[1..2] (joint strip 1024 x 6144, 50 688 tokens)
        self.view_size = int(view_size)
        # view count: 6 is the reference's hard-wired set (pipeline.py:206,239-255); 4 (f, r, b, l: export_nvdiffrast_video.py:931-932,
        # BASELINE configs[0]) and 8 (six + two upper diagonals, BASELINE configs[4]) are the builder-defined generalisations SURVEY 8a's
        # view-count caveat asks for -- same grid -> strip -> grid orchestration, cameras / priority from camera.generate_views_c2ws
        if n_views not in self.VIEW_LAYOUT:
            raise ValueError("n_views must be one of %s" % sorted(self.VIEW_LAYOUT))
        self.n_views = int(n_views)

    # view grid (rows x cols, tiles in the inverse renderer's view order) <-> 1 x n strip the DiT sees; `strip` lists the grid tiles in
    # strip order, `rot` is the grid tile turned by 180 degrees on the way (the 'down' view).  6 = reference (pipeline.py:239-255,279-288)
    VIEW_LAYOUT = {4: dict(rows=2, cols=2, strip=[0, 3, 1, 2], rot=None),
                   6: dict(rows=2, cols=3, strip=[0, 4, 1, 3, 2, 5], rot=5),
                   8: dict(rows=2, cols=4, strip=[0, 4, 1, 3, 2, 5, 6, 7], rot=5)}

    @CPUTimer("preprocess_blank_mesh")
    def preprocess_blank_mesh(self, save_dir, input_mesh_path, min_faces=20_000, max_faces=200_000, scale=0.95):
        """reference: open3d clean / decimate / subdivide / UVAtlas (geometry/uv/uv_atlas.py:131-194).  Here the
        host-side equivalents of texturetools/meshes.py: .obj / .glb / .gltf / .ply / .stl / .off in, rescaled to bbox*scale; a mesh with UVs passes
        through, one without is cleaned, brought into [min_faces, max_faces] and unwrapped (builder-defined atlas)."""
        from .texturetools import meshes
        # UV-less meshes get the chart unwrap: a handful of large charts at one texel density, as with the reference's UVAtlas charts
        # (the per-triangle grid of round 1 spent most of the atlas on gutters and flat-shaded nothing but lost resolution)
        verts, faces, uvs, faces_uv = meshes.prepare_blank_mesh(input_mesh_path, min_faces=min_faces, max_faces=max_faces, scale=scale,
                                                                atlas=self.atlas_size, gutter=4.0, unwrap="charts",
                                                                device=str(self.inverse_renderer.device))
        # shared positions + per-corner UVs: the condition render smooths normals over position indices (export_condition reads
        # them through load_obj), the inverse renderer splits per (v, vt) pair (load_mesh)
        meshes.save_obj(os.path.join(save_dir, "processed_mesh.obj"), verts, faces, uvs, faces_uv=faces_uv)

    @CPUTimer("preprocess_reference_image")
    def preprocess_reference_image(self, save_dir, input_image_path, scale=0.95, color="grey"):
        """reference: RMBG-2.0 matte, then crop to the matte's bbox, rescale to 0.95 of a 1024^2 frame on grey
        (pipeline.py:182-197, image/process_image.py:31-74; crop / rescale / paste pinned by fixture G10).  The matting
        model is [3p] and absent: the input's own alpha channel is the matte when it has one, else the whole frame."""
        from .texturetools.process_image import preprocess
        src = Image.open(input_image_path)
        src = src.resize((1024, 1024)) if src.mode == "RGBA" else src.convert("RGB").resize((1024, 1024))
        out = preprocess(src, alpha=None, H=1024, W=1024, scale=scale, color=color)
        out.save(os.path.join(save_dir, "rembg_image.png"), compress_level=PNG_LEVEL)
        out.convert("RGB").resize((512, 512)).save(os.path.join(save_dir, "processed_image.png"), compress_level=PNG_LEVEL)

    @CPUTimer("render_geometry_images")
    def render_geometry_images(self, save_dir, input_mesh_path, geometry_scale=0.95, scale=1.0, color="grey"):
        lay = self.VIEW_LAYOUT[self.n_views]
        out = self.video_exporter.export_condition(input_mesh_path, geometry_scale=geometry_scale, n_views=self.n_views, n_rows=lay["rows"], n_cols=lay["cols"],
                                                   H=self.view_size, W=self.view_size, fov_deg=49.1, scale=scale, perspective=False, orbit=False,
                                                   background=color, return_image=True, return_camera=True)
        with Encoders() as enc:
            for key in ("alpha", "ccm", "normal"):
                enc.submit(out[key].save, os.path.join(save_dir, "mv_%s.png" % key), compress_level=PNG_LEVEL)
        torch.save({"c2ws": out["c2ws"], "intrinsics": out["intrinsics"], "perspective": out["perspective"]},
                   os.path.join(save_dir, "camera_info.pth"))

    @CPUTimer("infer_mv")
    def infer_mv(self, save_dir, input_image_path, input_mv_image_path, add_input_mv_image_path):
        """reference pipeline.py:232-291: control = trunc(0.5*normal + 0.5*ccm) (A3); 2x3 grid frtbld -> 1x6 strip
        f,l,r,b,t,d with the 'down' tile rotated 180 deg (A2); texture pass then delight pass; inverse mapping."""
        reference_image = Image.open(input_image_path).convert("RGB")
        normal = np.array(Image.open(input_mv_image_path).convert("RGB"))
        ccm = np.array(Image.open(add_input_mv_image_path).convert("RGB"))
        steps = getattr(self.pipeline, "_num_inference_steps", 28)
        if self.pipeline_name != "texture_plus":
            raise NotImplementedError("pipeline_name %s is not supported" % self.pipeline_name)
        V = getattr(self, "view_size", 512)
        n = getattr(self, "n_views", 6)
        lay = RGBTextureFullPipelineBase.VIEW_LAYOUT[n]
        R, Cc, strip, rot = lay["rows"], lay["cols"], lay["strip"], lay["rot"]
        mix = (0.5 * normal.reshape(R, V, Cc, V, -1) + 0.5 * ccm.reshape(R, V, Cc, V, -1)).astype(np.uint8)
        if rot is not None:
            mix[rot // Cc, :, rot % Cc] = mix[rot // Cc, ::-1, rot % Cc, ::-1]
        tiles = mix.transpose(0, 2, 1, 3, 4).reshape(n, V, V, -1)[strip]
        control_image = Image.fromarray(tiles.transpose(1, 0, 2, 3).reshape(V, n * V, -1))
        common = dict(prompt="[MVFLUX]", prompt_embeds=None, pooled_prompt_embeds=None, height=V, width=n * V, n_rows=1,
                      n_cols=n, num_inference_steps=steps, guidance_scale=3.5, max_sequence_length=512, generator=self.generator)
        self.pipeline.set_adapters(adapter_names=self.adapter_names, adapter_weights=self.weights_for_texture)
        out_image = self.pipeline(control_image=control_image, dual_image=reference_image, **common).images[0]
        with Encoders(1) as enc:     # the lit strip is encoded while the delight pass runs (it reads the image, not the file)
            enc.submit(out_image.copy().save, os.path.join(save_dir, "mv_rgb_w_light.png"), compress_level=PNG_LEVEL)
            self.pipeline.set_adapters(adapter_names=self.adapter_names, adapter_weights=self.weights_for_delight)
            delit = self.pipeline(control_image=out_image, **common).images[0]
        t = np.array(delit).reshape(V, n, V, -1)
        if rot is not None:
            sp = strip.index(rot)                       # where the turned tile sits in the strip
            t[:, sp] = t[::-1, sp, ::-1]
        inv = [strip.index(g) for g in range(n)]        # strip position of every grid tile ([0, 2, 4, 3, 1, 5] for the reference's six)
        grid = t.transpose(1, 0, 2, 3)[inv].reshape(R, Cc, V, V, -1).transpose(0, 2, 1, 3, 4).reshape(R * V, Cc * V, -1)
        Image.fromarray(grid).save(os.path.join(save_dir, "mv_rgb.png"), compress_level=PNG_LEVEL)

    @CPUTimer("export_video")
    def export_video(self, save_dir, input_mesh_path, output_video_name):
        output_video_path = os.path.join(save_dir, os.path.splitext(output_video_name)[0] + ".mp4")
        self.video_exporter.export_orbit_video(input_mesh_path, output_video_path, n_frames=120, enhance_mode=None, perspective=True,
                                               video_type="rgb", save_frames=False, save_grid=False, save_cover=False,
                                               save_camera=False, rename_with_euler=False)

    @CPUTimer("reproject_and_query_field")
    def reproject_and_query_field(self, save_dir, input_mesh_path, input_mv_image_path, camera_info_path, four_or_six=False,
                                  flatten=False, method="reproject", inpainting=False):
        assert method in ["kdtree", "reproject"] and not four_or_six and not flatten
        img = np.asarray(Image.open(input_mv_image_path).convert("RGB"), dtype=np.float32) / 255.0   # image_to_tensor
        Hh, Ww, Cc = img.shape
        n = getattr(self, "n_views", 6)
        lay = RGBTextureFullPipelineBase.VIEW_LAYOUT[n]
        HP, WP = Hh // lay["rows"], Ww // lay["cols"]
        image_attrs = torch.from_numpy(img).reshape(lay["rows"], HP, lay["cols"], WP, Cc).permute(0, 2, 1, 3, 4).reshape(n, HP, WP, Cc)
        cam = torch.load(camera_info_path, weights_only=True, map_location="cpu")
        self.inverse_renderer.update_from_file(input_mesh_path)
        if n != 6:
            from .texturetools import camera as _cam
            self.inverse_renderer.index = list(_cam.generate_views_c2ws(n)[1])      # composite priority of the 4- / 8-view sets
        T = self.atlas_size
        textured, reprojected_uv, visable_mask, completed = self.inverse_renderer.infer(
            input_mesh_path, c2ws=cam["c2ws"], intrinsics=cam["intrinsics"], image_attrs=image_attrs, perspective=cam["perspective"],
            H=HP, W=WP, H2D=T, W2D=T, method=method, kdtree_n_neighbors=8, kdtree_n_neighbors_visiable=4, kdtree_inpainting=inpainting,
            reproject_inpainting=inpainting, grad_norm_threhold=0.15, ray_normal_angle_threhold=100,
            filt_gradient_points=inpainting)   # keyword set of the reference call, pipeline.py:333-348
        with Encoders() as enc:
            enc.submit(textured.export, os.path.join(save_dir, "textured_mesh.glb"))

            def save_mask(t, name):   # torchvision save_image: *255 + 0.5, clamp, uint8 [3p]
                a = (t.float().clamp(0, 1) * 255 + 0.5).clamp(0, 255).to(torch.uint8).cpu().numpy()
                enc.submit(Image.fromarray(a).save, os.path.join(save_dir, name), compress_level=PNG_LEVEL)
            save_mask(reprojected_uv.any(dim=0)[..., 0], "visable_uv_mask.png")
            save_mask(visable_mask[0, ..., 0], "valid_uv_mask.png")
            save_mask(completed[0], "completed_uv.png")
        self.inverse_renderer.clear()
        torch.cuda.empty_cache()


class RGBTextureFullPipeline(RGBTextureFullPipelineBase):
    def step_1_1(self, cache_dir, input_image_path, input_mesh_path, clear_cache=False, *args, **kwargs):
        print("step_1_1: %s, %s" % (input_image_path, input_mesh_path))
        self.preprocess_blank_mesh(cache_dir, input_mesh_path)
        self.preprocess_reference_image(cache_dir, input_image_path)
        # the geometry conditions (normal + CCM control image of the DiT) are rendered from the RAW input mesh, as the reference does
        # (pipeline.py:573): export_condition normalises it to the same bbox itself, and a mesh that preprocess_blank_mesh decimates or
        # subdivides still conditions the DiT on its original surface
        self.render_geometry_images(cache_dir, input_mesh_path)
        self.infer_mv(cache_dir, os.path.join(cache_dir, "processed_image.png"), os.path.join(cache_dir, "mv_normal.png"),
                      os.path.join(cache_dir, "mv_ccm.png"))

    def step_2_1(self, cache_dir, input_image_path, input_mesh_path, clear_cache=False, *args, **kwargs):
        self.reproject_and_query_field(cache_dir, os.path.join(cache_dir, "processed_mesh.obj"), os.path.join(cache_dir, "mv_rgb.png"),
                                       os.path.join(cache_dir, "camera_info.pth"), inpainting=False)
        if not clear_cache:     # pipeline.py:579-580
            self.export_video(cache_dir, os.path.join(cache_dir, "textured_mesh.glb"), "textured_mesh.mp4")

    step_seq = ["step_1_1", "step_2_1"]

    def __call__(self, save_dir: str, input_image_path: str, input_mesh_path: str, clear_cache=False) -> Tuple[str, str]:
        os.makedirs(save_dir, exist_ok=True)
        rank = getattr(self, "rank", 0)
        cache_dir = os.path.join(os.path.abspath(save_dir), "cache" if rank == 0 else "cache.rank%d" % rank)
        os.makedirs(cache_dir, exist_ok=True)
        for step in self.step_seq:
            getattr(self, step)(cache_dir=cache_dir, input_image_path=input_image_path, input_mesh_path=input_mesh_path,
                                clear_cache=clear_cache)
        if rank == 0:
            for name in ("rembg_image.png", "mv_rgb.png", "textured_mesh.glb"):
                shutil.copy(os.path.join(cache_dir, name), os.path.join(save_dir, name))
        if getattr(self, "world", 1) > 1:
            import torch.distributed as dist
            dist.barrier(group=self.process_group)      # rank 0's final artefacts exist when any rank returns
        if clear_cache:
            shutil.rmtree(cache_dir)
        return os.path.join(save_dir, "rembg_image.png"), os.path.join(save_dir, "textured_mesh.glb")


class CustomRGBTextureFullPipeline(RGBTextureFullPipeline):
    step_seq = ["step_1_1", "step_2_ablition"]

    def step_2_ablition(self, cache_dir, input_image_path, input_mesh_path, clear_cache=False, *args, **kwargs):
        os.makedirs(os.path.join(cache_dir, "wo_LTM"), exist_ok=True)
        os.makedirs(os.path.join(cache_dir, "w_LTM"), exist_ok=True)
        self.reproject_and_query_field(os.path.join(cache_dir, "wo_LTM"), os.path.join(cache_dir, "processed_mesh.obj"),
                                       os.path.join(cache_dir, "mv_rgb.png"), os.path.join(cache_dir, "camera_info.pth"), inpainting=False)
        self.export_video(os.path.join(cache_dir, "wo_LTM"), os.path.join(cache_dir, "wo_LTM/textured_mesh.glb"), "textured_mesh.mp4")
        shutil.copy(os.path.join(cache_dir, "wo_LTM/textured_mesh.glb"), os.path.join(cache_dir, "textured_mesh.glb"))
        print("The second stage (LTM part) is unreleased in the reference (pipeline.py:631).")
