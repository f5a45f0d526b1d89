"""Timing harness for the back-projection stage (SURVEY 8d metric (ii): s / mesh-texture, geometry part).

Synthetic inputs as SURVEY 8d prescribes: a UV-mapped bumpy sphere at a requested face count, the six
orthographic box views, seeded smooth view images.  Every stage of NVDiffRendererInverse.infer is bracketed
by HIP events on the launch stream (renderer_inverse._stage); algorithmic HBM bytes per stage are the
compulsory reads + writes of that stage's tensors (each counted once), stated below per texel.
"""
import numpy as np
import torch

from . import camera, meshes, ops
from .renderer_inverse import NVDiffRendererInverse

# compulsory bytes per atlas texel (T texels, n views), fp32 unless noted:
#   uv_raster        : rast record write 16 + 64-bit depth/id word write+read 16
#   backproject      : rast read 16 + per view (colour 12 + rayvis 1 + alpha 1) write
#   dilate_visibility: per view rayvis+alpha read 2 + vis write 1 ; rast read 16
#   composite        : per view colour 12 + vis 1 read ; atlas 12 + winner 1 write
#   seam_mask        : winner 1 + rast 16 read ; seam 1 write
#   nn_fill          : pos interpolate (rast 16 read, pos 12 write) + pos 12 / winner 1 read + atlas 12 r/w (grid cells: small)
#   lens_blur_seam   : atlas 12 read + 12 write + seam 1
#   pull_push        : 4/3 * (16 read + 16 write) down + the same up
#   to_u8            : 12 read + 3 write
def stage_bytes_per_texel(n_views):
    n = n_views
    return {
        "uv_raster": 32.0, "backproject": 16.0 + 14.0 * n, "dilate_visibility": 16.0 + 3.0 * n,
        "composite": 13.0 * n + 13.0, "seam_mask": 18.0, "nn_fill": 28.0 + 13.0 + 24.0,
        "lens_blur_seam": 25.0, "pull_push": 4.0 / 3.0 * 64.0, "to_u8": 15.0,
    }


def smooth_views(n, H, W, seed=7):
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, H, dtype=np.float32), np.linspace(0, 1, W, dtype=np.float32), indexing="ij")
    out = np.empty((n, H, W, 3), np.float32)
    for v in range(n):
        a = g.uniform(1.0, 6.0, size=(3, 2)).astype(np.float32)
        p = g.uniform(0.0, 6.28, size=3).astype(np.float32)
        for c in range(3):
            out[v, ..., c] = 0.5 + 0.5 * np.sin(a[c, 0] * 6.28 * xx + a[c, 1] * 6.28 * yy + p[c])
    return out


def time_backprojection(n_faces=50000, view_px=1024, atlas_px=2048, iters=3, warmup=1, device="cuda:0", view_shard=(0, 1),
                        process_group=None, n_views=6):
    """returns {"total_ms", "stages_ms": {...}, "stages_gbps": {...}, "faces", "texels", "covered_frac"} averaged over iters.
    view_shard=(rank, world): this rank back-projects its block of the views; the stage list then carries "all_gather"."""
    verts, faces, uvs = meshes.sphere_with_faces(n_faces)
    inv = NVDiffRendererInverse(device=device, view_shard=view_shard, process_group=process_group).update_from_arrays(verts, faces, uvs)
    if n_views == 6:
        c2ws = camera.generate_box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]
    else:
        c2ws, order = camera.generate_views_c2ws(n_views, 2.8)
        inv.index = order
    intr = camera.generate_intrinsics(1.0, 1.0, fov=False)
    images = torch.from_numpy(smooth_views(n_views, view_px, view_px)).to(device)
    acc, totals, host_enq = {}, [], []
    covered = 0.0
    for it in range(warmup + iters):
        inv.pbr_mesh._bvh = None          # the LBVH is rebuilt per mesh; count it
        inv.stage_events = []
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        out = inv.infer(None, c2ws=c2ws, intrinsics=intr, image_attrs=images, perspective=False, H=view_px, W=view_px,
                        H2D=atlas_px, W2D=atlas_px, filt_gradient_points=False, ray_normal_angle_threhold=100.0)
        t1.record()
        torch.cuda.synchronize()
        if it < warmup:
            continue
        totals.append(t0.elapsed_time(t1))
        host_enq.append(inv.host_enqueue_ms)
        for name, a, b in inv.stage_events:
            acc.setdefault(name, []).append(a.elapsed_time(b))
        covered = float(out[2].float().mean())
    inv.stage_events = None
    # nodes visited per ray -- the visibility rays of the first view of this rank, re-cast through the counting trace
    nodes_per_ray, depth = None, None
    try:
        m, rast2d = inv.pbr_mesh, inv.last["rast2d"]
        cov = rast2d[..., 3] > 0
        pos = ops.interpolate(m.vertices, rast2d, m.faces)[cov]
        from .distributed import view_range
        v0 = view_range(view_shard[0], view_shard[1], n_views)[0]
        d = (-torch.as_tensor(c2ws)[min(v0, n_views - 1), :3, 2]).to(pos.device, torch.float32)
        ro = pos - 2.0 * (3.0 ** 0.5) * d
        bvh = m.optix
        _, visited = bvh.trace_count(ro, d.expand_as(ro).contiguous())
        nodes_per_ray, depth = visited / max(1, ro.shape[0]), bvh.depth()
    except Exception as e:  # noqa: BLE001 -- a diagnostic must not break the timing
        nodes_per_ray = "error: %r" % (e,)
    stages = {k: float(np.mean(v)) for k, v in acc.items()}
    T = float(atlas_px * atlas_px)
    bpt = stage_bytes_per_texel(n_views)
    gbps = {k: bpt[k] * T / (stages[k] * 1e-3) / 1e9 for k in stages if k in bpt and stages[k] > 0}
    # total_ms brackets infer() with events on the launch stream: the stage chain + whatever the GPU idled waiting for the host.  kernel_sum_ms = the sum of the stage
    # brackets (each bracket holds its kernels and the gaps between them; the gaps BETWEEN stages -- torch allocations, the next stage's first launch -- are not in it);
    # host_enqueue_ms = the host's wall time up to the chain's last enqueue.  The timed iterations reuse the mesh's cached tree, so it holds no device wait here; on a FRESH tree the
    # first back-projection launch waits for the build's depth word (renderer_inverse.py), and the figure then includes that GPU time.  When it approaches total_ms with a cached tree the
    # chain is host-bound on that box.
    return {"total_ms": float(np.mean(totals)), "kernel_sum_ms": float(sum(stages.values())), "host_enqueue_ms": float(np.mean(host_enq)), "stages_ms": stages, "stages_gbps": gbps, "faces": int(len(faces)),
            "texels": int(T), "covered_frac": covered, "view_px": view_px, "atlas_px": atlas_px, "nodes_per_ray": nodes_per_ray, "bvh_depth": depth}
