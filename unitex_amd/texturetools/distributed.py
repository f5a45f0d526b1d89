"""View-sharded back-projection: the one exchange step of the path.

Each rank back-projects the views it owns (contiguous blocks of ceil(n/world) views), then ONE all-gather
assembles the per-view layers -- colour [T,3] f32 + visibility [T] u8 = 13 bytes per texel per view,
54.5 MB per view at a 2048^2 atlas -- after which every rank runs the (cheap, order-dependent) priority
composite redundantly.  xGMI is point-to-point: an all-gather lets every peer pair move its 54.5 MB on its
own direct link; a ring all-reduce would be the wrong shape (per-link bound and unnecessary).
Backend: RCCL ('nccl') on the GPUs; the same code runs over gloo on CPU tensors in the tests."""
import torch
import torch.distributed as dist


def view_range(rank, world, n_views):
    per = (n_views + world - 1) // world
    return min(rank * per, n_views), min((rank + 1) * per, n_views), per


def _all_gather(out, inp, group):
    """all_gather_into_tensor; gloo cannot take device tensors, so two ranks sharing one GPU in the tests stage through the
    host (production: NCCL = RCCL, device to device)."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def gather_view_layers(color, vis, rank, world, group=None):
    """color [n,H,W,3] f32, vis [n,H,W] u8: rank r has valid data in its own view_range only.
    Returns the fully populated (color, vis) on every rank."""
    n, H, W = vis.shape
    T = H * W
    v0, v1, per = view_range(rank, world, n)
    pay = torch.zeros(per, T * 13, dtype=torch.uint8, device=color.device)
    for j in range(v1 - v0):
        pay[j, : T * 12] = color[v0 + j].contiguous().reshape(-1).view(torch.uint8)
        pay[j, T * 12:] = vis[v0 + j].reshape(-1)
    allp = torch.empty(world * per, T * 13, dtype=torch.uint8, device=color.device)
    _all_gather(allp, pay, group)
    color_all = allp[:n, : T * 12].contiguous().view(torch.float32).view(n, H, W, 3)
    vis_all = allp[:n, T * 12:].contiguous().view(n, H, W)
    return color_all, vis_all


def gather_view_images(stack, rank, world, group=None):
    """stack [n, ...] uint8 (e.g. the condition renders [n,H,W,7] = normal rgb, ccm rgb, alpha): rank r rendered its own
    view_range only; ONE all-gather returns the complete stack on every rank (geometry-condition render)."""
    n = stack.shape[0]
    v0, v1, per = view_range(rank, world, n)
    flat = stack.reshape(n, -1)
    pay = torch.zeros(per, flat.shape[1], dtype=torch.uint8, device=stack.device)
    pay[: v1 - v0] = flat[v0:v1]
    allp = torch.empty(world * per, flat.shape[1], dtype=torch.uint8, device=stack.device)
    _all_gather(allp, pay, group)
    return allp[:n].reshape(stack.shape)
