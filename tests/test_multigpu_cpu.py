"""world_size-2/3 gloo tests (CPU) of the N>1 path: view-sharded back-projection layers + ONE all-gather +
replicated composite must reproduce the single-rank atlas bit for bit (oracle compute, product comm code)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import geom_ref as G
    from unitex_amd.texturetools.distributed import gather_view_layers, view_range
    rng = np.random.default_rng(5)   # same data on every rank
    n, T = 6, 48
    color = rng.uniform(0, 1, (n, T, T, 3)).astype(np.float32)
    vis = (rng.uniform(0, 1, (n, T, T)) > 0.6)
    v0, v1, per = view_range(rank, world, n)
    c_loc = np.zeros_like(color); v_loc = np.zeros((n, T, T), np.uint8)
    c_loc[v0:v1], v_loc[v0:v1] = color[v0:v1], vis[v0:v1]      # this rank only produced its own views
    c_all, v_all = gather_view_layers(torch.from_numpy(c_loc), torch.from_numpy(v_loc), rank, world)
    atlas, seen, win, _ = G.composite(c_all.numpy(), v_all.numpy().astype(bool))
    ref_atlas, _, ref_win, _ = G.composite(color, vis)
    ok = np.array_equal(c_all.numpy(), color) and np.array_equal(v_all.numpy().astype(bool), vis) and \
        np.array_equal(atlas, ref_atlas) and np.array_equal(win, ref_win)
    # geometry-condition render: per-view uint8 images (normal rgb, ccm rgb, alpha), same sharding, one all-gather
    from unitex_amd.texturetools.distributed import gather_view_images
    imgs = rng.integers(0, 256, (n, 16, 24, 7), dtype=np.uint8)
    loc = np.zeros_like(imgs); loc[v0:v1] = imgs[v0:v1]
    ok = ok and np.array_equal(gather_view_images(torch.from_numpy(loc), rank, world).numpy(), imgs)
    q.put((rank, bool(ok), (v0, v1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_view_sharded_layers_allgather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    covered = sorted(rg for _, _, rg in res)
    assert covered[0][0] == 0 and covered[-1][1] == 6


def _ulysses_worker(rank, world, port, q):
    """sequence-parallel attention exchange (product comm code, oracle compute): token-sharded q/k/v -> all-to-all ->
    each rank attends with H/P heads over the FULL sequence -> all-to-all back; must equal unsharded attention."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import dit_ref
    from unitex_amd.flux.ulysses import UlyssesExchange, local_slice
    H, S_txt, S_img = 12, 64 * world, 128 * world
    g = torch.Generator().manual_seed(3)                     # same data on every rank
    S = S_txt + S_img
    qf, kf, vf = (torch.randn(H, S, 128, generator=g) for _ in range(3))
    t0, t1 = local_slice(S_txt, rank, world)
    i0, i1 = local_slice(S_img, rank, world)
    own = torch.cat([torch.arange(t0, t1), S_txt + torch.arange(i0, i1)])          # this rank's tokens: [txt_loc | img_loc]
    ex = UlyssesExchange(H, own.numel(), device="cpu", dtype=torch.float32)
    q_h, k_h, vt_h = ex.heads_in(qf[:, own].contiguous(), kf[:, own].contiguous(), vf[:, own].transpose(1, 2).contiguous())
    o = dit_ref.sdpa(q_h, k_h, vt_h.transpose(1, 2), em=False)                     # [H/P, S, 128], rows = (src rank, local token)
    ex.o.copy_(o.permute(1, 0, 2).reshape(S, -1))
    out = torch.empty(own.numel(), 5 * H * 128)[:, : H * 128]                      # strided rows, like the single-block cat buffer
    ex.tokens_out(out)
    ref = dit_ref.sdpa(qf, kf, vf, em=False)[:, own].permute(1, 0, 2).reshape(own.numel(), -1)
    err = (out - ref).abs().max().item()
    q.put((rank, err < 1e-5, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ulysses_exchange_matches_unsharded_attention(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_ulysses_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
