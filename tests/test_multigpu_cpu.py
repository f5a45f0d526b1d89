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


@pytest.mark.parametrize("world", [2, 3, 4, 8])      # 8 ranks, 6 views: two ranks own no view and still take part in the ONE all-gather
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
    owned = [v for a, b in covered for v in range(a, b)]
    assert owned == list(range(6)), "every view owned by exactly one rank (ranks beyond the views own none): %s" % covered


def _ulysses_worker(rank, world, port, q):
    """sequence-parallel attention exchange (product comm code, oracle compute): token-sharded q/k/v -> all-to-all ->
    each rank attends with H/P heads over the FULL sequence -> all-to-all back; must equal unsharded attention."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import dit_ref
    from unitex_amd.flux.ulysses import UlyssesExchange, local_slice
    H, S_txt, S_img = (24 if world == 8 else 12), 64 * world, 128 * world      # world 8: FLUX's 24 heads -> 3 heads per rank, one head group (the 8-GPU target)
    g = torch.Generator().manual_seed(3)                     # same data on every rank
    S = S_txt + S_img
    qf, kf, vf = (torch.randn(H, S, 128, generator=g) for _ in range(3))
    t0, t1 = local_slice(S_txt, rank, world)
    i0, i1 = local_slice(S_img, rank, world)
    own = torch.cat([torch.arange(t0, t1), S_txt + torch.arange(i0, i1)])          # this rank's tokens: [txt_loc | img_loc]
    ref = dit_ref.sdpa(qf, kf, vf, em=False)[:, own].permute(1, 0, 2).reshape(own.numel(), -1)
    err = 0.0
    for G in [g for g in (1, 2, 3) if (H // world) % g == 0]:     # head groups whose exchanges are pipelined with attention (round 3)
        ex = UlyssesExchange(H, own.numel(), device="cpu", dtype=torch.float32, head_groups=G)
        assert ex.bytes_per_layer == 4 * own.numel() * H * 128 * 4
        ex.pack(qf[:, own].contiguous(), kf[:, own].contiguous(), vf[:, own].transpose(1, 2).contiguous())
        works = ex.start_heads_in()
        out = torch.full((own.numel(), 5 * H * 128), float("nan"))[:, : H * 128]       # strided rows, like the single-block cat buffer
        back = []
        for g in range(G):                                     # the product's order: group by group, return exchange started per group
            q_h, k_h, vt_h = ex.finish_heads_in_group(g, works[g])
            o = dit_ref.sdpa(q_h, k_h, vt_h.transpose(1, 2), em=False)              # [Hg, S, 128], rows = (src rank, local token)
            ex.o[g].copy_(o.permute(1, 0, 2).reshape(S, -1))
            back.append(ex.start_tokens_out_group(g))
        for g in range(G):
            ex.finish_tokens_out_group(g, back[g], out)
        err = max(err, (out - ref).abs().max().item())
        # the blocking forms (bench.py's exchange timing) do the same
        q2, k2, v2 = ex.heads_in(qf[:, own].contiguous(), kf[:, own].contiguous(), vf[:, own].transpose(1, 2).contiguous())
        ex.set_attention_output(dit_ref.sdpa(q2, k2, v2.transpose(1, 2), em=False))
        out2 = torch.empty(own.numel(), H * 128)
        ex.tokens_out(out2)
        err = max(err, (out2 - ref).abs().max().item())
    # round 6 -- every rank carries the SAME T text rows (the reference's identical text tokens, each key standing for w of them): the unpack keeps them once among the
    # keys (kv_text_rows), a rank attends with S = P S_loc queries over S_k = T + P (S_loc - T) keys in the single-GPU key order, key weight on the first T keys only
    import math
    T, w = 64, 4.0

    def sdpa_kw(qq, kk, vv):
        o = torch.empty_like(qq)
        for h in range(qq.shape[0]):
            sc = (qq[h] @ kk[h].t()) / math.sqrt(128.0)
            sc[:, :T] += math.log(w)
            o[h] = torch.softmax(sc, dim=-1) @ vv[h]
        return o
    seq = torch.cat([torch.arange(0, T), S_txt + torch.arange(0, S_img)])             # the de-duplicated sequence: T text rows + every image row
    ref_d = sdpa_kw(qf[:, seq], kf[:, seq], vf[:, seq])                                  # [H, T + S_img, 128]
    own_d = torch.cat([torch.arange(0, T), S_txt + torch.arange(i0, i1)])              # this rank's tokens: the shared text rows + its image slice
    ref_own = ref_d[:, torch.cat([torch.arange(0, T), T + torch.arange(i0, i1)])].permute(1, 0, 2).reshape(own_d.numel(), -1)
    for G in [g for g in (1, 2, 3) if (H // world) % g == 0]:
        ex = UlyssesExchange(H, own_d.numel(), device="cpu", dtype=torch.float32, head_groups=G, kv_text_rows=T)
        assert ex.S == world * own_d.numel() and ex.S_k == T + S_img and ex.k.shape == (H // world, T + S_img, 128) and ex.vt.shape == (H // world, 128, T + S_img)
        ex.pack(qf[:, own_d].contiguous(), kf[:, own_d].contiguous(), vf[:, own_d].transpose(1, 2).contiguous())
        works = ex.start_heads_in()
        out = torch.full((own_d.numel(), H * 128), float("nan"))
        back = []
        for g in range(G):
            q_h, k_h, vt_h = ex.finish_heads_in_group(g, works[g])
            # the keys ARE the single-GPU sequence, in its order
            assert torch.equal(k_h, kf[rank * (H // world) + g * ex.Hg: rank * (H // world) + (g + 1) * ex.Hg][:, seq])
            o = sdpa_kw(q_h, k_h, vt_h.transpose(1, 2))
            ex.o[g].copy_(o.permute(1, 0, 2).reshape(ex.S, -1))
            back.append(ex.start_tokens_out_group(g))
        for g in range(G):
            ex.finish_tokens_out_group(g, back[g], out)
        err = max(err, (out - ref_own).abs().max().item())
    q.put((rank, err < 1e-5, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
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


def test_sequence_parallel_model_bytes_overlap_window_and_predicted_scaling():
    """the arithmetic the first multi-GPU run is to be compared with (unitex_amd/flux/sp_model.py): bytes per rank and layer
    equal what UlyssesExchange moves; the exchanges of the G - 1 later head groups fit behind a group's attention at every P (so only the
    first-in / last-out share is exposed); with PER-DIRECTION xGMI bandwidth (76.5 GB/s per link x 65 %) the predicted speed-up at 4 ranks
    is >= 3.5 (BASELINE's target), and the un-pipelined round-2 form would not have reached it."""
    import ctypes as C
    from unitex_amd import _lib
    from unitex_amd.flux import sp_model
    from unitex_amd.flux.ulysses import UlyssesExchange, pick_head_groups
    lib = _lib.load_library()

    def plan(H, Sq, S, ncu):
        out = (C.c_int * 4)()
        assert lib.utx_attn_plan(H, Sq, S, ncu, C.byref(out)) == 0
        return list(out)
    rows = {r["P"]: r for r in sp_model.table(plan=plan)}
    for P in (2, 4, 8):
        r = rows[P]
        ex = UlyssesExchange(24, r["S_loc"], device="cpu", dtype=torch.bfloat16, head_groups=r["groups"])
        assert ex.P == 1                      # no process group here: the buffer arithmetic is what is checked
        # a rank keeps 1 / P of what it produces: (P - 1) / P of 4 S_loc D 2 B leaves it, 3 / 4 of that in exchange 1
        assert abs(r["bytes_per_rank_per_layer_on_fabric"] - 4 * r["S_loc"] * 3072 * 2 * (P - 1) / P) < 1
        assert r["bytes_per_peer_in"] == 3 * r["S_loc"] * (3072 // P) * 2
        assert r["groups"] == pick_head_groups(24 // P, r["S_loc"] * P, 256)
        # overlap window: the fabric time of a layer (both exchanges, all groups) is below the attention time of the layer
        assert r["exchange_in_ms"] + r["exchange_out_ms"] < r["attention_ms_per_layer"]
        assert r["exposed_comm_frac"] < 0.05
    assert rows[2]["groups"] == 4 and rows[4]["groups"] == 3 and rows[8]["groups"] == 1
    assert rows[4]["speedup"] >= 3.5, rows[4]
    assert rows[2]["speedup"] >= 1.8 and rows[8]["speedup"] >= 6.5
    # round 2's form (G = 1: both exchanges exposed in the double blocks, the return exchange everywhere) at 4 ranks
    unpip = sp_model.predict(4, groups=1, plan=plan)
    assert unpip["exposed_comm_ms"] > 3.5 * rows[4]["exposed_comm_ms"]
    # the attention launch of 3 heads at P = 8 is ONE launch whose third round is cut along the keys (utx_attn_plan), not 2.32 -> 3 whole rounds
    nwg, nfull, ns, tps = plan(3, 50688, 50688, 256)
    assert (nwg, nfull) == (594, 512) and ns == 3 and (nwg - nfull) * ns <= 256
    # the de-duplicated sequence-parallel launch has MORE queries than keys (P x 64 text rows among the queries, 64 among the keys): the plan and the scratch size
    # follow the query count -- a zero here would leave that launch without scratch, i.e. unsplit and on the 8 x 32 kernel
    for P in (2, 4, 8):
        S_q, S_k, Hg = P * 64 + 50176, 64 + 50176, 24 // P
        pq = plan(Hg, S_q, S_k, 256)
        assert pq[0] == Hg * ((S_q + 255) // 256) and pq[3] * pq[2] >= S_k // 64
        need = int(lib.utx_attn_workspace_bytes(None, Hg, S_q, S_k))
        flags = Hg * ((S_q + 255) // 256) * 4
        split = (pq[0] - pq[1]) * pq[2] * 256 * (128 * 2 + 4) if pq[2] > 1 else 0
        assert need >= split + flags > 0 and need - split - flags < 256
