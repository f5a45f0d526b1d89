"""Head-parallel ("Ulysses") sequence parallelism for the joint-attention FLUX DiT (parity-preserving
multi-GPU option; the reference itself is single-GPU, flux_piplines/texturing/pipeline.py:633-681).

All six views are ONE token sequence, so the DiT cannot shard by view without changing the result.  What does shard
exactly: every per-token op (linears, norms, RoPE, MLPs, scheduler) over a 1/P slice of the tokens, and attention over
a 1/P slice of the HEADS.  Two all-to-alls per layer over RCCL / xGMI move between the two layouts:

    tokens-sharded  Qh, Kh [H, S_loc, 128], Vt [H, 128, S_loc]      --all_to_all-->   heads-sharded [H/P, S, 128] / [H/P, 128, S]
    heads-sharded   O [S, (H/P)*128]                                --all_to_all-->   tokens-sharded [S_loc, H*128]

Q, K and V travel in ONE collective (packed send buffer), the attention output in a second one.  The gathered key
order is (source rank, local token); attention is invariant to the key order and the query order is undone by the
return exchange, so the result equals the unsharded computation up to fp32 summation order inside the kernel.
Per rank and layer 4 * S_loc * D * 2 B cross the fabric ((P-1)/P of it off-chip): 156 MB at S = 50 688, P = 8.
Constraints: H % P == 0 and the local token count is a multiple of 64 (no padded keys inside the gathered sequence).

This module is device-agnostic torch + torch.distributed (NCCL = RCCL on the GPUs, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


class UlyssesExchange:
    def __init__(self, H, S_loc, group=None, device="cpu", dtype=torch.bfloat16):
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        P = self.P
        if H % P:
            raise ValueError("number of heads %d is not divisible by the sequence-parallel degree %d" % (H, P))
        if S_loc % 64:
            raise ValueError("local token count %d must be a multiple of 64" % S_loc)
        self.H, self.Hp, self.S_loc, self.S = H, H // P, S_loc, S_loc * P
        # gloo cannot move device tensors through all_to_all: stage through the host (test configurations only --
        # production uses NCCL = RCCL, device to device over xGMI)
        self.host_staged = bool(P > 1 and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo")
        n = self.Hp * S_loc * 128
        z = lambda *s: torch.empty(*s, dtype=dtype, device=device)
        self.send = z(P, 3, n)
        self.recv = z(P, 3, n)
        self.q = z(self.Hp, self.S, 128)
        self.k = z(self.Hp, self.S, 128)
        self.vt = z(self.Hp, 128, self.S)
        self.o = z(self.S, self.Hp * 128)
        self.o_recv = z(P, S_loc, self.Hp * 128)

    def heads_in(self, Qh, Kh, Vt):
        """local Qh, Kh [H, S_loc, 128], Vt [H, 128, S_loc]  ->  self.q, self.k [H/P, S, 128], self.vt [H/P, 128, S]."""
        P, Hp, S_loc = self.P, self.Hp, self.S_loc
        self.send[:, 0].view(P, Hp, S_loc, 128).copy_(Qh.view(P, Hp, S_loc, 128))
        self.send[:, 1].view(P, Hp, S_loc, 128).copy_(Kh.view(P, Hp, S_loc, 128))
        self.send[:, 2].view(P, Hp, 128, S_loc).copy_(Vt.view(P, Hp, 128, S_loc))
        if P > 1 and self.host_staged:
            rc, sc = torch.empty_like(self.recv, device="cpu"), self.send.cpu()
            dist.all_to_all_single(rc, sc, group=self.group)
            self.recv.copy_(rc)
            r = self.recv
        elif P > 1:
            dist.all_to_all_single(self.recv, self.send, group=self.group)
            r = self.recv
        else:
            r = self.send
        self.q.view(Hp, P, S_loc, 128).copy_(r[:, 0].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
        self.k.view(Hp, P, S_loc, 128).copy_(r[:, 1].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
        self.vt.view(Hp, 128, P, S_loc).copy_(r[:, 2].view(P, Hp, 128, S_loc).permute(1, 2, 0, 3))
        return self.q, self.k, self.vt

    def tokens_out(self, out):
        """self.o [S, (H/P)*128] (rows ordered (source rank, local token))  ->  out [S_loc, H*128] (rows may be strided)."""
        P, Hp, S_loc = self.P, self.Hp, self.S_loc
        src = self.o.view(P, S_loc, Hp * 128)
        if P > 1 and self.host_staged:
            rc, sc = torch.empty_like(self.o_recv, device="cpu"), src.cpu()
            dist.all_to_all_single(rc, sc, group=self.group)
            self.o_recv.copy_(rc)
            r = self.o_recv
        elif P > 1:
            dist.all_to_all_single(self.o_recv, src, group=self.group)
            r = self.o_recv
        else:
            r = src
        out.unflatten(1, (P, Hp * 128)).copy_(r.permute(1, 0, 2))
        return out


def local_slice(n_total, rank, world):
    """contiguous 1/world slice of n_total tokens owned by `rank`."""
    if n_total % world:
        raise ValueError("%d tokens do not split evenly over %d ranks" % (n_total, world))
    per = n_total // world
    return rank * per, (rank + 1) * per
