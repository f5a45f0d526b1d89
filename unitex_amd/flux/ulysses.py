"""Head-parallel ("Ulysses") sequence parallelism for the joint-attention FLUX DiT (parity-preserving
multi-GPU split; the reference itself is single-GPU, flux_piplines/texturing/pipeline.py:633-681).

All six views are ONE token sequence, so the DiT cannot shard by view without changing the result.  What does shard
exactly: every per-token op (linears, norms, RoPE, MLPs, scheduler) over a 1/P slice of the tokens, and attention over
a 1/P slice of the HEADS.  Two all-to-alls per layer over RCCL / xGMI move between the two layouts:

    tokens-sharded  Q, K [H, S_loc, 128], V^T [H, 128, S_loc]    --all_to_all-->   heads-sharded [H/P, S, 128] / [H/P, 128, S]
    heads-sharded   O [S, (H/P)*128]                             --all_to_all-->   tokens-sharded [S_loc, H*128]

Data movement per layer and rank (GPU path):
  * send side, exchange 1: NONE -- `utx_qkv_post` writes Q, K, V^T straight into the send buffer [P][3][H/P][S_loc*128]
    through its grouped head addressing (one group of H/P heads per destination rank);
  * ONE `all_to_all_single` carries Q, K and V together (xGMI is point-to-point: every peer pair moves its block over its own
    link, all links busy at once);  it is started asynchronously so that independent work of the layer (the MLP half of the
    single-block projection) overlaps it (`start_heads_in` / `finish_heads_in`);
  * receive side: ONE HIP copy kernel (`utx_sp_unpack_qkv`) puts the P received blocks in the attention layout;
  * exchange 2: attention writes its output [S, (H/P)*128] = [P][S_loc][(H/P)*128], which already IS the send layout; after the
    all-to-all ONE HIP copy kernel (`utx_sp_unpack_o`) interleaves the P column blocks into the consumer's rows.
The gathered key order is (source rank, local token); attention is invariant to the key order and the query order is undone by
the return exchange, so the result equals the unsharded computation up to fp32 summation order inside the kernel.
Per rank and layer 4 * S_loc * D * 2 B cross the fabric ((P-1)/P of it off-chip): 156 MB at S = 50 688, P = 8.
Constraints: H % P == 0 and the local token count is a multiple of 64 (no padded keys inside the gathered sequence).

Backends: NCCL (= RCCL) device to device on the GPUs.  gloo is supported for tests only: CPU tensors (exchange logic against the
oracle, tests/test_multigpu_cpu.py: relayouts by torch copies) and two processes sharing one GPU (host-staged collective).
"""
import ctypes as C

import torch
import torch.distributed as dist


class UlyssesExchange:
    def __init__(self, H, S_loc, group=None, device="cpu", dtype=torch.bfloat16, ctx=None):
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        P = self.P
        if H % P:
            raise ValueError("number of heads %d is not divisible by the sequence-parallel degree %d" % (H, P))
        if S_loc % 64:
            raise ValueError("local token count %d must be a multiple of 64" % S_loc)
        self.H, self.Hp, self.S_loc, self.S = H, H // P, S_loc, S_loc * P
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.ctx = ctx                      # unitex_amd._lib.Context: the HIP unpack kernels (GPU path)
        if self.on_gpu and ctx is None:
            raise RuntimeError("UlyssesExchange on a GPU needs the library context (HIP unpack kernels)")
        backend = dist.get_backend(group) if (dist.is_initialized() and P > 1) else None
        # gloo cannot move device tensors through all_to_all: stage through the host (two ranks sharing one GPU in the tests --
        # production uses NCCL = RCCL, device to device over xGMI)
        self.host_staged = bool(P > 1 and self.on_gpu and backend == "gloo")
        self.can_async = bool(P > 1 and backend == "nccl")
        self.E = S_loc * 128
        n = self.Hp * self.E
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=device)
        self.send = z(P, 3, self.Hp, self.E)          # [dest rank][q|k|v][head of that rank][S_loc*128]
        self.recv = z(P, 3, self.Hp, self.E) if P > 1 else self.send
        self.q = z(self.Hp, self.S, 128)
        self.k = z(self.Hp, self.S, 128)
        self.vt = z(self.Hp, 128, self.S)
        self.o = z(self.S, self.Hp * 128)             # attention output = [P][S_loc][Hp*128]: the send buffer of exchange 2
        self.o_recv = z(P, S_loc, self.Hp * 128) if P > 1 else self.o.view(P, S_loc, self.Hp * 128)
        self.group_stride = 3 * n                     # elements between the head groups of two destination ranks
        self.bytes_per_layer = 4 * S_loc * H * 128 * self.send.element_size()

    # ---- send-side views: where utx_qkv_post writes (grouped head addressing: head h -> (h // Hp) * group_stride + (h % Hp) * E)
    def send_base(self, which):
        """flat view starting at the first element of Q (0) / K (1) / V^T (2) of destination rank 0, head 0."""
        return self.send.view(-1)[which * self.Hp * self.E:]

    def pack(self, Qh, Kh, Vt):
        """torch relayout of plain head-major tensors into the send buffer -- CPU tests only (on the GPU utx_qkv_post writes the
        send buffer directly).  Qh, Kh [H, S_loc, 128], Vt [H, 128, S_loc]."""
        P, Hp, S_loc = self.P, self.Hp, self.S_loc
        self.send[:, 0].view(P, Hp, S_loc, 128).copy_(Qh.reshape(P, Hp, S_loc, 128))
        self.send[:, 1].view(P, Hp, S_loc, 128).copy_(Kh.reshape(P, Hp, S_loc, 128))
        self.send[:, 2].view(P, Hp, 128, S_loc).copy_(Vt.reshape(P, Hp, 128, S_loc))

    def _a2a(self, out, inp, async_op=False):
        if self.P == 1:
            return None
        if self.host_staged:
            rc, sc = torch.empty_like(out, device="cpu"), inp.cpu()
            dist.all_to_all_single(rc, sc, group=self.group)
            out.copy_(rc)
            return None
        if async_op and self.can_async:
            return dist.all_to_all_single(out, inp, group=self.group, async_op=True)
        dist.all_to_all_single(out, inp, group=self.group)
        return None

    # ---- exchange 1
    def start_heads_in(self):
        """launch the Q/K/V all-to-all (asynchronously on NCCL: kernels issued on the current stream before finish_heads_in
        run beside it).  Returns a handle for finish_heads_in."""
        return self._a2a(self.recv, self.send, async_op=True)

    def finish_heads_in(self, work=None, stream=None):
        """wait for the exchange, then ONE relayout pass: recv [P][3][Hp][E] -> self.q, self.k [Hp, S, 128], self.vt [Hp, 128, S]."""
        if work is not None:
            work.wait()                                   # the current stream waits for the collective; the host does not block
        P, Hp, S_loc = self.P, self.Hp, self.S_loc
        r = self.recv
        if self.on_gpu:
            lib, h = self.ctx.lib, self.ctx.handle
            st = self.ctx.stream() if stream is None else stream
            self.ctx.check(lib.utx_sp_unpack_qkv(h, C.c_void_p(r.data_ptr()), P, Hp, S_loc, C.c_void_p(self.q.data_ptr()),
                                                 C.c_void_p(self.k.data_ptr()), C.c_void_p(self.vt.data_ptr()), st))
        else:
            self.q.view(Hp, P, S_loc, 128).copy_(r[:, 0].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
            self.k.view(Hp, P, S_loc, 128).copy_(r[:, 1].view(P, Hp, S_loc, 128).permute(1, 0, 2, 3))
            self.vt.view(Hp, 128, P, S_loc).copy_(r[:, 2].view(P, Hp, 128, S_loc).permute(1, 2, 0, 3))
        return self.q, self.k, self.vt

    def heads_in(self, Qh=None, Kh=None, Vt=None):
        """blocking form.  With arguments (CPU tests): pack them first; without: the send buffer has been written in place."""
        if Qh is not None:
            self.pack(Qh, Kh, Vt)
        return self.finish_heads_in(self.start_heads_in())

    # ---- exchange 2
    def tokens_out(self, out, stream=None):
        """self.o [S, (H/P)*128] (rows ordered (source rank, local token))  ->  out [S_loc, >= H*128] (rows may be strided)."""
        P, Hp, S_loc = self.P, self.Hp, self.S_loc
        self._a2a(self.o_recv, self.o.view(P, S_loc, Hp * 128))
        r = self.o_recv
        if self.on_gpu:
            lib, h = self.ctx.lib, self.ctx.handle
            st = self.ctx.stream() if stream is None else stream
            self.ctx.check(lib.utx_sp_unpack_o(h, C.c_void_p(r.data_ptr()), P, Hp, S_loc, C.c_void_p(out.data_ptr()), out.stride(0), st))
        else:
            out[:, : P * Hp * 128].unflatten(1, (P, Hp * 128)).copy_(r.permute(1, 0, 2))
        return out


def local_slice(n_total, rank, world):
    """contiguous 1/world slice of n_total tokens owned by `rank`."""
    if n_total % world:
        raise ValueError("%d tokens do not split evenly over %d ranks" % (n_total, world))
    per = n_total // world
    return rank * per, (rank + 1) * per
