"""Head-parallel ("Ulysses") sequence parallelism for the joint-attention FLUX DiT (parity-preserving
multi-GPU split; the reference itself is single-GPU, flux_piplines/texturing/pipeline.py:633-681).

All six views are ONE token sequence, so the DiT cannot shard by view without changing the result.  What does shard
exactly: every per-token op (linears, norms, RoPE, MLPs, scheduler) over a 1/P slice of the tokens, and attention over
a 1/P slice of the HEADS.  Two all-to-alls per layer over RCCL / xGMI move between the two layouts:

    tokens-sharded  Q, K [H, S_loc, 128], V^T [H, 128, S_loc]    --all_to_all-->   heads-sharded [H/P, S, 128] / [H/P, 128, S]
    heads-sharded   O [S, (H/P)*128]                             --all_to_all-->   tokens-sharded [S_loc, H*128]

Round 3 -- both exchanges are PIPELINED WITH ATTENTION over head groups.  The H/P heads of a rank are cut into G groups of
Hg heads; every buffer is group-major, so a group's all-to-all moves one contiguous block per peer:

    send / recv  [G][P][3][Hg][S_loc*128]      o (attention output) [G][S][Hg*128]      o_recv [G][P][S_loc][Hg*128]

    layer:   qkv_post -> send           (written in place: two-level grouped head addressing, no pack pass)
             start a2a_in(0..G-1)       (asynchronous, in order, on RCCL's stream)
             for g: wait a2a_in(g); unpack(g); attention(g)  ||  a2a_in(g+1..) and a2a_out(..g-1) on the fabric
                    start a2a_out(g)
             for g: wait a2a_out(g); unpack_o(g)
    exposed communication per layer = a2a_in(0) + a2a_out(G-1) = 1/G of the two exchanges (+ whatever the fabric cannot hide
    behind G-1 groups of attention: at S = 50 688 a group's attention is 4-8 x its two exchanges -- DESIGN 7, sp_model.py);
    un-pipelined (G = 1, round 2) all of both exchanges was exposed in the 19 double blocks and the return exchange in all 57.
G is chosen so that a group's attention launch still fills the chip (`pick_head_groups`): 4 at P = 2, 3 at P = 4, 1 at P = 8
(3 heads x 198 query blocks are 2.3 rounds of 256 CUs as ONE launch -- the key-split tail round of the attention kernel balances
the third -- and would be three under-filled rounds as three launches).

Data movement per layer and rank (GPU path):
  * send side, exchange 1: NONE -- `utx_qkv_post` writes Q, K, V^T straight into the send buffer;
  * ONE `all_to_all_single` per head group carries its Q, K and V together (xGMI is point-to-point: every peer pair moves its block
    over its own link, all links busy at once); in the 38 single blocks the exchanges also run beside the MLP half of the projection;
  * receive side: ONE HIP copy kernel per group (`utx_sp_unpack_qkv`) puts the P received blocks in the attention layout;
  * exchange 2: attention writes a group's output [S, Hg*128] = [P][S_loc][Hg*128], which already IS the send layout; after the
    all-to-all ONE HIP copy kernel (`utx_sp_unpack_o_cols`) interleaves the P column blocks into the consumer's rows.
The gathered key order is (source rank, local token); attention is invariant to the key order and the query order is undone by
the return exchange, so the result equals the unsharded computation up to fp32 summation order inside the kernel.
Round 6 -- `kv_text_rows` = T > 0 (the transformer's default when every rank carries the same T identical text rows): the unpack keeps those
rows ONCE in K / V^T (from source rank 0) followed by the ranks' other tokens in rank order: S_k = T + P (S_loc - T) keys in exactly the
single-GPU key order, so a rank's attention launch is the single-GPU launch (key multiplicity on tile 0 only -- the 4 x 64 kernel takes it,
+10 % over the 8 x 32 loop the periodic-multiplicity launches ran) over H / P heads, S = P S_loc queries over S_k keys, and its rows are
bit-identical to the single-GPU attention's on the same operands.  "The same rows": the ranks compute their text rows from identical inputs with
identical kernels; the one place where two ranks' copies can part by a bf16 ulp is the attention output of a text tile that falls into the
key-split tail round of a launch while another rank's falls into a full round (same softmax, other summation order) -- rank 0's copy stands
for all, a deviation of the size every other summation-order difference of the sharded run has (tests/test_dit_ops_gpu.py: two / four ranks
against the unsharded forward).
Per rank and layer 4 * S_loc * D * 2 B cross the fabric ((P-1)/P of it off-chip): 156 MB at S = 50 688, P = 8.
Constraints: H % P == 0 and the local token count is a multiple of 64 (no padded keys inside the gathered sequence).

Backends: NCCL (= RCCL) device to device on the GPUs.  gloo is supported for tests only: CPU tensors (exchange logic against the
oracle, tests/test_multigpu_cpu.py: relayouts by torch copies) and two processes sharing one GPU (host-staged collective).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def pick_head_groups(Hp, S, n_cus=256, max_groups=4):
    """head groups per rank for the pipelined exchanges: the largest G <= max_groups dividing Hp whose attention launch
    (Hp / G heads x ceil(S / 256) query blocks) still covers >= 1.5 rounds of the chip's CUs; 1 = no pipelining.
    UTX_SP_GROUPS overrides (tests; must divide Hp)."""
    env = os.environ.get("UTX_SP_GROUPS")
    if env:
        g = int(env)
        if g < 1 or Hp % g:
            raise ValueError("UTX_SP_GROUPS=%s does not divide the %d heads of a rank" % (env, Hp))
        return g
    nqb = (S + 255) // 256
    for g in range(min(max_groups, Hp), 1, -1):
        if Hp % g == 0 and (Hp // g) * nqb >= 1.5 * n_cus:
            return g
    return 1


class UlyssesExchange:
    def __init__(self, H, S_loc, group=None, device="cpu", dtype=torch.bfloat16, ctx=None, head_groups=None, n_cus=256, kv_text_rows=0):
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        P = self.P
        if H % P:
            raise ValueError("number of heads %d is not divisible by the sequence-parallel degree %d" % (H, P))
        if S_loc % 64:
            raise ValueError("local token count %d must be a multiple of 64" % S_loc)
        self.H, self.Hp, self.S_loc, self.S = H, H // P, S_loc, S_loc * P
        # keys kept by the unpack: all of them, or the ranks' identical leading text rows once + everything else (module docstring, round 6)
        self.kv_text_rows = int(kv_text_rows)
        if self.kv_text_rows and (self.kv_text_rows % 64 or not 0 < self.kv_text_rows < S_loc):
            raise ValueError("kv_text_rows = %d must be a multiple of 64 below the local token count %d" % (self.kv_text_rows, S_loc))
        self.S_k = self.kv_text_rows + P * (S_loc - self.kv_text_rows)
        self.G = int(head_groups) if head_groups else pick_head_groups(self.Hp, self.S, n_cus)
        if self.Hp % self.G:
            raise ValueError("%d head groups do not divide the %d heads of a rank" % (self.G, self.Hp))
        self.Hg = self.Hp // self.G
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.ctx = ctx                      # unitex_amd._lib.Context: the HIP unpack kernels (GPU path)
        if self.on_gpu and ctx is None:
            raise RuntimeError("UlyssesExchange on a GPU needs the library context (HIP unpack kernels)")
        # UTX_SP_FORCE_A2A=1 (tests): a 1-rank group still goes through the collectives (separate receive buffers, asynchronous all-to-alls to itself) -- the
        # only way to run the RCCL call pattern and its stream ordering against the HIP kernels on a box with one GPU
        self.force = bool(P == 1 and dist.is_initialized() and os.environ.get("UTX_SP_FORCE_A2A", "0") == "1")
        backend = dist.get_backend(group) if (dist.is_initialized() and (P > 1 or self.force)) else None
        # gloo cannot move device tensors through all_to_all: stage through the host (two ranks sharing one GPU in the tests --
        # production uses NCCL = RCCL, device to device over xGMI)
        self.host_staged = bool((P > 1 or self.force) and self.on_gpu and backend == "gloo")
        self.can_async = bool((P > 1 or self.force) and backend == "nccl")
        self.E = S_loc * 128
        G, Hg = self.G, self.Hg
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=device)
        self.send = z(G, P, 3, Hg, self.E)            # [head group][dest rank][q|k|v][head of the group][S_loc*128]
        self.recv = z(G, P, 3, Hg, self.E) if (P > 1 or self.force) else self.send
        # zero copy (GPU, opt-in: UTX_SP_ZERO_COPY=1): the attention kernel reads Q / K / V^T where the all-to-all put them -- blocks of S_loc tokens per
        # source rank, 3 Hg E elements apart (utx_attn_fwd_bf16_blk) -- and the relayout pass disappears.  Built, bit-identical, and MEASURED A NET LOSS
        # (profiles/r03_attn_blk_ab.log, r03_bench_sp_self_test_zero_copy / _relayout): attention on the block-strided operands runs 2.1-2.4 % slower at
        # 2 / 4 / 8 blocks (+0.6 % with one block: the cursor arithmetic; the rest is the V^T rows lying S_loc instead of S columns apart), i.e. +0.16 ms per
        # layer at 4 ranks against 0.09 ms of relayout saved.  The default keeps the relayout into head-major tensors.
        self.zero_copy = bool(self.on_gpu and os.environ.get("UTX_SP_ZERO_COPY", "0") == "1")
        if self.zero_copy and self.kv_text_rows:
            raise ValueError("the zero-copy exchange reads the receive buffer as it lies: no key de-duplication (kv_text_rows) with UTX_SP_ZERO_COPY=1")
        self._z = z
        self._qkv = None                              # head-major copies (relayout form / CPU): allocated on first use
        self.o = z(G, self.S, Hg * 128)               # attention output of group g = [P][S_loc][Hg*128]: the send buffer of its exchange 2
        self.o_recv = z(G, P, S_loc, Hg * 128) if (P > 1 or self.force) else self.o.view(G, P, S_loc, Hg * 128)
        # utx_qkv_post's two-level grouped head addressing into `send`: head h -> dest h // Hp, group (h % Hp) // Hg, head h % Hg
        self.dest_stride = 3 * Hg * self.E            # elements between the blocks of two destination ranks (inside a head group)
        self.group_stride = P * 3 * Hg * self.E       # elements between two head groups
        self.bytes_per_layer = 4 * S_loc * H * 128 * self.send.element_size()

    def _heads(self):
        if self._qkv is None:
            z = self._z
            self._qkv = (z(self.Hp, self.S, 128), z(self.Hp, self.S_k, 128), z(self.Hp, 128, self.S_k))     # heads of group g = rows [g*Hg, (g+1)*Hg)
        return self._qkv

    q = property(lambda self: self._heads()[0])
    k = property(lambda self: self._heads()[1])
    vt = property(lambda self: self._heads()[2])

    def heads_blocks(self, g):
        """zero-copy form of group g's Q / K / V^T for utx_attn_fwd_bf16_blk: (q, k, vt device pointers of head 0 / source rank 0, head stride, block stride,
        block rows) -- receive buffer recv[g] = [source rank][q | k | v][head][S_loc * 128]."""
        r = self.recv[g]
        return r[0, 0].data_ptr(), r[0, 1].data_ptr(), r[0, 2].data_ptr(), self.E, 3 * self.Hg * self.E, self.S_loc

    # ---- send-side views: where utx_qkv_post writes
    def send_base(self, which):
        """flat view starting at the first element of Q (0) / K (1) / V^T (2) of head group 0, destination rank 0, head 0."""
        return self.send.view(-1)[which * self.Hg * self.E:]

    def pack(self, Qh, Kh, Vt):
        """torch relayout of plain head-major tensors into the send buffer -- CPU tests only (on the GPU utx_qkv_post writes the
        send buffer directly).  Qh, Kh [H, S_loc, 128], Vt [H, 128, S_loc]."""
        P, G, Hg, S_loc = self.P, self.G, self.Hg, self.S_loc
        self.send[:, :, 0].view(G, P, Hg, S_loc, 128).copy_(Qh.reshape(P, G, Hg, S_loc, 128).transpose(0, 1))
        self.send[:, :, 1].view(G, P, Hg, S_loc, 128).copy_(Kh.reshape(P, G, Hg, S_loc, 128).transpose(0, 1))
        self.send[:, :, 2].view(G, P, Hg, 128, S_loc).copy_(Vt.reshape(P, G, Hg, 128, S_loc).transpose(0, 1))

    def set_attention_output(self, o_heads):
        """o_heads [Hp, S, 128] (rows ordered (source rank, local token)) -> self.o -- tests only (the attention kernel writes self.o[g] itself)."""
        G, Hg, S = self.G, self.Hg, self.S
        self.o.view(G, S, Hg, 128).copy_(o_heads.reshape(G, Hg, S, 128).transpose(1, 2))

    def _a2a(self, out, inp, async_op=False):
        if self.P == 1 and not self.force:
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
        """launch the Q/K/V all-to-alls of all head groups, in group order (asynchronously on NCCL: kernels issued on the current stream
        before finish_heads_in_group run beside them).  Returns the handles for finish_heads_in_group / finish_heads_in."""
        return [self._a2a(self.recv[g], self.send[g], async_op=True) for g in range(self.G)]

    def finish_heads_in_group(self, g, work=None, stream=None, unpack=None):
        """wait for group g's exchange, then (unless zero copy) ONE relayout pass: recv[g] [P][3][Hg][E] -> q [Hg, S, 128], k [Hg, S_k, 128], vt [Hg, 128, S_k] of the group
        (S_k = S unless kv_text_rows).
        Zero copy: returns None -- the attention call takes heads_blocks(g)."""
        if work is not None:
            work.wait()                                   # the current stream waits for the collective; the host does not block
        if (self.zero_copy if unpack is None else not unpack):
            return None
        P, Hg, S_loc = self.P, self.Hg, self.S_loc
        r = self.recv[g]
        q, k, vt = self.q[g * Hg:(g + 1) * Hg], self.k[g * Hg:(g + 1) * Hg], self.vt[g * Hg:(g + 1) * Hg]
        if self.on_gpu:
            lib, h = self.ctx.lib, self.ctx.handle
            st = self.ctx.stream() if stream is None else stream
            if self.kv_text_rows:
                self.ctx.check(lib.utx_sp_unpack_qkv_dedup(h, C.c_void_p(r.data_ptr()), P, Hg, S_loc, self.kv_text_rows, C.c_void_p(q.data_ptr()),
                                                           C.c_void_p(k.data_ptr()), C.c_void_p(vt.data_ptr()), st))
            else:
                self.ctx.check(lib.utx_sp_unpack_qkv(h, C.c_void_p(r.data_ptr()), P, Hg, S_loc, C.c_void_p(q.data_ptr()),
                                                     C.c_void_p(k.data_ptr()), C.c_void_p(vt.data_ptr()), st))
        else:
            q.view(Hg, P, S_loc, 128).copy_(r[:, 0].view(P, Hg, S_loc, 128).permute(1, 0, 2, 3))
            T = self.kv_text_rows
            if T:
                rk, rv = r[:, 1].view(P, Hg, S_loc, 128), r[:, 2].view(P, Hg, 128, S_loc)
                k[:, :T].copy_(rk[0, :, :T])
                k[:, T:].view(Hg, P, S_loc - T, 128).copy_(rk[:, :, T:].permute(1, 0, 2, 3))
                vt[:, :, :T].copy_(rv[0, :, :, :T])
                vt[:, :, T:].view(Hg, 128, P, S_loc - T).copy_(rv[:, :, :, T:].permute(1, 2, 0, 3))
            else:
                k.view(Hg, P, S_loc, 128).copy_(r[:, 1].view(P, Hg, S_loc, 128).permute(1, 0, 2, 3))
                vt.view(Hg, 128, P, S_loc).copy_(r[:, 2].view(P, Hg, 128, S_loc).permute(1, 2, 0, 3))
        return q, k, vt

    def finish_heads_in(self, works=None, stream=None):
        """all groups (blocking form): self.q [Hp, S, 128], self.k [Hp, S_k, 128], self.vt [Hp, 128, S_k]."""
        for g in range(self.G):
            self.finish_heads_in_group(g, None if works is None else works[g], stream, unpack=True)
        return self.q, self.k, self.vt

    def heads_in(self, Qh=None, Kh=None, Vt=None):
        """blocking form.  With arguments (CPU tests): pack them first; without: the send buffer has been written in place."""
        if Qh is not None:
            self.pack(Qh, Kh, Vt)
        return self.finish_heads_in(self.start_heads_in())

    # ---- exchange 2
    def start_tokens_out_group(self, g):
        """launch the return all-to-all of head group g (its attention output self.o[g] has been enqueued on the current stream)."""
        return self._a2a(self.o_recv[g], self.o[g].view(self.P, self.S_loc, self.Hg * 128), async_op=True)

    def finish_tokens_out_group(self, g, work, out, stream=None):
        """o_recv[g] [P][S_loc][Hg*128] -> columns (src * Hp + g * Hg) * 128 .. of `out` [S_loc, >= H*128] (rows may be strided)."""
        if work is not None:
            work.wait()
        P, Hp, Hg, S_loc = self.P, self.Hp, self.Hg, self.S_loc
        r = self.o_recv[g]
        if self.on_gpu:
            lib, h = self.ctx.lib, self.ctx.handle
            st = self.ctx.stream() if stream is None else stream
            dst = out[:, g * Hg * 128:]
            self.ctx.check(lib.utx_sp_unpack_o_cols(h, C.c_void_p(r.data_ptr()), P, Hg, S_loc, C.c_void_p(dst.data_ptr()), out.stride(0),
                                                    Hp * 128, st))
        else:
            out[:, : P * Hp * 128].unflatten(1, (P, Hp * 128))[:, :, g * Hg * 128:(g + 1) * Hg * 128].copy_(r.permute(1, 0, 2))
        return out

    def tokens_out(self, out, stream=None):
        """blocking form over all groups: self.o -> out [S_loc, >= H*128]."""
        works = [self.start_tokens_out_group(g) for g in range(self.G)]
        for g in range(self.G):
            self.finish_tokens_out_group(g, works[g], out, stream)
        return out


def local_slice(n_total, rank, world):
    """contiguous 1/world slice of n_total tokens owned by `rank`."""
    if n_total % world:
        raise ValueError("%d tokens do not split evenly over %d ranks" % (n_total, world))
    per = n_total // world
    return rank * per, (rank + 1) * per
