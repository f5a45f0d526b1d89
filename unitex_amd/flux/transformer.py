"""FluxDiT -- host-side driver of the FLUX.1-dev transformer forward on MI355X.

Mirrors the call signature of diffusers' FluxTransformer2DModel.forward as used by the reference
(/root/reference/flux_piplines/texturing/pipeline.py:646-656) for batch size 1.  All compute is HIP:
the forward is a pre-built *plan* -- a flat list of (C-ABI entry point, descriptor) pairs over
pre-allocated HBM workspaces -- replayed once per denoise step (and capturable into a hipGraph, since
every launch is ordered on torch's current stream and nothing synchronises).

Weight packing (done once at load):
  double block : Wqkv_x = [to_q; to_k; to_v], Wqkv_c = [add_q; add_k; add_v]      -> one GEMM each
  single block : Wqkvm  = [to_q; to_k; to_v; proj_mlp]                            -> one GEMM, split epilogue
  every AdaLN modulation Linear of every block is concatenated into ONE [N_mod, D] matrix: all
  shift/scale/gate vectors of a step come from a single weight-streaming GEMV.
LoRA (peft semantics, reference pipeline.py:108-112,245,263): adapters stay un-merged; the active ones
are concatenated along the rank axis (padded to 64) and consumed as an extra K-segment of the base GEMM.
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from .._lib import GemvDesc, LnModDesc, QkvPostDesc, ptr
from . import ops

BF16 = torch.bfloat16

LORA_TARGETS_DOUBLE = ["attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0", "attn.add_q_proj",
                       "attn.add_k_proj", "attn.add_v_proj", "attn.to_add_out", "ff.net.0.proj", "ff.net.2",
                       "ff_context.net.0.proj", "ff_context.net.2"]


class FluxShape:
    """FLUX.1-dev transformer hyper-parameters [3p config]; head_dim must be 128."""

    def __init__(self, num_heads=24, num_double=19, num_single=38, in_channels=64, joint_dim=4096,
                 pooled_dim=768, axes_dim=(16, 56, 56), theta=10000.0, mlp_ratio=4, guidance_embeds=True):
        self.num_heads, self.head_dim = num_heads, 128
        self.num_double, self.num_single = num_double, num_single
        self.in_channels, self.joint_dim, self.pooled_dim = in_channels, joint_dim, pooled_dim
        self.axes_dim, self.theta, self.mlp_ratio = tuple(axes_dim), theta, mlp_ratio
        self.guidance_embeds = guidance_embeds
        self.dim = num_heads * 128
        assert sum(self.axes_dim) == 128


def _pad64(n):
    return (n + 63) // 64 * 64


def rope_tables(ids: torch.Tensor, axes_dim, theta):
    """FluxPosEmbed [3p]: float64 angles -> float32 cos/sin, one column per rotation pair: [S, 64]."""
    cos, sin = [], []
    pos = ids.detach().to("cpu", torch.float64)
    for a, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = pos[:, a][:, None] * freqs[None, :]
        cos.append(torch.cos(ang).to(torch.float32))
        sin.append(torch.sin(ang).to(torch.float32))
    return torch.cat(cos, dim=-1).contiguous(), torch.cat(sin, dim=-1).contiguous()


def _timestep_proj(value_bf16_scaled: float):
    """sinusoidal embedding of a scalar (diffusers Timesteps(256, flip_sin_to_cos=True, shift 0) [3p])."""
    half = 128
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = torch.tensor([value_bf16_scaled], dtype=torch.float32)[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1).to(BF16)


def _bf16_scalar(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(BF16).to(torch.float32))


class FluxDiT:
    # the big linears that take OCP MX fp8 operands with fp8_weights=True (BASELINE configs[4]): image-side QKV and MLP of the double
    # blocks, fused QKV|MLP projection and output projection of the single blocks -- 93 % of the linear FLOPs at S = 50 688
    FP8_LINEARS_DOUBLE = ("qkv_x", "ff1_x", "ff2_x")
    FP8_LINEARS_SINGLE = ("qkvm", "out")

    def __init__(self, state_dict: Dict[str, torch.Tensor], shape: Optional[FluxShape] = None, device="cuda:0",
                 sequence_parallel=False, sp_group=None, fp8_weights=False, fp8_attention=False):
        """sequence_parallel: head-parallel ("Ulysses") sharding of ONE job over the ranks of `sp_group` (ulysses.py):
        set_positions / set_conditioning still take the FULL id / embedding tensors, forward() takes and returns this
        rank's slice of the image tokens (local_image_range)."""
        self.shape = shape or FluxShape()
        self.sp = None
        if sequence_parallel:
            import torch.distributed as dist
            self.sp = (dist.get_rank(sp_group), dist.get_world_size(sp_group), sp_group) if dist.is_initialized() else (0, 1, None)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FluxDiT runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self.ctx = ops.get_ctx(self.device.index or 0)
        self.lib = self.ctx.lib
        self._plans = {}
        self._graphs = {}
        # q / k post-processing (RMSNorm + RoPE + head-major store) fused into the QKV projection's epilogue where the GEMM runs on the
        # one-wave-per-SIMD kernel (utx_gemm_desc.qk_cols): UTX_FUSE_QK=1.  Bit-identical to GEMM -> utx_qkv_post, but OFF by default: that
        # kernel's epilogue is exposed (nothing overlaps it with one wave per SIMD) and the per-piece RMSNorm / RoPE arithmetic costs it more
        # than the saved pass over qkv -- 2045.2 -> 2052.1 ms per step at S = 50 688, equal at S = 13 824 (profiles/r02_bench_fuse_qk_ab.log)
        self.fuse_qk = os.environ.get("UTX_FUSE_QK", "0") == "1"
        self.out_rows = None       # set_output_rows: image rows whose prediction the caller consumes (None = all)
        self._lora_active: List[Tuple[Dict, float]] = []
        self._lora_version = 0
        self.attn_events = None
        self.gemm_events = None    # bench.py: list that receives (start event, end event, FLOPs) of every large-M GEMM launched on the main stream
        # double blocks: the text-token half (M = 512: a fraction of one round of tiles) runs on a second HIP stream beside the image-token half, which fills CUs
        # that the image GEMMs' tail rounds leave idle (-0.5 % per step at S = 50 688, -2.7 % at 13 824).  Opt-in since round 5 (below).
        # Round 4 found this form NOT reproducible run to run as it was built until then: with the text half's MFMA GEMMs running beside the image half's
        # utx_qkv_post, ONE q or k row of one head came out wrong in 1-1.5 % of the forwards of the full-width fp8 plan and 1 of 2000 of the bf16 plan (0 of 2000
        # on one stream).  Bisected (tools/plan_determinism_matrix.py, fp8_plan_bisect2.py, two_stream_probe.py) to the PACKED fp32 instructions hipcc emits in
        # the elementwise kernels (v_pk_mul / v_pk_add_f32: the low element in lanes 48-63 of a wave): dit_elementwise.hip is now built without them
        # (csrc/build.py) and the same plans ran 11 000 forwards without a difference (profiles/r04_two_stream_probe_nopk.log, r04_plan_determinism_soak_nopk.log);
        # tests/test_determinism_stress_gpu.py keeps a 600-forward two-stream loop in the GPU suite.  DESIGN section 9.
        # ROUND 5: OFF BY DEFAULT (UTX_TXT_STREAM=1 / set_text_stream(True) opt in).  The packed-fp32 explanation above is not a mechanism: two standalone probes that run the failing
        # instruction stream itself beside the library's GEMMs show 0 differences (DESIGN 9 b), so the build flag may only have moved the timing of a race that is still there;
        # a 600-forward stress loop bounds a 1 % rate, not a 1e-4 one.  One stream costs 0.5 % of the step at S = 50 688 (2.7 % at 13 824) -- correctness before that.
        self.overlap_text = os.environ.get("UTX_TXT_STREAM", "0") == "1"      # also under sequence parallelism (the fork / join events end before "sp_start")
        self._side = torch.cuda.Stream(device=self.device) if self.overlap_text else None
        # text-token dedup (last bullet): the reference feeds 512 all-zero text embeddings with all-zero position ids
        # (flux_piplines/texturing/pipeline.py:538-543) -- 512 IDENTICAL tokens at every layer.  When set_conditioning sees
        # identical rows (and identical ids) it carries TEXT_KEEP of them and tells the attention kernel that each stands for
        # S_txt / TEXT_KEEP keys (utx_attn_fwd_bf16_kb): the same softmax, up to fp32 summation order.  UTX_TEXT_DEDUP=0 disables.
        self.fp8_weights = bool(fp8_weights)
        # OPT-IN (round 4, BASELINE configs[4] "fp8 MFMA"): QK^T and PV on the fp8 matrix pipe (utx_attn_fwd_fp8, csrc/attention_fp8.hip): after utx_qkv_post the
        # head-major Q / K / V^T are MX-quantised (three passes over them, ~1 % of the attention's time) and the MX fp8 attention kernel runs instead of the
        # bf16 one.  A different numerics contract with its own stated tolerance (tests/test_attention_fp8_gpu.py, tests/test_e2e_tolerance_gpu.py);
        # never the default, never the bf16 bench line.  Under sequence parallelism the exchange stays bf16 and a rank quantises each head group's Q / K / V^T
        # (its heads x the full sequence) behind the group's unpack.
        self.fp8_attention = bool(fp8_attention)
        self.text_dedup = os.environ.get("UTX_TEXT_DEDUP", "1") != "0"
        self.key_bias_log2, self.key_bias_period, self.text_rows = 0.0, 0, None
        self.sp_kv_dedup = False
        self._ids = None
        self._pack(state_dict)

    def set_text_stream(self, on: bool):
        """run the text half of the double blocks on a second stream (opt-in, see __init__) or on the caller's; drops the plan"""
        self.overlap_text = bool(on)
        self._side = torch.cuda.Stream(device=self.device) if self.overlap_text else None
        self._drop_plans()

    # ------------------------------------------------------------------ weights
    def _t(self, sd, name):
        if name not in sd:
            raise KeyError("missing weight '%s' (expected diffusers FluxTransformer2DModel key names)" % name)
        return sd[name].to(device=self.device, dtype=BF16)

    def _cat(self, sd, names, suffix):
        return torch.cat([self._t(sd, n + suffix) for n in names], dim=0).contiguous()

    def _pack(self, sd):
        sh, D = self.shape, self.shape.dim
        W = {}
        for nm in ("x_embedder", "context_embedder", "proj_out"):
            W[nm + ".w"], W[nm + ".b"] = self._t(sd, nm + ".weight").contiguous(), self._t(sd, nm + ".bias")
        emb = ["timestep_embedder", "text_embedder"] + (["guidance_embedder"] if sh.guidance_embeds else [])
        for nm in emb:
            for l in ("linear_1", "linear_2"):
                k = "time_text_embed.%s.%s" % (nm, l)
                W[k + ".w"], W[k + ".b"] = self._t(sd, k + ".weight").contiguous(), self._t(sd, k + ".bias")
        mod_w, mod_b, self.mod_off = [], [], {}
        off = 0

        def add_mod(key, name):
            nonlocal off
            w = self._t(sd, name + ".weight")
            mod_w.append(w)
            mod_b.append(self._t(sd, name + ".bias"))
            self.mod_off[key] = off
            off += w.shape[0]

        self.double, self.single = [], []
        for i in range(sh.num_double):
            p = "transformer_blocks.%d." % i
            add_mod(("d", i, "x"), p + "norm1.linear")
            add_mod(("d", i, "c"), p + "norm1_context.linear")
            b = {}
            b["qkv_x.w"] = self._cat(sd, [p + "attn.to_q", p + "attn.to_k", p + "attn.to_v"], ".weight")
            b["qkv_x.b"] = self._cat(sd, [p + "attn.to_q", p + "attn.to_k", p + "attn.to_v"], ".bias")
            b["qkv_c.w"] = self._cat(sd, [p + "attn.add_q_proj", p + "attn.add_k_proj", p + "attn.add_v_proj"], ".weight")
            b["qkv_c.b"] = self._cat(sd, [p + "attn.add_q_proj", p + "attn.add_k_proj", p + "attn.add_v_proj"], ".bias")
            for short, full in (("out_x", "attn.to_out.0"), ("out_c", "attn.to_add_out"), ("ff1_x", "ff.net.0.proj"),
                                ("ff2_x", "ff.net.2"), ("ff1_c", "ff_context.net.0.proj"), ("ff2_c", "ff_context.net.2")):
                b[short + ".w"] = self._t(sd, p + full + ".weight").contiguous()
                b[short + ".b"] = self._t(sd, p + full + ".bias")
            for short, full in (("nq", "norm_q"), ("nk", "norm_k"), ("naq", "norm_added_q"), ("nak", "norm_added_k")):
                b[short] = self._t(sd, p + "attn.%s.weight" % full).contiguous()
            self.double.append(b)
        for i in range(sh.num_single):
            p = "single_transformer_blocks.%d." % i
            add_mod(("s", i), p + "norm.linear")
            b = {}
            names = [p + "attn.to_q", p + "attn.to_k", p + "attn.to_v", p + "proj_mlp"]
            b["qkvm.w"] = self._cat(sd, names, ".weight")
            b["qkvm.b"] = self._cat(sd, names, ".bias")
            b["out.w"] = self._t(sd, p + "proj_out.weight").contiguous()
            b["out.b"] = self._t(sd, p + "proj_out.bias")
            b["nq"] = self._t(sd, p + "attn.norm_q.weight").contiguous()
            b["nk"] = self._t(sd, p + "attn.norm_k.weight").contiguous()
            self.single.append(b)
        add_mod(("out",), "norm_out.linear")
        W["mod.w"] = torch.cat(mod_w, dim=0).contiguous()
        W["mod.b"] = torch.cat(mod_b, dim=0).contiguous()
        self.n_mod = off
        self.W = W
        if self.fp8_weights:
            self._quantize_fp8_weights()

    def _quantize_fp8_weights(self):
        """weights of the big linears once more as MX fp8 (e4m3 + E8M0 per 32 along K), quantised at load and again by set_lora: in fp8 mode the
        switched-on adapters are MERGED into the weight before it is quantised, W' = W + sum_i s_i B_i A_i (fp32 sum, rounded to bf16, then MX
        fp8) -- a LoRA update is far below the e4m3 quantisation step of the base weight, so carrying it as a separate bf16 K-segment buys no
        accuracy, and the merged form needs neither the LoRA-down GEMM nor a second K-segment in the fp8 kernel.  (The bf16 path keeps the
        adapters un-merged, as peft does.)  Off the denoise path: twice per mesh (texture pass, delight pass), ~12 G parameters re-quantised.
        Scales in both layouts: row-major for the 128^2-tile kernel (small / ragged shapes), tile-packed for the one-wave-per-SIMD kernel."""
        from .mx8 import quantize_weight
        for blocks, names in ((self.double, self.FP8_LINEARS_DOUBLE), (self.single, self.FP8_LINEARS_SINGLE)):
            for b in blocks:
                for nm in names:
                    W = b[nm + ".w"]
                    if W.shape[1] % 128:
                        continue
                    lora = b.get("lora." + nm)
                    if lora is not None:
                        A_cat, B_cat, alpha, Rp = lora
                        nseg = A_cat.shape[0] // Rp
                        seg = B_cat.shape[0] // nseg
                        Wf = W.to(torch.float32)
                        for si in range(nseg):
                            Wf[si * seg:(si + 1) * seg] += float(alpha) * (B_cat[si * seg:(si + 1) * seg].float() @ A_cat[si * Rp:(si + 1) * Rp].float())
                        W = Wf.to(BF16)
                        del Wf
                    b[nm + ".q"], b[nm + ".s"] = quantize_weight(W, self.ctx)
                    if W.shape[0] % 256 == 0:
                        _, b[nm + ".sp"] = quantize_weight(W, self.ctx, packed=True)

    def num_params(self):
        n = sum(v.numel() for v in self.W.values())
        for b in self.double + self.single:
            n += sum(v.numel() for v in b.values())
        return n

    # ------------------------------------------------------------------ LoRA
    def set_lora(self, adapters: List[Tuple[Dict[str, Tuple[torch.Tensor, torch.Tensor]], float]]):
        """adapters: [(lora_dict, scale)], lora_dict[module] = (A [r,in], B [out,r]) with diffusers module
        names relative to the transformer.  Zero-scaled adapters are dropped (the reference keeps them
        injected with weight 0: pipeline.py:110-111,245,263 -- contributing exactly 0)."""
        self._lora_active = [(d, float(s)) for d, s in adapters if float(s) != 0.0]
        self._lora_version += 1
        self._drop_plans()
        self._pack_full_overrides()
        self._pack_lora()
        if self.fp8_weights:
            self._quantize_fp8_weights()

    # what to do with whole-module copies an adapter file carries beside its LoRA pairs (peft modules_to_save): "swap" = the switched-on adapter's
    # copy replaces the base module for the pass (what peft's ModulesToSaveWrapper does for the ACTIVE adapter when the checkpoint is loaded
    # through peft), "ignore" = keep the base module (what a loader that drops non-LoRA keys does), "error".  The reference loads adapters with
    # diffusers' load_lora_weights of an unpinned version (pipeline.py:108-109, env.sh:12) [3p]: which of the two it does cannot be pinned here,
    # so the choice is explicit, and applying a copy is logged loudly once per set_lora.
    adapter_module_copies = os.environ.get("UTX_ADAPTER_MODULE_COPIES", "swap")
    FULL_OVERRIDE_MODULES = ("x_embedder",)   # the only parameterised entry of the trainer's modules_to_save (trainer.py:297-304)

    def _pack_full_overrides(self):
        """Whole-module copies saved with an adapter (lora_io.FULL_KEY; peft modules_to_save, trainer.py:297-304): the copy of
        the adapter that is switched on replaces the base module for this pass (texture vs delight x_embedder).  peft's
        ModulesToSaveWrapper serves ONE adapter's copy [3p]; two switched-on adapters that both carry a copy are ambiguous
        and raise instead of silently picking one."""
        self.W_override = {}
        owners = {}
        for idx, (d, s) in enumerate(self._lora_active):
            full = d.get("__full__") if isinstance(d, dict) else None
            if full and self.adapter_module_copies == "ignore":
                continue
            if full and self.adapter_module_copies == "error":
                raise ValueError("adapter %d carries whole-module copies %s (adapter_module_copies='error')" % (idx, sorted(full)))
            if full:
                import warnings
                warnings.warn("FluxDiT.set_lora: adapter %d replaces base module tensors %s with its own copies for this pass (peft modules_to_save "
                              "semantics; FluxDiT.adapter_module_copies / UTX_ADAPTER_MODULE_COPIES = 'ignore' keeps the base module)" % (idx, sorted(full)),
                              RuntimeWarning, stacklevel=3)
            for k, v in (full or {}).items():
                mod, kind = k.rsplit(".", 1)
                if mod not in self.FULL_OVERRIDE_MODULES:
                    raise NotImplementedError("adapter carries a full copy of '%s'; only %s can be swapped per adapter"
                                              % (mod, list(self.FULL_OVERRIDE_MODULES)))
                if owners.setdefault(mod, idx) != idx:
                    raise ValueError("two active adapters both carry a full copy of '%s' -- which one applies is undefined" % mod)
                base = self.W[mod + (".w" if kind == "weight" else ".b")]
                if tuple(v.shape) != tuple(base.shape):
                    raise ValueError("adapter copy of %s has shape %s, base has %s" % (k, tuple(v.shape), tuple(base.shape)))
                self.W_override[mod + (".w" if kind == "weight" else ".b")] = v.to(device=self.device, dtype=BF16).contiguous()

    def _w(self, key):
        return self.W_override.get(key, self.W[key]) if getattr(self, "W_override", None) else self.W[key]

    def _pack_lora(self):
        sh, D = self.shape, self.shape.dim
        act = self._lora_active
        self.lora_rank = 0
        for b in self.double + self.single:      # stale packed adapters go first: set_lora([]) must leave none behind
            for k in [k for k in b if k.startswith("lora.")]:
                del b[k]
        if not act:
            return

        def build(mod_names):
            """Concatenate the active adapters along rank for a (possibly fused) GEMM over `mod_names`
            (its output segments).  Returns A_cat [nseg*Rp, in], B_cat [sum(out), Rp], alpha."""
            ranks = [[(d[m][0].shape[0] if m in d else 0) for d, _ in act] for m in mod_names]
            R = max(sum(r) for r in ranks)
            if R == 0:
                return None
            Rp = _pad64(R)
            single_scale = len(act) == 1
            A_rows, B_rows = [], []
            for m in mod_names:
                a_seg, b_seg = [], []
                for d, s in act:
                    if m not in d:
                        continue
                    A, B = d[m]
                    A = A.to(self.device, torch.float32)
                    if not single_scale:
                        A = A * s
                    a_seg.append(A.to(BF16))
                    b_seg.append(B.to(self.device, BF16))
                in_f = a_seg[0].shape[1] if a_seg else None
                out_f = b_seg[0].shape[0] if b_seg else None
                A_rows.append((a_seg, in_f))
                B_rows.append((b_seg, out_f))
            return A_rows, B_rows, Rp, (act[0][1] if single_scale else 1.0)

        def finish(built, in_f, out_fs):
            if built is None:
                return None
            A_rows, B_rows, Rp, alpha = built
            A_cat = torch.zeros(len(A_rows) * Rp, in_f, dtype=BF16, device=self.device)
            B_cat = torch.zeros(sum(out_fs), Rp, dtype=BF16, device=self.device)
            ro = 0
            for si, ((a_seg, _), (b_seg, _), of) in enumerate(zip(A_rows, B_rows, out_fs)):
                r0 = 0
                for A, B in zip(a_seg, b_seg):
                    r = A.shape[0]
                    A_cat[si * Rp + r0: si * Rp + r0 + r] = A
                    B_cat[ro: ro + of, r0: r0 + r] = B
                    r0 += r
                ro += of
            return A_cat, B_cat, alpha, Rp

        for i, b in enumerate(self.double):
            p = "transformer_blocks.%d." % i
            groups = {"qkv_x": [p + "attn.to_q", p + "attn.to_k", p + "attn.to_v"],
                      "qkv_c": [p + "attn.add_q_proj", p + "attn.add_k_proj", p + "attn.add_v_proj"],
                      "out_x": [p + "attn.to_out.0"], "out_c": [p + "attn.to_add_out"],
                      "ff1_x": [p + "ff.net.0.proj"], "ff2_x": [p + "ff.net.2"],
                      "ff1_c": [p + "ff_context.net.0.proj"], "ff2_c": [p + "ff_context.net.2"]}
            for short, mods in groups.items():
                w = b[short + ".w"]
                out_fs = [w.shape[0] // len(mods)] * len(mods)
                r = finish(build(mods), w.shape[1], out_fs)
                if r is not None:
                    b["lora." + short] = r
        for i, b in enumerate(self.single):
            p = "single_transformer_blocks.%d." % i
            mods = [p + "attn.to_q", p + "attn.to_k", p + "attn.to_v"]
            r = finish(build(mods), D, [D, D, D])
            if r is not None:
                b["lora.qkvm"] = r
        rs = [v[3] for b in self.double + self.single for k, v in b.items() if k.startswith("lora.")]
        self.lora_rank = max(rs) if rs else 0

    # ------------------------------------------------------------------ plan
    def _gemm(self, plan, A, B, Cout, bias=None, lora=None, lora_n_limit=None, lora_seg_n=None, T=None, mx8=None, **kw):
        """append (optional LoRA-down GEMM +) the main GEMM to the plan.  mx8 = (Wq, Ws, aq, as_): the base product runs on OCP MX fp8
        operands -- the activation is quantised by utx_quant_mx8 into (aq, as_) first; the LoRA branch keeps its bf16 operands."""
        if mx8 is not None:
            Wq, Ws, aq, as_ = mx8[:4]
            opt = mx8[4] if len(mx8) > 4 else {}
            M, K = A.shape
            c0 = int(opt.get("a_col0", 0))     # first column of the activation's fp8 image inside the scratch (a producer's fused output lies behind x_n's)
            aqv = aq[:M, c0: c0 + K]
            if hasattr(Ws, "row_blocks"):      # tile-packed scales: the scratch buffer is shared by every fp8 GEMM of the plan (>= K/128 slabs, >= M/128 row blocks)
                from .mx8 import PackedScales
                asv = PackedScales(as_[c0 // 128:], M, K)
            else:
                asv = as_[:M, : K // 32]
            nq = int(opt.get("quant_cols", K))
            if nq > 0:      # columns [0, nq) of A are quantised here; the rest (0 = all of it) was written as fp8 by its producer (ln_mod / a GELU epilogue)
                plan.append(("quant_mx8", (A[:, :nq], aqv[:, :nq], PackedScales(as_[c0 // 128:], M, nq) if hasattr(Ws, "row_blocks") else asv)))
            kw = dict(kw, a_scale=asv, b_scale=Ws)
            if opt.get("q_out") is not None:
                kw["q_out"] = opt["q_out"]
            A_main, B_main = aqv, Wq
            lora = None     # merged into the fp8 weight (_quantize_fp8_weights)
        else:
            A_main, B_main = A, B
        if lora is not None:
            A_cat, B_cat, alpha, Rp = lora
            Tv = T[: A.shape[0], : A_cat.shape[0]]
            d0 = ops.make_gemm_desc(A, A_cat, Tv, alpha=alpha)
            plan.append((self.lib.utx_gemm_bf16, d0))
            d = ops.make_gemm_desc(A_main, B_main, Cout, bias=bias, A2=Tv, B2=B_cat, lora_n_limit=lora_n_limit,
                                   lora_seg_n=lora_seg_n, **kw)
        else:
            d = ops.make_gemm_desc(A_main, B_main, Cout, bias=bias, **kw)
        plan.append((self.lib.utx_gemm_bf16, d))

    def _par(self, plan, main_ops, side_ops):
        """two independent op lists: side by side on two streams (fork / join events) or one after the other."""
        if self.overlap_text and side_ops:
            plan.append(("par", (main_ops, side_ops, torch.cuda.Event(), torch.cuda.Event())))
        else:
            plan.extend(main_ops)
            plan.extend(side_ops)

    def _lnmod(self, plan, x, y, shift, scale, mxq=None):
        """mxq = (aq uint8 scratch rows, packed scale buffer): the result leaves as MX fp8 into aq[:n_tok, :D] + tile-packed scales instead of bf16 into y
        (utx_ln_mod_desc.q: the activation operand of the fp8 GEMM behind it, without the bf16 round trip and the quantiser's pass)."""
        d = LnModDesc()
        d.x, d.ldx, d.shift, d.scale = ptr(x), x.stride(0), ptr(shift), ptr(scale)
        d.y, d.ldy, d.n_tok, d.D, d.eps = ptr(y), y.stride(0), x.shape[0], x.shape[1], 1e-6
        if mxq is not None:
            aq, asp = mxq
            d.q, d.ldq, d.qs, d.qs_row_blocks = ptr(aq), aq.stride(0), ptr(asp), asp.stride(0) // 512
        plan.append((self.lib.utx_ln_mod, d))

    def _qkvpost(self, plan, qkv, wq, wk, ws, n_tok, tok_off, skip_qk=False):
        sh, D = self.shape, self.shape.dim
        d = QkvPostDesc()
        d.qkv, d.ld, d.q_col, d.k_col, d.v_col = ptr(qkv), qkv.stride(0), 0, D, 2 * D
        d.wq, d.wk, d.cosb, d.sinb = ptr(wq), ptr(wk), ptr(ws["cos"]), ptr(ws["sin"])
        if self.sp is not None:
            # sequence parallel: write Q, K, V^T straight into the all-to-all send buffer [G][P][3][Hg][S_loc*128] (no pack pass)
            ex = self.ex
            d.Qh, d.Kh, d.Vt = ptr(ex.send_base(0)), ptr(ex.send_base(1)), ptr(ex.send_base(2))
            d.hs_qk, d.hs_v, d.S_pad = ex.E, ex.E, ex.S_loc
            d.heads_per_group, d.gs_qk, d.gs_v = ex.Hp, ex.dest_stride, ex.dest_stride       # level 1: destination rank
            d.sub_heads, d.gs2_qk, d.gs2_v = ex.Hg, ex.group_stride, ex.group_stride        # level 2: head group (pipelined exchanges)
        else:
            d.Qh, d.Kh, d.Vt = ptr(ws["Qh"]), ptr(ws["Kh"]), ptr(ws["Vt"])
            d.hs_qk, d.hs_v, d.S_pad = ws["Qh"].stride(0), ws["Vt"].stride(0), ws["Vt"].shape[2]
        d.n_tok, d.tok_off, d.H, d.eps = n_tok, tok_off, sh.num_heads, 1e-6
        d.q_scale = (1.0 / math.sqrt(128.0)) * 1.4426950408889634   # scores become base-2 exponents (attention scale=0)
        d.skip_qk = int(bool(skip_qk))
        plan.append((self.lib.utx_qkv_post, d))

    def _attn(self, plan, ws, out, S, q_rows=None):
        if self.sp is not None:
            # exchange 1 was started by an earlier "sp_start" entry; here, head group by head group: wait + unpack, attention over the group's
            # heads x the full sequence, start its return exchange -- the fabric works beside the next group's attention (ulysses.py)
            plan.append(("sp_attn", out[:, : self.shape.dim]))
            return
        sh = self.shape
        Qh, Kh, Vt = ws["Qh"], ws["Kh"], ws["Vt"]
        r0, r1 = (0, S) if q_rows is None else q_rows     # q_rows: queries = token rows [r0, r1) only (last-block pruning); `out` starts at row r0 as well
        if self.fp8_attention:
            plan.append(("quant_qk", (Qh, ws["Q8"], ws["Q8s"])))
            plan.append(("quant_qk", (Kh, ws["K8"], ws["K8s"])))
            plan.append(("quant_vt", (Vt, ws["V8"], ws["V8s"])))
            wk8 = self._attn_work(ws, sh.num_heads, r1 - r0, S)      # scratch of the key-split tail round (round 6: the fp8 kernel splits its last round like the bf16 one)
            plan.append(("attn8", (ws["Q8"][:, r0:], ws["Q8s"][:, r0:], ws["K8"], ws["K8s"], ws["V8"], ws["V8s"], out, r1 - r0, S, Qh.shape[1], wk8)))
            return
        Qs = Qh[:, r0:]
        wk = self._attn_work(ws, sh.num_heads, r1 - r0, S)
        args = (ptr(Qs), ptr(Kh), ptr(Vt), ptr(out), Qh.stride(0), Qh.stride(1), Kh.stride(0), Kh.stride(1),
                Vt.stride(0), Vt.stride(1), out.stride(0), sh.num_heads, r1 - r0, S, 0.0, float(self.key_bias_log2), int(self.key_bias_period),
                ptr(wk), 0 if wk is None else wk.numel())
        plan.append((self.lib.utx_attn_fwd_bf16_ws, args))

    def _attn_work(self, ws, H, S_q, S_kv):
        """scratch of the attention tail split (utx_attn_fwd_bf16_ws: caller-owned, nothing allocated on the launch path, capture-safe): one buffer per
        plan, sized for its largest attention launch; the launches of a plan are ordered on one stream."""
        need = int(self.lib.utx_attn_workspace_bytes(self.ctx.handle, int(H), int(S_q), int(S_kv)))
        if need == 0:
            return None
        cur = ws.get("attn_ws")
        if cur is None or cur.numel() < need:
            # earlier plan entries hold the old buffer's pointer AND its (smaller, sufficient for them) size: keep it alive beside the new one
            if cur is not None:
                ws.setdefault("attn_ws_old", []).append(cur)
            ws["attn_ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws["attn_ws"]

    def _gemv(self, plan, x, W, b, y, silu_in=False, silu_out=False):
        d = GemvDesc()
        d.x, d.ldx, d.W, d.ldw, d.bias = ptr(x), x.stride(0), ptr(W), W.stride(0), ptr(b)
        d.y, d.ldy, d.M, d.N, d.K = ptr(y), y.stride(0), x.shape[0], W.shape[0], W.shape[1]
        d.silu_in, d.silu_out = int(silu_in), int(silu_out)
        plan.append((self.lib.utx_gemv_bf16, d))

    def _build(self, S_txt, S_img):
        sh, D, H, dev = self.shape, self.shape.dim, self.shape.num_heads, self.device
        S = S_txt + S_img
        S_pad = _pad64(S)
        Rp = self.lora_rank if self._lora_active else 0
        z = lambda *s, dtype=BF16: torch.zeros(*s, dtype=dtype, device=dev)
        ws = {
            "lat": z(S_img, sh.in_channels), "enc": z(S_txt, sh.joint_dim), "pooled": z(1, sh.pooled_dim),
            "tproj": z(1, 256), "gproj": z(1, 256),
            "e1": z(1, D), "e_t": z(1, D), "e_g": z(1, D), "e_p": z(1, D), "temb": z(1, D),
            "mod": z(1, self.n_mod),
            "h": z(S, D), "xn": z(S, D), "qkv": z(S, 3 * D), "cat": z(S, (1 + sh.mlp_ratio) * D),
            "attn": z(S, D),
            "cos": z(S, 64, dtype=torch.float32), "sin": z(S, 64, dtype=torch.float32),
            "out": z(S_img, sh.in_channels),
        }
        if self.sp is None:
            ws.update({"Qh": z(H, S_pad, 128), "Kh": z(H, S_pad, 128), "Vt": z(H, 128, S_pad)})
            if self.fp8_attention:
                u8 = torch.uint8
                ws.update({"Q8": z(H, S_pad, 128, dtype=u8), "K8": z(H, S_pad, 128, dtype=u8), "V8": z(H, 128, S_pad, dtype=u8),
                           "Q8s": z(H, S_pad, 4, dtype=u8), "K8s": z(H, S_pad, 4, dtype=u8), "V8s": z(H, S_pad // 32, 32, 4, dtype=u8)})
        if self.fp8_weights:
            from .mx8 import packed_scale_buffer
            Kmax = (1 + sh.mlp_ratio) * D
            ws["aq"] = z(S, Kmax, dtype=torch.uint8)
            ws["as"] = z(S, Kmax // 32, dtype=torch.uint8)
            ws["asp"] = packed_scale_buffer(S, Kmax, dev)
        if Rp:
            ws["T"] = z(S, 3 * Rp)
            ws["Tc"] = z(S_txt, 3 * Rp)     # LoRA-down temp of the text half (it runs concurrently with the image half)
        if self.sp is not None:
            from .ulysses import UlyssesExchange
            if S_pad != S:
                raise ValueError("sequence parallel: the local token count %d must be a multiple of 64" % S)
            self.ex = UlyssesExchange(H, S, group=self.sp[2], device=dev, dtype=BF16, ctx=self.ctx,
                                      n_cus=torch.cuda.get_device_properties(dev).multi_processor_count,
                                      kv_text_rows=(S_txt if getattr(self, "sp_kv_dedup", False) else 0))
            if self.fp8_attention:      # MX operands of ONE head group (the groups' attentions are ordered on the stream): Hg heads x the full sequence
                ex, u8 = self.ex, torch.uint8
                if ex.zero_copy:
                    raise ValueError("fp8 attention quantises the unpacked head-major operands: not with the zero-copy exchange (UTX_SP_ZERO_COPY)")
                ws.update({"Q8": z(ex.Hg, ex.S, 128, dtype=u8), "K8": z(ex.Hg, ex.S, 128, dtype=u8), "V8": z(ex.Hg, 128, ex.S, dtype=u8),
                           "Q8s": z(ex.Hg, ex.S, 4, dtype=u8), "K8s": z(ex.Hg, ex.S, 4, dtype=u8), "V8s": z(ex.Hg, ex.S // 32, 32, 4, dtype=u8)})
        T = ws.get("T")
        Tc = ws.get("Tc") if self.overlap_text else T

        def mx(b, nm, row0=0, M=None, opt=(), **shape_kw):
            """MX fp8 operands of linear `nm` of block b (None = bf16 path); the activation scratch rows start at row0.  Shapes that fill the
            chip with 256 x 256 tiles take the one-wave-per-SIMD kernel (tile-packed scales), the rest the 128 x 128-tile kernel (row-major)."""
            if not self.fp8_weights or (nm + ".q") not in b:
                return None
            if (nm + ".sp") in b and ops.mx8_uses_packed(S if M is None else M, b[nm + ".q"].shape[0], **shape_kw):
                return (b[nm + ".q"], b[nm + ".sp"], ws["aq"][row0:], ws["asp"], dict(opt))
            return (b[nm + ".q"], b[nm + ".s"], ws["aq"][row0:], ws["as"][row0:], {})

        def packed(m_):
            return m_ is not None and hasattr(m_[1], "row_blocks")
        # fp8 mode, tile-packed operands: the activation of an fp8 GEMM is written AS fp8 by its producer -- LayerNorm-modulation (utx_ln_mod_desc.q) and
        # the GELU epilogue of the GEMM in front (utx_gemm_desc.q_out) -- instead of bf16 + a quantiser pass.  Scratch layout per token row:
        # columns [0, D) = the LayerNorm output (or the attention output of a single block), [D, 5D) = the GELU output.  UTX_FP8_FUSE_QUANT=0: separate passes.
        fuse_quant = self.fp8_weights and os.environ.get("UTX_FP8_FUSE_QUANT", "1") != "0"
        W, mod = self.W, ws["mod"][0]

        def qk_fused(M, N, tok_off, wq, wk, **shape_kw):
            """qk_post argument of the QKV projection when its GEMM takes the kernel that has the fused epilogue, else None."""
            if not (self.fuse_qk and self.sp is None and not self.fp8_weights and ops.gemm_takes_w4(M, N, **shape_kw)):
                return None
            return dict(cols=2 * D, tok_off=tok_off, eps=1e-6, q_scale=(1.0 / math.sqrt(128.0)) * 1.4426950408889634,
                        wq=wq, wk=wk, cos=ws["cos"], sin=ws["sin"], Qh=ws["Qh"], Kh=ws["Kh"])
        h, xn, qkv, cat, attn = ws["h"], ws["xn"], ws["qkv"], ws["cat"], ws["attn"]
        h_c, h_x = h[:S_txt], h[S_txt:]
        xn_c, xn_x = xn[:S_txt], xn[S_txt:]
        ff = cat[:, : sh.mlp_ratio * D]  # double-block MLP hidden reuses the single-block cat buffer
        plan = []
        # ---- conditioning embeddings (M = 1 GEMVs) : CombinedTimestepGuidanceTextProjEmbeddings [3p]
        k = "time_text_embed.%s.%s"
        self._gemv(plan, ws["tproj"], W[k % ("timestep_embedder", "linear_1") + ".w"], W[k % ("timestep_embedder", "linear_1") + ".b"], ws["e1"], silu_out=True)
        self._gemv(plan, ws["e1"], W[k % ("timestep_embedder", "linear_2") + ".w"], W[k % ("timestep_embedder", "linear_2") + ".b"], ws["e_t"])
        if sh.guidance_embeds:
            self._gemv(plan, ws["gproj"], W[k % ("guidance_embedder", "linear_1") + ".w"], W[k % ("guidance_embedder", "linear_1") + ".b"], ws["e1"], silu_out=True)
            self._gemv(plan, ws["e1"], W[k % ("guidance_embedder", "linear_2") + ".w"], W[k % ("guidance_embedder", "linear_2") + ".b"], ws["e_g"])
        self._gemv(plan, ws["pooled"], W[k % ("text_embedder", "linear_1") + ".w"], W[k % ("text_embedder", "linear_1") + ".b"], ws["e1"], silu_out=True)
        self._gemv(plan, ws["e1"], W[k % ("text_embedder", "linear_2") + ".w"], W[k % ("text_embedder", "linear_2") + ".b"], ws["e_p"])
        plan.append(("temb_sum", None))
        # every AdaLN modulation vector of the step in one weight-streaming pass
        self._gemv(plan, ws["temb"], W["mod.w"], W["mod.b"], ws["mod"], silu_in=True)
        # ---- embedders
        self._gemm(plan, ws["lat"], self._w("x_embedder.w"), h_x, bias=self._w("x_embedder.b"))   # per-adapter copy if one is active
        self._gemm(plan, ws["enc"], W["context_embedder.w"], h_c, bias=W["context_embedder.b"])

        def chunks(key, n):
            o = self.mod_off[key]
            return [mod[o + j * D: o + (j + 1) * D] for j in range(n)]

        for i, b in enumerate(self.double):
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = chunks(("d", i, "x"), 6)
            csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = chunks(("d", i, "c"), 6)
            # image half / text half of the block are independent except at the joint attention: two op lists per segment
            px, pc = [], []
            m_qkv = mx(b, "qkv_x", S_txt, S_img)
            m_ff1, m_ff2 = mx(b, "ff1_x", S_txt, S_img), mx(b, "ff2_x", S_txt, S_img)
            f_qkv = fuse_quant and packed(m_qkv)
            f_ff = fuse_quant and packed(m_ff1) and packed(m_ff2)
            if f_qkv:
                m_qkv[4]["quant_cols"] = 0
            if f_ff:      # ln_mod -> fp8 -> ff1 -> GELU -> fp8 (columns D.. of the scratch) -> ff2
                m_ff1[4].update(quant_cols=0, q_out=(ws["aq"][S_txt:][:S_img, D:], ws["asp"], D // 128))
                m_ff2[4].update(quant_cols=0, a_col0=D)
            self._lnmod(px, h_x, xn_x, sh_a, sc_a, mxq=(ws["aq"][S_txt:], ws["asp"]) if f_qkv else None)
            self._lnmod(pc, h_c, xn_c, csh_a, csc_a)
            has_l = b.get("lora.qkv_x") is not None
            qkx = qk_fused(S_img, 3 * D, S_txt, b["nq"], b["nk"], K2=(Rp if has_l else 0), lora_seg_n=D, lora_n_limit=3 * D)
            self._gemm(px, xn_x, b["qkv_x.w"], qkv[S_txt:], bias=b["qkv_x.b"], lora=b.get("lora.qkv_x"),
                       lora_n_limit=3 * D, lora_seg_n=D, T=T, mx8=m_qkv, qk_post=qkx)
            self._gemm(pc, xn_c, b["qkv_c.w"], qkv[:S_txt], bias=b["qkv_c.b"], lora=b.get("lora.qkv_c"),
                       lora_n_limit=3 * D, lora_seg_n=D, T=Tc)
            self._qkvpost(px, qkv[S_txt:], b["nq"], b["nk"], ws, S_img, S_txt, skip_qk=qkx is not None)
            self._qkvpost(pc, qkv[:S_txt], b["naq"], b["nak"], ws, S_txt, 0)
            self._par(plan, px, pc)
            if self.sp is not None:
                plan.append(("sp_start", None))
            self._attn(plan, ws, attn, S)
            px, pc = [], []
            self._gemm(px, attn[S_txt:], b["out_x.w"], h_x, bias=b["out_x.b"], lora=b.get("lora.out_x"), T=T,
                       gate=g_a, res=h_x)
            self._gemm(pc, attn[:S_txt], b["out_c.w"], h_c, bias=b["out_c.b"], lora=b.get("lora.out_c"), T=Tc,
                       gate=cg_a, res=h_c)
            self._lnmod(px, h_x, xn_x, sh_m, sc_m, mxq=(ws["aq"][S_txt:], ws["asp"]) if f_ff else None)
            self._gemm(px, xn_x, b["ff1_x.w"], ff[S_txt:], bias=b["ff1_x.b"], lora=b.get("lora.ff1_x"), T=T, gelu_from=0,
                       mx8=m_ff1)
            self._gemm(px, ff[S_txt:], b["ff2_x.w"], h_x, bias=b["ff2_x.b"], lora=b.get("lora.ff2_x"), T=T,
                       gate=g_m, res=h_x, mx8=m_ff2)
            self._lnmod(pc, h_c, xn_c, csh_m, csc_m)
            self._gemm(pc, xn_c, b["ff1_c.w"], ff[:S_txt], bias=b["ff1_c.b"], lora=b.get("lora.ff1_c"), T=Tc, gelu_from=0)
            self._gemm(pc, ff[:S_txt], b["ff2_c.w"], h_c, bias=b["ff2_c.b"], lora=b.get("lora.ff2_c"), T=Tc,
                       gate=cg_m, res=h_c)
            self._par(plan, px, pc)
        n_out = S_img if (self.out_rows is None or self.sp is not None) else max(1, min(int(self.out_rows), S_img))
        for i, b in enumerate(self.single):
            sh_, sc_, g_ = chunks(("s", i), 3)
            pruned = i == len(self.single) - 1 and n_out < S_img
            m_qkvm = None if (pruned or self.sp is not None) else mx(b, "qkvm", n_split=3 * D, gelu_from=3 * D)
            m_out = None if pruned else mx(b, "out")
            # sequence parallel: the projection is cut at column 3D (q|k|v first, so that their exchange starts early; the MLP half runs beside the
            # all-to-all) -- in fp8 mode both halves take the MX kernel on row slices of the same quantised weight (round 4; round 3 kept them in bf16)
            sp_mx = None
            if self.sp is not None and not pruned and self.fp8_weights and ("qkvm.sp") in b and \
                    ops.mx8_uses_packed(S, 3 * D) and ops.mx8_uses_packed(S, sh.mlp_ratio * D):
                sp_mx = (b["qkvm.q"], b["qkvm.sp"])
            f_sgl = fuse_quant and (packed(m_qkvm) or sp_mx is not None) and packed(m_out)
            if f_sgl:     # ln_mod -> fp8 -> [q|k|v|mlp] projection, GELU(mlp) -> fp8 (columns D..) ; attention output quantised into columns [0, D) ; out-projection
                if m_qkvm is not None:
                    m_qkvm[4].update(quant_cols=0, q_out=(ws["aq"][:S, D:], ws["asp"], D // 128))
                m_out[4].update(quant_cols=D)
            self._lnmod(plan, h, xn, sh_, sc_, mxq=(ws["aq"], ws["asp"]) if f_sgl else None)
            if pruned:
                # LAST block, only rows [r0, r1) of its output are consumed (set_output_rows): keys / values for every token, but query,
                # MLP and output projection for those rows only.  Same weights, same K order per output element: the rows that are
                # computed equal the unpruned block's bit for bit (up to which query blocks the attention tail split picks).
                r0, r1 = S_txt, S_txt + n_out
                Wm, bm = b["qkvm.w"], b["qkvm.b"]
                if self.fp8_weights and (("qkvm.sp") in b) and ops.mx8_uses_packed(n_out, D) and ops.mx8_uses_packed(S, 2 * D):
                    # the same block on MX fp8 operands (adapters merged into the weights): x_n quantised once for the k | v projection over all rows
                    # and once more, as a matrix of its own, for the rows that keep a query (tile-packed scales are addressed from a 128-row-aligned
                    # origin; r0 = the text rows is not one) -- second activation scratch `aq2`
                    from .mx8 import PackedScales, packed_scale_buffer
                    Kmax = (1 + sh.mlp_ratio) * D
                    if "aq2" not in ws:
                        ws["aq2"] = torch.zeros(n_out, Kmax, dtype=torch.uint8, device=dev)
                        ws["asp2"] = packed_scale_buffer(n_out, Kmax, dev)
                    Wq, Wsp = b["qkvm.q"], b["qkvm.sp"]
                    a_all = PackedScales(ws["asp"], S, D)
                    plan.append(("quant_mx8", (xn, ws["aq"][:S, :D], a_all)))
                    plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(ws["aq"][:S, :D], Wq[D: 3 * D], qkv[:, D: 3 * D], bias=bm[D: 3 * D],
                                                                            a_scale=a_all, b_scale=Wsp.row_slice(D, 3 * D))))
                    a_q = PackedScales(ws["asp2"], n_out, D)
                    plan.append(("quant_mx8", (xn[r0:r1], ws["aq2"][:, :D], a_q)))
                    plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(ws["aq2"][:, :D], Wq[:D], qkv[r0:r1, :D], bias=bm[:D],
                                                                            a_scale=a_q, b_scale=Wsp.row_slice(0, D))))
                    plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(ws["aq2"][:, :D], Wq[3 * D:], cat[r0:r1, D:], bias=bm[3 * D:], gelu_from=0,
                                                                            a_scale=a_q, b_scale=Wsp.row_slice(3 * D, Wq.shape[0]))))
                    self._qkvpost(plan, qkv, b["nq"], b["nk"], ws, S, 0)
                    self._attn(plan, ws, cat[r0:], S, q_rows=(r0, r1))
                    a_o = PackedScales(ws["asp2"], n_out, Kmax)
                    plan.append(("quant_mx8", (cat[r0:r1], ws["aq2"], a_o)))
                    plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(ws["aq2"], b["out.q"], h[r0:r1], bias=b["out.b"], gate=g_, res=h[r0:r1],
                                                                            a_scale=a_o, b_scale=b["out.sp"])))
                    continue
                lora = b.get("lora.qkvm")
                kw_kv, kw_q = {}, {}
                if lora is not None:
                    A_cat, B_cat, alpha, _rp = lora
                    R = A_cat.shape[0] // 3
                    Tv = T[:S, : 3 * R]
                    plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(xn, A_cat, Tv, alpha=alpha)))
                    kw_kv = dict(A2=Tv[:, R:], B2=B_cat[D: 3 * D], lora_n_limit=2 * D, lora_seg_n=D)
                    kw_q = dict(A2=Tv[r0:r1, :R], B2=B_cat[:D], lora_n_limit=D, lora_seg_n=D)
                plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(xn, Wm[D: 3 * D], qkv[:, D: 3 * D], bias=bm[D: 3 * D], **kw_kv)))
                plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(xn[r0:r1], Wm[:D], qkv[r0:r1, :D], bias=bm[:D], **kw_q)))
                plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(xn[r0:r1], Wm[3 * D:], cat[r0:r1, D:], bias=bm[3 * D:], gelu_from=0)))
                self._qkvpost(plan, qkv, b["nq"], b["nk"], ws, S, 0)      # Q rows outside [r0, r1) are stale (finite) and never read
                self._attn(plan, ws, cat[r0:], S, q_rows=(r0, r1))
                plan.append((self.lib.utx_gemm_bf16, ops.make_gemm_desc(cat[r0:r1], b["out.w"], h[r0:r1], bias=b["out.b"], gate=g_, res=h[r0:r1])))
                continue
            if self.sp is None:
                # one GEMM for [q|k|v|proj_mlp]: qkv -> qkv buffer, GELU(mlp) -> cat[:, D:]
                has_l = b.get("lora.qkvm") is not None
                qkm = qk_fused(S, (3 + sh.mlp_ratio) * D, 0, b["nq"], b["nk"], n_split=3 * D, gelu_from=3 * D, K2=(Rp if has_l else 0),
                               lora_seg_n=D, lora_n_limit=3 * D)
                self._gemm(plan, xn, b["qkvm.w"], qkv, bias=b["qkvm.b"], lora=b.get("lora.qkvm"), lora_n_limit=3 * D,
                           lora_seg_n=D, T=T, gelu_from=3 * D, n_split=3 * D, C1=cat[:, D:], mx8=m_qkvm, qk_post=qkm)
                self._qkvpost(plan, qkv, b["nq"], b["nk"], ws, S, 0, skip_qk=qkm is not None)
            else:
                # sequence parallel: the same GEMM cut at column 3D (identical arithmetic per column) so that the Q/K/V exchange
                # starts as soon as q|k|v exist and the MLP half of the projection runs beside the all-to-all
                if sp_mx is not None:
                    Wq, Wsp = sp_mx
                    NW = Wq.shape[0]
                    m_a = (Wq[: 3 * D], Wsp.row_slice(0, 3 * D), ws["aq"], ws["asp"], {"quant_cols": 0 if f_sgl else D})
                    m_b = (Wq[3 * D:], Wsp.row_slice(3 * D, NW), ws["aq"], ws["asp"],
                           dict({"quant_cols": 0}, **({"q_out": (ws["aq"][:S, D:], ws["asp"], D // 128)} if f_sgl else {})))
                else:
                    m_a = m_b = None
                self._gemm(plan, xn, b["qkvm.w"][: 3 * D], qkv, bias=b["qkvm.b"][: 3 * D], lora=b.get("lora.qkvm"),
                           lora_n_limit=3 * D, lora_seg_n=D, T=T, mx8=m_a)
                self._qkvpost(plan, qkv, b["nq"], b["nk"], ws, S, 0)
                plan.append(("sp_start", None))
                self._gemm(plan, xn, b["qkvm.w"][3 * D:], cat[:, D:], bias=b["qkvm.b"][3 * D:], gelu_from=0, mx8=m_b)
            self._attn(plan, ws, cat, S)  # attention output lands in cat[:, :D] (row stride 5D)
            self._gemm(plan, cat, b["out.w"], h, bias=b["out.b"], gate=g_, res=h, mx8=m_out)
        o = self.mod_off[("out",)]
        scale, shift = mod[o: o + D], mod[o + D: o + 2 * D]  # AdaLayerNormContinuous: (scale, shift) [3p]
        self._lnmod(plan, h_x[:n_out], xn_x[:n_out], shift, scale)
        self._gemm(plan, xn_x[:n_out], W["proj_out.w"], ws["out"][:n_out], bias=W["proj_out.b"])
        self._assign_streamk(plan, ws)
        return {"ws": ws, "plan": plan, "S_txt": S_txt, "S_img": S_img, "n_out": n_out}

    def gemm_census(self, n_cus=None):
        """{kernel name: launches per forward} of the current plan, as the library's own dispatch (utx_gemm_plan) sees each descriptor, plus
        "w4_split_tail": the launches whose partly filled last round is cut along K (what bench.py reports in `config`)."""
        if not self._plans:
            return {}
        if n_cus is None:
            n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        out = {}
        arr = (C.c_int * 4)()

        def walk(entries):
            for e in entries:
                if e[0] == "par":
                    walk(e[1][0]); walk(e[1][1])
                elif e[0] is self.lib.utx_gemm_bf16:
                    if self.lib.utx_gemm_plan(C.byref(e[1]), int(n_cus), C.byref(arr)) != 0:
                        continue
                    k = ops.GEMM_KERNELS[arr[0]]
                    out[k] = out.get(k, 0) + 1
                    if arr[2] > 0:
                        out["w4_split_tail"] = out.get("w4_split_tail", 0) + 1
        walk(next(iter(self._plans.values()))["plan"])
        return out

    def _assign_streamk(self, plan, ws):
        """scratch of the large-M GEMM's balanced tail round (utx_gemm_desc.sk_work): one buffer for the GEMMs of the main stream -- they
        are ordered among themselves; the text-side ops of a "par" entry run beside them on the second stream and get none (their GEMMs are
        far below the size where the tail is split).  UTX_GEMM_STREAMK=0 at library level switches the split off."""
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count

        def main_gemms(entries):
            for e in entries:
                if e[0] is self.lib.utx_gemm_bf16:
                    yield e[1]
                elif e[0] == "par":
                    yield from main_gemms(e[1][0])
        descs = list(main_gemms(plan))
        # only launches with more 256 x 256 tiles than CUs can have a partly filled LAST round (small models never pay for the buffer)
        if not any(((d.M + 255) // 256) * (d.N // 256) > ncu for d in descs):
            return
        if "sk" not in ws:
            ws["sk"] = ops.streamk_workspace(self.device, shared=False)
        sk = ws["sk"]
        for d in descs:
            d.sk_work, d.sk_work_bytes = ptr(sk), sk.numel()

    # ------------------------------------------------------------------ forward
    TEXT_KEEP = 64      # rows of the (identical) text tokens that are carried when the dedup applies: one 64-key attention tile

    def local_text_range(self, S_txt):
        from .ulysses import local_slice
        return (0, S_txt) if self.sp is None else local_slice(S_txt, self.sp[0], self.sp[1])

    def local_image_range(self, S_img):
        from .ulysses import local_slice
        return (0, S_img) if self.sp is None else local_slice(S_img, self.sp[0], self.sp[1])

    def __del__(self):
        try:
            self._drop_plans()      # the C-side plans own a HIP stream and events
        except Exception:  # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass

    def _drop_plans(self):
        for p in self._plans.values():
            if p.get("cplan") is not None:
                self.lib.utx_plan_free(p["cplan"])
                p["cplan"] = None
        self._plans.clear()
        self._graphs = {}

    def _tproj_device(self, t1000):
        """the 256-channel sinusoidal projection of a timestep, resident on the device: computed once per distinct value with torch's fp32 host
        exp / cos / sin (so that it equals the oracle's bit for bit) and cached -- a schedule has 28 distinct timesteps and runs twice per mesh (texture
        pass, delight pass); afterwards a step's only host-to-device traffic is gone (device-to-device copy of 512 bytes)."""
        c = self.__dict__.setdefault("_tproj_cache", {})
        t = c.get(t1000)
        if t is None:
            if len(c) > 4096:
                c.clear()
            t = c[t1000] = _timestep_proj(t1000).to(self.device)
        return t

    def set_output_rows(self, n):
        """Only the first n image tokens' prediction will be read from forward()'s result (None = all).  Call it BETWEEN set_positions and
        set_conditioning: set_positions starts a new job and resets it to None, so a later user of the same FluxDiT never inherits a previous
        caller's pruning (rows >= n of the result are undefined).
        The texturing pipeline discards the prediction of the condition tokens: the condition tail of the latents is re-pinned before
        every transformer call and cut off at the end (flux_piplines/texturing/pipeline.py:645,660,684), so in the LAST block -- whose
        output no later block attends to -- only the noise tokens need a query, an MLP row and an output projection, and the final
        norm / proj_out only those rows.  Keys and values of every token are still computed.  Rows >= n of the result are undefined.
        Exact for the rows that are read; not applied under sequence parallelism or with fp8 weights (the full block runs)."""
        self.out_rows = None if n is None else int(n)

    def set_positions(self, txt_ids, img_ids):
        """Position ids of the joint sequence cat(txt_ids, img_ids) (FULL tensors, also under sequence parallelism).  The plan
        and the rotary tables are built by set_conditioning, which knows whether the text tokens can be deduplicated."""
        self._ids = (txt_ids.detach().to("cpu", torch.float32).contiguous(), img_ids.detach().to("cpu", torch.float32).contiguous())
        self.out_rows = None          # sticky pruning would hand the next caller stale condition-token rows (set_output_rows)

    def set_conditioning(self, encoder_hidden_states, pooled_projections, guidance: float):
        if self._ids is None:
            raise RuntimeError("set_positions must be called before set_conditioning")
        txt_ids, img_ids = self._ids
        S_txt = encoder_hidden_states.shape[-2]
        assert txt_ids.shape[0] == S_txt, "txt_ids and encoder_hidden_states disagree on the number of text tokens"
        enc = encoder_hidden_states.reshape(S_txt, -1)
        world = 1 if self.sp is None else self.sp[1]
        K = self.TEXT_KEEP
        identical = bool(self.text_dedup and S_txt > K * world and S_txt % (K * world) == 0 and
                         torch.equal(txt_ids, txt_ids[:1].expand_as(txt_ids)) and torch.equal(enc, enc[:1].expand_as(enc)))
        # sequence parallel, identical text rows: every rank carries its own K copies.  Default (round 6): the receive-side unpack keeps ONE copy of them in K / V^T
        # (ulysses.py kv_text_rows) -- a rank's attention launch is then the single-GPU launch (S_txt / K-fold keys in tile 0 only: the 4 x 64 kernel) over its heads.
        # The zero-copy exchange and the fp8 attention read every rank's copy: a text tile at the start of every rank's block of the gathered key sequence
        # (keys ordered (source rank, local token)), each key counting S_txt / (K world)-fold (key_bias_period).  UTX_SP_KV_DEDUP=0 forces that form (A/B, tests).
        self.sp_kv_dedup = bool(identical and world > 1 and not self.fp8_attention and os.environ.get("UTX_SP_ZERO_COPY", "0") != "1" and
                                os.environ.get("UTX_SP_KV_DEDUP", "1") != "0")
        if identical:
            t0, t1 = 0, K
            self.key_bias_log2 = math.log2(S_txt / float(K * (1 if self.sp_kv_dedup else world)))
            self.text_rows = K
        else:
            t0, t1 = self.local_text_range(S_txt)
            self.key_bias_log2, self.text_rows = 0.0, None
        i0, i1 = self.local_image_range(img_ids.shape[0])
        S_loc = (t1 - t0) + (i1 - i0)
        self.key_bias_period = (S_loc // 64) if (identical and world > 1 and not self.sp_kv_dedup) else 0
        key = (t1 - t0, i1 - i0, self._lora_version, self.key_bias_log2, self.key_bias_period, self.out_rows, self.sp_kv_dedup)
        if key not in self._plans:
            self._drop_plans()   # one live plan: workspaces are large
            self._plans[key] = self._build(t1 - t0, i1 - i0)
            if os.environ.get("UTX_C_PLAN", "1") != "0":
                self.compile_plan(self._plans[key])
        p = self._plans[key]
        ws = p["ws"]
        cos, sin = rope_tables(torch.cat([txt_ids[t0:t1], img_ids[i0:i1]], dim=0), self.shape.axes_dim, self.shape.theta)
        ws["cos"].copy_(cos, non_blocking=True)
        ws["sin"].copy_(sin, non_blocking=True)
        ws["enc"].copy_(enc[t0:t1].to(BF16))
        ws["pooled"].copy_(pooled_projections.reshape(1, -1).to(BF16))
        g1000 = _bf16_scalar(_bf16_scalar(guidance) * 1000.0)  # guidance.to(dtype) * 1000 in bf16 [3p]
        ws["gproj"].copy_(_timestep_proj(g1000))

    def _launch(self, fn, d, st, timed=True):
        """one stream-ordered launch of a plan entry that is a C-ABI call: (entry point, descriptor) or the MX fp8 quantiser"""
        gev = self.gemm_events
        if gev is not None and timed and fn is self.lib.utx_gemm_bf16 and d.M >= 4096:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(self.ctx.handle, C.byref(d), st)
            b.record()
            k2 = d.K2 * min(d.lora_n_limit, d.N) / d.N if (d.K2 > 0 and d.lora_n_limit > 0) else 0.0     # the LoRA K-segment runs on the columns < lora_n_limit
            gev.append((a, b, 2.0 * d.M * d.N * (d.K + k2)))
            if rc:
                self.ctx.check(rc)
            return
        if fn == "quant_qk":
            ops.quant_qk_mx8(d[0], out=(d[1], d[2]))
            return
        if fn == "quant_vt":
            ops.quant_vt_mx8(d[0], out=(d[1], d[2]))
            return
        if fn == "attn8":
            q8, qs, k8, ks, v8, vs, out, n_q, S_kv, S_pad, wk8 = d
            ev = getattr(self, "attn_events", None)
            if ev is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            rc = self.lib.utx_attn_fwd_fp8_ws(self.ctx.handle, ptr(q8), ptr(qs), ptr(k8), ptr(ks), ptr(v8), ptr(vs), ptr(out), out.stride(0), self.shape.num_heads,
                                              int(n_q), int(S_kv), int(S_pad), float(self.key_bias_log2), int(self.key_bias_period), ptr(wk8), 0 if wk8 is None else wk8.numel(), st)
            if ev is not None:
                b.record()
                ev.append((a, b))
            if rc:
                self.ctx.check(rc)
            return
        if fn == "quant_mx8":
            x_, q_, s_ = d
            if hasattr(s_, "row_blocks"):
                rc = self.lib.utx_quant_mx8_packed(self.ctx.handle, ptr(x_), x_.stride(0), ptr(q_), q_.stride(0), ptr(s_.data), s_.row_blocks,
                                                   x_.shape[0], x_.shape[1], st)
            else:
                rc = self.lib.utx_quant_mx8(self.ctx.handle, ptr(x_), x_.stride(0), ptr(q_), q_.stride(0), ptr(s_), s_.stride(0),
                                            x_.shape[0], x_.shape[1], st)
        else:
            rc = fn(self.ctx.handle, C.byref(d), st)
        if rc:
            self.ctx.check(rc)

    def run_plan(self, p):
        h, st = self.ctx.handle, self.ctx.stream()
        lib, ws = self.lib, p["ws"]
        for fn, d in p["plan"]:
            if fn == "par":
                main_ops, side_ops, ev_fork, ev_join = d
                main = torch.cuda.current_stream(self.device)
                ev_fork.record(main)
                self._side.wait_event(ev_fork)
                st2 = C.c_void_p(self._side.cuda_stream)
                for f2, d2 in side_ops:
                    self._launch(f2, d2, st2, timed=False)
                for f2, d2 in main_ops:
                    self._launch(f2, d2, st)
                ev_join.record(self._side)
                main.wait_event(ev_join)
            elif fn in ("quant_mx8", "quant_qk", "quant_vt", "attn8"):
                self._launch(fn, d, st)
            elif fn == "temb_sum":
                # conditioning = (timesteps_emb + guidance_emb) + pooled_projections, bf16 adds [3p]
                t = ws["e_t"]
                if self.shape.guidance_embeds:
                    t = t + ws["e_g"]
                torch.add(t, ws["e_p"], out=ws["temb"])
            elif fn == "sp_start" or fn == "sp_attn":
                self._sp_entry(fn, d, ws, st)
            elif fn is lib.utx_attn_fwd_bf16_ws:
                ev = getattr(self, "attn_events", None)
                if ev is not None:  # bench.py: HIP events on the launch stream around the dominant kernel
                    a = torch.cuda.Event(enable_timing=True)
                    b = torch.cuda.Event(enable_timing=True)
                    a.record()
                    rc = fn(h, *d, st)
                    b.record()
                    ev.append((a, b))
                else:
                    rc = fn(h, *d, st)
                if rc:
                    self.ctx.check(rc)
            else:
                self._launch(fn, d, st)

    def _sp_entry(self, fn, d, ws, st):
        """the host-side entries of a sequence-parallel plan: start of the Q / K / V exchange; per head group wait + unpack, attention, start of the return
        exchange, then the waits + unpacks of the return exchanges (ulysses.py)"""
        lib, h = self.lib, self.ctx.handle
        if fn == "sp_start":
            self._sp_work = self.ex.start_heads_in()
            return
        if fn == "sp_attn":
            ex = self.ex
            works, self._sp_work = self._sp_work, None
            ev = getattr(self, "attn_events", None)
            wk = self._attn_work(ws, ex.Hg, ex.S, ex.S_k)      # S queries over S_k keys (S_k < S: the ranks' identical text rows kept once)
            back = []
            for g in range(ex.G):
                hd = ex.finish_heads_in_group(g, None if works is None else works[g])
                og = ex.o[g]
                if self.fp8_attention:      # the group's MX operands (in front of the timed bracket, as on one GPU: `attn_events` bracket the attention kernel)
                    q, k, vt = hd
                    for src, d8, s8 in ((q, ws["Q8"], ws["Q8s"]), (k, ws["K8"], ws["K8s"])):
                        self.ctx.check(lib.utx_quant_mx8(h, ptr(src), 128, ptr(d8), 128, ptr(s8), 4, ex.Hg * ex.S, 128, st))
                    self.ctx.check(lib.utx_quant_vt_mx8(h, ptr(vt), ptr(ws["V8"]), ptr(ws["V8s"]), ex.Hg, ex.S, st))
                if ev is not None:
                    a = torch.cuda.Event(enable_timing=True)
                    b = torch.cuda.Event(enable_timing=True)
                    a.record()
                if hd is None:
                    # zero copy: Q / K / V^T read from the receive buffer of the all-to-all, one block of S_loc tokens per source rank (utx_attn_fwd_bf16_blk)
                    qp, kp, vp, hs, bs, rows = ex.heads_blocks(g)
                    rc = lib.utx_attn_fwd_bf16_blk(h, C.c_void_p(qp), C.c_void_p(kp), C.c_void_p(vp), ptr(og), hs, 128, hs, 128, hs, rows, og.stride(0),
                                                   ex.Hg, ex.S, ex.S, 0.0, float(self.key_bias_log2), int(self.key_bias_period), ptr(wk),
                                                   0 if wk is None else wk.numel(), rows, bs, bs, bs, st)
                elif self.fp8_attention:
                    rc = lib.utx_attn_fwd_fp8(h, ptr(ws["Q8"]), ptr(ws["Q8s"]), ptr(ws["K8"]), ptr(ws["K8s"]), ptr(ws["V8"]), ptr(ws["V8s"]), ptr(og), og.stride(0),
                                              ex.Hg, ex.S, ex.S, ex.S, float(self.key_bias_log2), int(self.key_bias_period), st)
                else:
                    q, k, vt = hd
                    rc = lib.utx_attn_fwd_bf16_ws(h, ptr(q), ptr(k), ptr(vt), ptr(og), q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                                  vt.stride(0), vt.stride(1), og.stride(0), ex.Hg, ex.S, ex.S_k, 0.0, float(self.key_bias_log2),
                                                  int(self.key_bias_period), ptr(wk), 0 if wk is None else wk.numel(), st)
                if ev is not None:
                    b.record()
                    ev.append((a, b))
                if rc:
                    self.ctx.check(rc)
                back.append(ex.start_tokens_out_group(g))
            for g in range(ex.G):
                ex.finish_tokens_out_group(g, back[g], d)

    # ------------------------------------------------------------------ C-side replay (utx_plan)
    def compile_plan(self, p=None):
        """Copy the per-step plan into a utx_plan (include/unitex_hip.h, csrc/plan.cpp): forward() then replays it with ONE C call per step
        (utx_plan_run = SURVEY 8b's `utx_dit_step`) instead of ~700 ctypes calls -- the same launchers in the same order on the same two streams, so
        the result is bit-identical.  Under sequence parallelism the collectives stay torch.distributed calls of this host and the launches between them are
        replayed range by range (p["segments"], utx_plan_run_range).  Not used while per-kernel events are requested (bench.py's roofline timing walks the
        Python list).  Returns the handle, or None when the plan has entries C cannot replay."""
        p = next(iter(self._plans.values())) if p is None else p
        lib, ws = self.lib, p["ws"]
        h = C.c_void_p()
        self.ctx.check(lib.utx_plan_create(self.ctx.handle, C.byref(h)))
        # sequence parallel (round 4): the collectives stay torch.distributed calls of the host, the launches BETWEEN them are replayed by ranges
        # (utx_plan_run_range): p["segments"] = [(first entry, end entry, host op behind the range or None)] -- a rank then issues two C calls +
        # its 2 G collectives + 3 G exchange-side launches per layer instead of one ctypes call per kernel (~1000 per step)
        segments, seg_begin = ([] if self.sp is not None else None), 0

        def add(fn, d):
            if fn is lib.utx_gemm_bf16:
                return lib.utx_plan_add_gemm(h, C.byref(d))
            if fn is lib.utx_gemv_bf16:
                return lib.utx_plan_add_gemv(h, C.byref(d))
            if fn is lib.utx_ln_mod:
                return lib.utx_plan_add_ln_mod(h, C.byref(d))
            if fn is lib.utx_qkv_post:
                return lib.utx_plan_add_qkv_post(h, C.byref(d))
            if fn is lib.utx_attn_fwd_bf16_ws:
                return lib.utx_plan_add_attn(h, *d)
            if isinstance(fn, str) and fn == "quant_mx8":
                x_, q_, s_ = d
                if hasattr(s_, "row_blocks"):
                    return lib.utx_plan_add_quant_mx8(h, ptr(x_), x_.stride(0), ptr(q_), q_.stride(0), ptr(s_.data), s_.row_blocks, x_.shape[0], x_.shape[1], 1)
                return lib.utx_plan_add_quant_mx8(h, ptr(x_), x_.stride(0), ptr(q_), q_.stride(0), ptr(s_), s_.stride(0), x_.shape[0], x_.shape[1], 0)
            if isinstance(fn, str) and fn == "quant_qk":      # head-major Q / K rows: utx_quant_mx8 over the [H * S_pad, 128] view (ops.quant_qk_mx8)
                x_, q_, s_ = d
                return lib.utx_plan_add_quant_mx8(h, ptr(x_), 128, ptr(q_), 128, ptr(s_), 4, x_.shape[0] * x_.shape[1], 128, 0)
            if isinstance(fn, str) and fn == "quant_vt":
                x_, q_, s_ = d
                return lib.utx_plan_add_quant_vt_mx8(h, ptr(x_), ptr(q_), ptr(s_), x_.shape[0], x_.shape[2])
            if isinstance(fn, str) and fn == "attn8":
                q8, qs, k8, ks, v8, vs, out, n_q, S_kv, S_pad, wk8 = d
                return lib.utx_plan_add_attn_fp8_ws(h, ptr(q8), ptr(qs), ptr(k8), ptr(ks), ptr(v8), ptr(vs), ptr(out), out.stride(0), self.shape.num_heads,
                                                    int(n_q), int(S_kv), int(S_pad), float(self.key_bias_log2), int(self.key_bias_period), ptr(wk8), 0 if wk8 is None else wk8.numel())
            if isinstance(fn, str) and fn == "temb_sum":
                g = ws["e_g"] if self.shape.guidance_embeds else None
                return lib.utx_plan_add_add3(h, ptr(ws["e_t"]), ptr(g), ptr(ws["e_p"]), ptr(ws["temb"]), ws["temb"].numel())
            return -100
        ok = True
        for fn, d in p["plan"]:
            if isinstance(fn, str) and fn == "par":
                main_ops, side_ops = d[0], d[1]
                rcs = [lib.utx_plan_fork(h)] + [add(f2, d2) for f2, d2 in side_ops] + [lib.utx_plan_main(h)] + \
                      [add(f2, d2) for f2, d2 in main_ops] + [lib.utx_plan_join(h)]
            elif segments is not None and isinstance(fn, str) and fn in ("sp_start", "sp_attn"):
                n = int(lib.utx_plan_size(h))
                segments.append((seg_begin, n, (fn, d)))
                seg_begin = n
                rcs = [0]
            else:
                rcs = [add(fn, d)]
            if any(rc != 0 for rc in rcs):
                ok = False
                break
        if not ok:
            lib.utx_plan_free(h)
            return None
        if segments is not None:
            segments.append((seg_begin, int(lib.utx_plan_size(h)), None))
        old = p.get("cplan")
        if old is not None:
            lib.utx_plan_free(old)
        p["cplan"] = h
        p["segments"] = segments
        return h

    def _run_segments(self, p):
        """sequence parallel: the step as C-replayed launch ranges with the host's exchange operations between them"""
        st, ws = self.ctx.stream(), p["ws"]
        bad = C.c_int(-1)
        for b0, b1, host_op in p["segments"]:
            if b1 > b0:
                rc = self.lib.utx_plan_run_range(p["cplan"], b0, b1, st, C.byref(bad))
                if rc:
                    raise RuntimeError("utx_plan_run_range [%d, %d): entry %d failed with code %d" % (b0, b1, bad.value, rc))
            if host_op is not None:
                self._sp_entry(host_op[0], host_op[1], ws, st)

    def build_c_dit_plan(self, p=None):
        """The same step built by the C-side builder (utx_dit_load, csrc/dit_plan.cpp): this object's packed weights and the plan's workspaces are
        handed over as plain pointer structs and the library assembles the launch list itself -- what a non-Python host would do.  Single-GPU paths, bf16 and
        MX fp8 (returns None under sequence parallelism or with the fused q / k epilogue).  The result must equal compile_plan()'s entry for entry
        (tests/test_dit_ops_gpu.py::test_c_built_dit_plan_equals_the_python_built_one); the caller frees it with utx_plan_free."""
        from .._lib import DitConfig, DitDoubleBlock, DitLinear, DitSingleBlock, DitWeights, DitWorkspace
        if self.sp is not None or self.fuse_qk or self.fp8_attention:      # (the C-side builder assembles the bf16 attention only)
            return None
        p = next(iter(self._plans.values())) if p is None else p
        ws, sh, W = p["ws"], self.shape, self.W

        def lin(w, b_, lora=None, blk=None, nm=None):
            L = DitLinear()
            L.w, L.b = ptr(w), ptr(b_)
            if lora is not None:
                A_cat, B_cat, alpha, Rp = lora
                L.lora_A, L.lora_B, L.lora_alpha, L.lora_rp, L.lora_nseg = ptr(A_cat), ptr(B_cat), float(alpha), int(Rp), A_cat.shape[0] // Rp
            if self.fp8_weights and blk is not None and (nm + ".q") in blk:      # the MX fp8 image of the weight (adapters merged in, _quantize_fp8_weights)
                L.q, L.s, L.lds_s = ptr(blk[nm + ".q"]), ptr(blk[nm + ".s"]), blk[nm + ".s"].stride(0)
                sp = blk.get(nm + ".sp")
                if sp is not None:
                    L.sp, L.sp_row_blocks = ptr(sp.data), sp.row_blocks
            return L
        wt = DitWeights()
        wt.x_embedder = lin(self._w("x_embedder.w"), self._w("x_embedder.b"))
        wt.context_embedder = lin(W["context_embedder.w"], W["context_embedder.b"])
        wt.proj_out = lin(W["proj_out.w"], W["proj_out.b"])
        k = "time_text_embed.%s.%s"
        for dst, nm in (("t", "timestep_embedder"), ("g", "guidance_embedder"), ("p", "text_embedder")):
            if nm == "guidance_embedder" and not sh.guidance_embeds:
                continue
            for j in (1, 2):
                setattr(wt, "%s_lin%d" % (dst, j), lin(W[k % (nm, "linear_%d" % j) + ".w"], W[k % (nm, "linear_%d" % j) + ".b"]))
        wt.mod = lin(W["mod.w"], W["mod.b"])
        dbl = (DitDoubleBlock * max(1, len(self.double)))()
        for i, b in enumerate(self.double):
            for nm in ("qkv_x", "qkv_c", "out_x", "out_c", "ff1_x", "ff2_x", "ff1_c", "ff2_c"):
                setattr(dbl[i], nm, lin(b[nm + ".w"], b[nm + ".b"], b.get("lora." + nm), b, nm))
            dbl[i].nq, dbl[i].nk, dbl[i].naq, dbl[i].nak = ptr(b["nq"]), ptr(b["nk"]), ptr(b["naq"]), ptr(b["nak"])
            dbl[i].mod_x, dbl[i].mod_c = self.mod_off[("d", i, "x")], self.mod_off[("d", i, "c")]
        sgl = (DitSingleBlock * max(1, len(self.single)))()
        for i, b in enumerate(self.single):
            sgl[i].qkvm = lin(b["qkvm.w"], b["qkvm.b"], b.get("lora.qkvm"), b, "qkvm")
            sgl[i].out = lin(b["out.w"], b["out.b"], None, b, "out")
            sgl[i].nq, sgl[i].nk, sgl[i].mod = ptr(b["nq"]), ptr(b["nk"]), self.mod_off[("s", i)]
        wt.dbl, wt.sgl = dbl, sgl
        wt.mod_out, wt.n_mod = self.mod_off[("out",)], self.n_mod
        cfg = DitConfig()
        cfg.num_heads, cfg.num_double, cfg.num_single = sh.num_heads, sh.num_double, sh.num_single
        cfg.in_channels, cfg.joint_dim, cfg.pooled_dim, cfg.mlp_ratio, cfg.guidance_embeds = sh.in_channels, sh.joint_dim, sh.pooled_dim, sh.mlp_ratio, int(sh.guidance_embeds)
        cfg.S_txt, cfg.S_img, cfg.n_out = p["S_txt"], p["S_img"], p["n_out"]
        cfg.key_bias_log2, cfg.key_bias_period = float(self.key_bias_log2), int(self.key_bias_period)
        cfg.two_streams = int(bool(self.overlap_text))
        cfg.n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        cfg.lora_rank_padded = int(self.lora_rank) if self._lora_active else 0
        cfg.fp8 = int(bool(self.fp8_weights))
        cfg.fp8_fuse_quant = int(bool(self.fp8_weights) and os.environ.get("UTX_FP8_FUSE_QUANT", "1") != "0")
        w = DitWorkspace()
        for nm in ("lat", "enc", "pooled", "tproj", "gproj", "e1", "e_t", "e_g", "e_p", "temb", "mod", "h", "xn", "qkv", "cat", "attn", "out", "cos", "sin", "Qh", "Kh",
                   "Vt", "T", "Tc"):
            setattr(w, nm, ptr(ws.get(nm)))
        sk, aw = ws.get("sk"), ws.get("attn_ws")
        w.sk_work, w.sk_work_bytes = ptr(sk), 0 if sk is None else sk.numel()
        w.attn_work, w.attn_work_bytes = ptr(aw), 0 if aw is None else aw.numel()
        if self.fp8_weights:
            w.aq, w.as_rm, w.asp, w.asp_row_blocks = ptr(ws["aq"]), ptr(ws["as"]), ptr(ws["asp"]), ws["asp"].stride(0) // 512
            if "aq2" in ws:
                w.aq2, w.asp2, w.asp2_row_blocks = ptr(ws["aq2"]), ptr(ws["asp2"]), ws["asp2"].stride(0) // 512
        h = C.c_void_p()
        self.ctx.check(self.lib.utx_dit_load(self.ctx.handle, C.byref(cfg), C.byref(wt), C.byref(w), C.byref(h)))
        return h

    def forward(self, hidden_states, timestep: float, out: Optional[torch.Tensor] = None):
        """One transformer evaluation.  hidden_states [S_img, 64] bf16 (noise ++ condition tokens);
        `timestep` is the value the pipeline passes (t/1000, already rounded to the latent dtype).
        Positions / conditioning must have been set.  Returns noise_pred [S_img, 64] bf16."""
        p = next(iter(self._plans.values()))
        ws = p["ws"]
        t1000 = _bf16_scalar(_bf16_scalar(timestep) * 1000.0)  # timestep.to(dtype) * 1000 in bf16 [3p]
        ws["tproj"].copy_(self._tproj_device(t1000), non_blocking=True)
        if hidden_states.data_ptr() != ws["lat"].data_ptr():
            ws["lat"].copy_(hidden_states.reshape(ws["lat"].shape))
        g = self._graphs.get(id(p))
        if g is not None:
            g.replay()
        elif p.get("cplan") is not None and self.attn_events is None and self.gemm_events is None:
            if p.get("segments") is not None:
                self._run_segments(p)
            else:
                bad = C.c_int(-1)
                rc = self.lib.utx_plan_run(p["cplan"], self.ctx.stream(), C.byref(bad))
                if rc:
                    raise RuntimeError("utx_plan_run: entry %d failed with code %d" % (bad.value, rc))
        else:
            self.run_plan(p)
        if out is not None:
            out.copy_(ws["out"])
            return out
        return ws["out"]

    def capture_graph(self, warm=True):
        """Record the current plan (~700 stream-ordered launches, fixed descriptors and workspaces, no host syncs) into a
        HIP graph; forward() then replays it.  The two host->device copies of forward (timestep projection, latents) stay
        outside the graph.  One eager run first: first-use setup (kernel attributes, lazily sized buffers) must not happen
        under capture.  Positions, conditioning and LoRA scales are read from device buffers, so they can change between
        replays; a new sequence length or adapter set builds a new plan and drops the graph."""
        p = next(iter(self._plans.values()))
        if getattr(self, "attn_events", None) is not None:
            raise RuntimeError("per-kernel event timing and graph replay are exclusive")
        if warm:        # skip when the plan has already run eagerly (the denoise loop captures after its first step)
            self.run_plan(p)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run_plan(p)
        self._graphs = {id(p): g}
        return g

    def release_graph(self):
        self._graphs = {}

    __call__ = forward
