#!/usr/bin/env python
"""bench.py -- denoising-steps/sec of the UniTEX FLUX-DiT texture denoise loop on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input: the full FLUX.1-dev transformer
forward (19 double + 38 single blocks, texture LoRA rank 64 active, delight LoRA injected with weight 0)
over the joint [text | noise | control | dual] token sequence, plus the fused flow-match Euler update +
condition re-pin.  Inputs (latents, weights) are resident in HBM before the timed region.

Workload (BASELINE.json configs[1], "1024^2 x 6 views, bf16, 1xMI355X"), in the reference's own
joint-strip semantics: one 1024x6144 strip -> 24576 noise + 24576 control + 1024 dual
(512^2 reference image) + 512 text tokens = 50688 tokens, 2454 TFLOP/step.
`--workload ref512x6` is the reference's shipped operating point (512x3072 strip, S = 13824).

N > 1 (default `--parallelism ulysses`): ONE job over the N GPUs, the way north_star asks for a single mesh -- the joint-attention
DiT cannot shard by view, so it runs head-parallel sequence parallelism (unitex_amd/flux/ulysses.py: per-token
work on S/N tokens, attention on 24/N heads over the full sequence, two all-to-alls per layer over RCCL/xGMI): value = steps /
max-over-ranks time, scaling "strong".  The view-sharded back-projection (one view block per rank + ONE all-gather of the atlas
layers) is run and timed after the timed region, with the all-gather reported separately.  `--parallelism replicas` keeps the
round-1 behaviour (N independent meshes, no data-path collective, weak scaling) as a secondary mode.

Contract: python bench.py --gpus N --steps K --warmup W  -> rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this image needs dmabuf IPC (without it RCCL fails with `hipIpcGetMemHandle: invalid argument`); the image exports the variable already -- a launcher
# that scrubbed the environment must not cost the first multi-GPU run.  Set before the HIP runtime loads; an explicit value of the caller stands.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

WORKLOADS = {
    # name: (strip_h, strip_w, dual_px, description)
    "strip1024x6": (1024, 6144, 512, "FLUX.1-dev + texture LoRA r64, joint strip 1024x6144 (6 views @1024^2) + control strip + 512^2 dual"),
    "ref512x6": (512, 3072, 512, "FLUX.1-dev + texture LoRA r64, joint strip 512x3072 (reference operating point) + control + 512^2 dual"),
    "strip2048x8": (2048, 16384, 512, "FLUX.1-dev + texture LoRA r64, joint strip 2048x16384 (8 views @2048^2, the reference's joint semantics carried to BASELINE configs[4]'s resolution: 263 680 tokens) + control strip + 512^2 dual"),
    "view2048": (2048, 2048, 512, "FLUX.1-dev + texture LoRA r64, single 2048^2 view + control + 512^2 dual (BASELINE configs[4] per-view shape; NOT the reference's joint semantics)"),
    "view1024": (1024, 1024, 512, "FLUX.1-dev + texture LoRA r64, single 1024^2 view + control + 512^2 dual (per-view variant; NOT the reference's joint semantics)"),
}
D, HEADS, N_DOUBLE, N_SINGLE = 3072, 24, 19, 38
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, MI355X_MICROARCH.md


def token_counts(name):
    h, w, dual, _ = WORKLOADS[name]
    n_noise = (h // 16) * (w // 16)
    n_dual = (dual // 16) ** 2
    return 512, n_noise, n_noise, n_dual


def step_flops(S):
    attn = 4.0 * S * S * D * (N_DOUBLE + N_SINGLE)
    lin = 24.0 * D * D * S * (N_DOUBLE + N_SINGLE)
    return attn + lin, attn


def cpu_baseline(S_full, threads, full_depth=False):
    """Oracle (oracle/dit_ref.py, fp32 torch CPU) at BASELINE configs[0]'s shape -- full width (D = 3072, 24 heads), 512 text + 4096 noise +
    4096 control + 1024 dual = 9728 tokens -- on a bounded sample of its depth: 1 double + 1 single block (2 / 57 of a configs[0] step; every
    block has the same attention : linear ratio, 34.5 % : 65.5 % at this length, so the sample has the step's own mix), about 20-30 s on the
    GPU box's host.  The attention core is timed once more on its own (oracle sdpa, 24 heads x 9728 tokens), which gives separate attention and
    linear rates; the S_full-token step (attention share 73 %) is extrapolated with THOSE two rates, not with one blended FLOP rate.
    full_depth=True (bench.py --cpu-full-step, ~9 min): the whole 57-block step at 9728 tokens, nothing about depth extrapolated."""
    from oracle import dit_ref
    torch.set_num_threads(threads)
    nd, ns = (N_DOUBLE, N_SINGLE) if full_depth else (1, 1)
    cfg = dit_ref.FluxConfig(num_double=nd, num_single=ns)
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0, dtype=torch.float32)
    HL, WL, DL, S_txt = 32, 128, 32, 512
    g = torch.Generator().manual_seed(63)
    img_ids = torch.cat([dit_ref.latent_image_ids(HL, WL), dit_ref.latent_image_ids(HL, WL, offset_y=HL),
                         dit_ref.latent_image_ids(DL, DL, offset_x=WL, offset_y=HL)], 0)
    S_img = img_ids.shape[0]
    lat = torch.randn(S_img, 64, generator=g)
    enc = torch.zeros(S_txt, cfg.joint_dim)
    pooled = torch.zeros(1, cfg.pooled_dim)
    txt_ids = torch.zeros(S_txt, 3)
    t0 = time.perf_counter()
    dit_ref.flux_forward(sd, cfg, lat, enc, pooled, 0.5, 3.5, txt_ids, img_ids, emulate_bf16=False)
    dt = time.perf_counter() - t0
    S = S_txt + S_img
    q = torch.randn(HEADS, S, 128, generator=g)
    t1 = time.perf_counter()
    dit_ref.sdpa(q, q, q, False)
    dt_attn1 = time.perf_counter() - t1                      # one attention call of the sample
    nb = nd + ns
    fl_attn1, fl_lin1 = 4.0 * S * S * D, 24.0 * D * D * S
    t_attn = min(dt_attn1 * nb, 0.9 * dt)
    rate_attn, rate_lin = fl_attn1 * nb / t_attn, fl_lin1 * nb / (dt - t_attn)
    full, full_attn = step_flops(S_full)
    t_full = full_attn / rate_attn + (full - full_attn) / rate_lin
    return {"value": 1.0 / t_full, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "oracle/dit_ref.py fp32 torch-CPU at BASELINE configs[0]'s shape (D=3072, 24 heads, S=%d): %d double + %d single FLUX blocks%s, "
                      "%.1f s measured = %.2f TFLOP/s blended; attention core alone %.2f s per call -> %.2f TFLOP/s attention, %.2f TFLOP/s linears + "
                      "elementwise; the %d-token 57-block step (attention %.0f %% of its FLOPs) extrapolated with those two rates"
                      % (S, nd, ns, " = one COMPLETE 57-block step, depth not extrapolated" if full_depth else " (2/57 of a step)", dt,
                         (fl_attn1 + fl_lin1) * nb / dt / 1e12, dt_attn1, rate_attn / 1e12, rate_lin / 1e12, S_full, 100.0 * full_attn / full),
            "measured_seconds": dt, "config0_step_seconds": dt * (N_DOUBLE + N_SINGLE) / nb, "config0_step_is_measured": bool(full_depth),
            "attention_tflops": rate_attn / 1e12, "linear_tflops": rate_lin / 1e12,
            "full_step_reference": "profiles/r03_cpu_full_step.json (bench.py --cpu-full-step on a GPU box's host), when present"}


def host_cores():
    """(threads usable by this process, physical cores from lscpu or None)."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = None
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":")[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        phys = int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
    except Exception:  # noqa: BLE001 -- lscpu missing / unparsable: report None, never guess
        pass
    return ncpu, phys


def cpu_config1(threads):
    """BASELINE configs[0] run to COMPLETION on the host cores (the one CPU number that is not extrapolated):
    oracle/dit_ref.py fp32, full FLUX.1-dev shape (19 + 38 blocks, D = 3072), 512 x 2048 strip of 4 views + control strip +
    512^2 dual + 512 text tokens = 9728 tokens, 4 flow-match Euler steps with the condition re-pin."""
    from oracle import dit_ref
    torch.set_num_threads(threads)
    cfg = dit_ref.FluxConfig()
    sd = dit_ref.make_synthetic_state_dict(cfg, seed=0, dtype=torch.float32)
    HL, WL, DL = 32, 128, 32
    g = torch.Generator().manual_seed(63)
    noise = torch.randn(HL * WL, 64, generator=g)
    cond = torch.randn(HL * WL + DL * DL, 64, generator=g)
    img_ids = torch.cat([dit_ref.latent_image_ids(HL, WL), dit_ref.latent_image_ids(HL, WL, offset_y=HL),
                         dit_ref.latent_image_ids(DL, DL, offset_x=WL, offset_y=HL)], 0)
    t0 = time.perf_counter()
    out = dit_ref.denoise_loop(sd, cfg, noise, cond, torch.zeros(512, cfg.joint_dim), torch.zeros(1, cfg.pooled_dim),
                               torch.zeros(512, 3), img_ids, 4, guidance=3.5, emulate_bf16=False)
    dt = time.perf_counter() - t0
    S = 512 + noise.shape[0] + cond.shape[0]
    fl, _ = step_flops(S)
    return {"seconds": dt, "steps": 4, "steps_per_s": 4.0 / dt, "tokens": S, "tflops": 4.0 * fl / dt / 1e12, "cores": threads,
            "finite": bool(torch.isfinite(torch.as_tensor(out)).all()), "kind": "port",
            "sample": "BASELINE configs[0] complete: 4 steps x 57 blocks at S = %d, fp32, not extrapolated" % S}


def cpu_baseline_backprojection():
    """oracle/geom_ref (single-threaded C + numpy) on a bounded sample of the back-projection: 20k-face mesh, six
    256^2 views, 512^2 atlas (1/16 of the texels of the GPU measurement); scaled by texel count."""
    import numpy as np
    from oracle import geom_ref as G
    from unitex_amd.texturetools import meshes
    from unitex_amd.texturetools.benchmarks import smooth_views
    T, HW = 512, 256
    v, f, uv = meshes.sphere_with_faces(20000)
    c2ws = G.box_views_c2ws(2.8)[[0, 1, 4, 2, 3, 5]]
    mvp = G.mvp_matrices(c2ws, G.intrinsics(1.0, 1.0, fov=False), False)
    t0 = time.perf_counter()
    clip = G.transform_points(v, mvp)
    vndc = (clip[..., :2] / clip[..., 3:4]).astype(np.float32)
    alpha = np.stack([(G.rasterize(clip[i], f, HW, HW)[..., 3] > 0).astype(np.float32) for i in range(6)])
    imgs = np.concatenate([smooth_views(6, HW, HW), alpha[..., None]], -1).astype(np.float32)
    uvclip = np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32), np.ones((len(uv), 1), np.float32)], -1)
    rast2d = G.rasterize(uvclip, f, T, T)
    mask2d = rast2d[..., 3] > 0
    bvh = G.BVH(v, f)
    col, rv, ao = G.backproject(rast2d, v, f, G.face_normals(v, f), vndc, (-c2ws[:, :3, 2]).astype(np.float32), imgs, bvh, angle_deg=100.0)
    vis = G.dilate_visibility(rv, mask2d, ao)
    atlas, seen, win, bnd = G.composite(col, vis)
    filled, _ = G.nn_fill(atlas, seen, mask2d, G.interpolate(v, rast2d, f))
    blur = G.lens_blur_collapsed(filled, G.seam_mask(bnd, mask2d))
    G.tensor_to_u8(G.pull_push(blur.transpose(2, 0, 1), mask2d).transpose(1, 2, 0))
    dt = time.perf_counter() - t0
    return {"seconds_sample": dt, "seconds_scaled_to_2048": dt * 16.0, "cores": 1, "kind": "port",
            "sample": "oracle/geom_ref.c + numpy, 20k faces, 6 x 256^2 views, 512^2 atlas (x16 texels -> 2048^2)"}


VISIBLE_ENV = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def device_identity(index):
    """what one rank knows about the GPU it sits on: (the *_VISIBLE_DEVICES strings that number the devices of this process, its device index,
    the hardware identity torch reports: uuid string + PCI domain / bus / device)."""
    vis = tuple(os.environ.get(k, "") for k in VISIBLE_ENV)
    hw = None
    try:
        pr = torch.cuda.get_device_properties(index)
        hw = (str(getattr(pr, "uuid", "")), int(getattr(pr, "pci_domain_id", -1)), int(getattr(pr, "pci_bus_id", -1)), int(getattr(pr, "pci_device_id", -1)))
    except Exception:  # noqa: BLE001 -- an identity that cannot be read is reported as unknown, never guessed
        pass
    return {"visible": vis, "index": int(index), "hw": hw}


def count_distinct_devices(idents):
    """(number of distinct GPUs under the ranks of a job, how it was counted) from the gathered device_identity() records.  Ranks that number the
    same device list (identical *_VISIBLE_DEVICES -- what torch.distributed.run gives its workers) are on distinct GPUs exactly when their device
    indices differ: that count needs nothing from the driver.  Ranks with different device lists are told apart by the hardware identity (PCI
    address + uuid); if that is missing or the same placeholder everywhere while the lists differ, the answer is (None, ...) -- unknown is reported,
    not turned into a refusal: an all-zero uuid must not abort a valid 8-GPU run."""
    if all(i["visible"] == idents[0]["visible"] for i in idents):
        return len({i["index"] for i in idents}), "device index under identical *_VISIBLE_DEVICES"
    hws = [i["hw"] for i in idents]
    if any(h is None for h in hws):
        return None, "unknown (hardware identity unreadable on some rank)"
    if len(set(hws)) == 1 and len({(i["visible"], i["index"]) for i in idents}) > 1:
        return None, "unknown (every rank reports the same uuid / PCI address under different device lists)"
    return len(set(hws)), "uuid + PCI address"


def attention_loop_ab(dev, S_exec, heads):
    """Beside the line, never part of `value`: the 4 x 64 attention kernel (UTX_ATTN_Q64=1, the default since round 6: one wave per SIMD, generated hand-placed stream) against
    the 8 x 32 fast loop (=0, round 5's default) on this workload's shape, same process, interleaved, behind every other measurement: bit identity + ms per call.  Both are
    validated kernels (tests/test_attention_q64_gpu.py); the option's previous value is restored."""
    from unitex_amd import _lib
    from unitex_amd.flux import ops
    g = torch.Generator(device=dev).manual_seed(S_exec)
    S_pad = (S_exec + 63) // 64 * 64
    Qh = (torch.randn(heads, S_pad, 128, generator=g, device=dev) * (1.4426950408889634 / math.sqrt(128.0))).to(torch.bfloat16)
    Kh = torch.randn(heads, S_pad, 128, generator=g, device=dev).to(torch.bfloat16)
    Vt = torch.randn(heads, 128, S_pad, generator=g, device=dev).to(torch.bfloat16)
    o = torch.empty(S_exec, heads * 128, dtype=torch.bfloat16, device=dev)
    res, outs = {}, {}
    prev = _lib.get_options()["UTX_ATTN_Q64"]
    try:
        ms = {0: [], 1: []}
        for rnd in range(3):
            for arm in (0, 1):
                _lib.set_option("UTX_ATTN_Q64", arm)
                ops.attention(Qh, Kh, Vt, S=S_exec, scale=0.0, key_bias_log2=3.0, out=o)
                if rnd == 0:
                    torch.cuda.synchronize()
                    outs[arm] = o.clone()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(2):
                    ops.attention(Qh, Kh, Vt, S=S_exec, scale=0.0, key_bias_log2=3.0, out=o)
                b.record()
                torch.cuda.synchronize()
                ms[arm].append(a.elapsed_time(b) / 2.0)
        fl = 4.0 * S_exec * S_exec * 128 * heads
        res = {"fast8x32_ms": sorted(ms[0])[1], "q64_ms": sorted(ms[1])[1], "bit_identical": bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))),
               "fast8x32_tflops": fl / sorted(ms[0])[1] / 1e9, "q64_tflops": fl / sorted(ms[1])[1] / 1e9, "tokens": S_exec, "heads": heads}
    finally:
        _lib.set_option("UTX_ATTN_Q64", prev)
    return res


def ref_point_ms(model, ops, sched_unused, calculate_shift, shape, dev, prune, steps=3):
    """ms per denoise step of the SAME model at the reference's own operating point (WORKLOADS['ref512x6']: 512^2 x 6 views, 13 824 tokens), 1 warm-up + `steps` timed."""
    from unitex_amd.flux.scheduler import FlowMatchEulerScheduler
    S_txt, n_noise, n_ctrl, n_dual = token_counts("ref512x6")
    h_px, w_px, dual_px, _ = WORKLOADS["ref512x6"]
    S_img = n_noise + n_ctrl + n_dual
    g = torch.Generator(device=dev).manual_seed(64)
    lat = torch.randn(S_img, 64, generator=g, device=dev).to(torch.bfloat16)
    cond = lat[n_noise:].clone()
    HL, WL = h_px // 16, w_px // 16
    ids = [torch.zeros(HL, WL, 3), torch.zeros(HL, WL, 3), torch.zeros(dual_px // 16, dual_px // 16, 3)]
    for t, (oy, ox) in zip(ids, [(0, 0), (HL, 0), (HL, WL)]):
        t[..., 1] += torch.arange(oy, oy + t.shape[0])[:, None]
        t[..., 2] += torch.arange(ox, ox + t.shape[1])[None, :]
    model.set_positions(torch.zeros(S_txt, 3), torch.cat([t.reshape(-1, 3) for t in ids], 0))
    if prune:
        model.set_output_rows(n_noise)
    model.set_conditioning(torch.zeros(S_txt, shape.joint_dim, device=dev), torch.zeros(1, shape.pooled_dim, device=dev), 3.5)
    sched = FlowMatchEulerScheduler()
    ts = sched.set_timesteps(28, calculate_shift(n_noise))

    def step(i):
        t_bf = torch.tensor(float(ts[i]), dtype=torch.float32).to(torch.bfloat16)
        v = model.forward(lat, float((t_bf / 1000).to(torch.float32)))
        ops.sched_step(lat, v, sched.dsigma(i), n_noise_tokens=n_noise, cond=cond)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, 1 + steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _gemm_census(model):
    try:
        return model.gemm_census()
    except Exception as e:  # noqa: BLE001 -- a reporting extra must never cost the bench line
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("UTX_WORKLOAD", "strip1024x6"), choices=sorted(WORKLOADS))
    ap.add_argument("--lora-rank", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the per-step plan as one HIP graph (FluxDiT.capture_graph) in the timed region; the roofline "
                         "kernel is then timed in one extra eager step after it (events cannot bracket kernels inside a graph)")
    ap.add_argument("--parallelism", default=os.environ.get("UTX_PARALLELISM", "ulysses"), choices=["replicas", "ulysses"],
                    help="N > 1: 'ulysses' (default) = ONE job, head-parallel sequence parallelism with two all-to-alls per layer "
                         "over RCCL (strong scaling) + view-sharded back-projection; 'replicas' = N independent jobs (weak scaling)")
    ap.add_argument("--sp-self-test", action="store_true",
                    help="diagnostic, N = 1 only: run the sequence-parallel plan on a 1-rank NCCL group with every collective issued (all-to-alls to itself, "
                         "UTX_SP_FORCE_A2A=1): the step's non-fabric cost of the exchange machinery (RCCL launches, copies, unpack kernels) against the plain step")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[4] numerics: the five big linears on OCP MX fp8 operands (utx_gemm_desc.mx8); NOT the default "
                         "bench line (the metric is quoted in bf16) -- reported with dtype 'mx-fp8 linears + bf16 attention'")
    ap.add_argument("--fp8-attn", action="store_true",
                    help="with --fp8: QK^T / PV on the fp8 matrix pipe as well (utx_attn_fwd_fp8, opt-in; N = 1).  Reported with its own dtype string; the "
                         "roofline object then prices the attention launch against the 5 PF fp8 nameplate")
    ap.add_argument("--cpu-full-step", action="store_true",
                    help="CPU baseline on ONE COMPLETE 57-block step at BASELINE configs[0]'s shape (S = 9728) instead of 1 + 1 blocks: ~9 min of "
                         "host time, so not the default")
    ap.add_argument("--cpu-config1", action="store_true",
                    help="also run BASELINE configs[0] (512^2 x 4 views, S = 9728, 4 denoise steps, fp32) to COMPLETION on the host "
                         "cores with the oracle (~35 min on the GPU box's 64 threads; 50 min on the 8-core build container: profiles/r02_cpu_config1_container.json): the one CPU number that is not extrapolated (SURVEY 8d)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device visible (there is no CPU fallback)")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: become the N-rank job (one process per GPU over RCCL) instead of silently measuring one GPU
        if torch.cuda.device_count() < args.gpus and os.environ.get("UTX_DIST_BACKEND", "nccl") == "nccl":
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, torch.cuda.device_count()))
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the job's rank count and --gpus must agree" % (args.gpus, world))
    if local_rank >= torch.cuda.device_count():   # fewer visible devices than ranks (single-GPU control-flow test only)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("UTX_DIST_BACKEND", "nccl")   # "gloo" only to exercise the N > 1 control flow on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
        # what the job really is, from the group itself: ranks counted by a collective, and the devices they sit on (distinct GPUs, or the line says so)
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.float32)
        dist.all_reduce(one)
        idents = [None] * world
        dist.all_gather_object(idents, device_identity(local_rank))
        distinct, how = count_distinct_devices(idents)
        group_info = {"backend": dist.get_backend(), "world_size_from_group": dist.get_world_size(), "ranks_counted_by_all_reduce": int(one.item()),
                      "distinct_devices": distinct, "distinct_devices_from": how}
        if group_info["ranks_counted_by_all_reduce"] != args.gpus or group_info["world_size_from_group"] != args.gpus:
            raise SystemExit("bench.py --gpus %d: the process group holds %d rank(s)" % (args.gpus, group_info["ranks_counted_by_all_reduce"]))
        if backend == "nccl" and distinct is not None and distinct != world:
            raise SystemExit("bench.py --gpus %d: the ranks share devices (%d distinct, by %s): one process per GPU is the contract" % (args.gpus, distinct, how))
    else:
        group_info = None

    from unitex_amd.flux import ops
    from unitex_amd.flux.synthetic import SyntheticFluxStateDict, synthetic_lora
    from unitex_amd.flux.transformer import FluxDiT, FluxShape
    from unitex_amd.flux.scheduler import FlowMatchEulerScheduler, calculate_shift

    S_txt, n_noise, n_ctrl, n_dual = token_counts(args.workload)
    S_img = n_noise + n_ctrl + n_dual
    S = S_txt + S_img
    h_px, w_px, dual_px, desc = WORKLOADS[args.workload]

    shape = FluxShape()
    sd = SyntheticFluxStateDict(shape, seed=0, device=dev)
    ulysses = args.parallelism == "ulysses" and world > 1
    if args.sp_self_test:
        if world != 1:
            raise SystemExit("--sp-self-test is a single-GPU diagnostic")
        import torch.distributed as dist
        os.environ["UTX_SP_FORCE_A2A"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        ulysses = True
    model = FluxDiT(sd, shape, device=dev, sequence_parallel=ulysses, fp8_weights=args.fp8, fp8_attention=args.fp8_attn)
    tex = synthetic_lora(sd, shape, rank=args.lora_rank, seed=1, device=dev)
    dlt = synthetic_lora(sd, shape, rank=args.lora_rank, seed=2, device=dev)
    model.set_lora([(tex, 1.0), (dlt, 0.0)])  # reference: weights_for_texture = [1, 0] (pipeline.py:110)

    # synthetic latents: seed 63 (run.py:5) + rank so replicas differ
    g = torch.Generator(device=dev).manual_seed(63 + (0 if ulysses else rank))
    lat = torch.randn(S_img, 64, generator=g, device=dev).to(torch.bfloat16)
    cond = lat[n_noise:].clone()
    n_noise_sched = n_noise
    if ulysses:   # this rank's contiguous slice of the image tokens; the scheduler step / re-pin are per token
        i0, i1 = model.local_image_range(S_img)
        cond = lat[max(i0, n_noise):i1].clone() if i1 > n_noise else None
        n_noise_sched = min(max(n_noise - i0, 0), i1 - i0)
        lat = lat[i0:i1].clone()
    HL, WL = h_px // 16, w_px // 16
    ids = [torch.zeros(HL, WL, 3), torch.zeros(HL, WL, 3), torch.zeros(dual_px // 16, dual_px // 16, 3)]
    offs = [(0, 0), (HL, 0), (HL, WL)]
    for t, (oy, ox) in zip(ids, offs):
        t[..., 1] += torch.arange(oy, oy + t.shape[0])[:, None]
        t[..., 2] += torch.arange(ox, ox + t.shape[1])[None, :]
    img_ids = torch.cat([t.reshape(-1, 3) for t in ids], 0)
    prune = os.environ.get("UTX_PRUNE_LAST", "1") != "0" and not ulysses
    model.set_positions(torch.zeros(S_txt, 3), img_ids)
    if prune:
        model.set_output_rows(n_noise)    # as the texturing pipeline does: only the noise tokens' prediction is consumed (sched_step reads no other row)
    model.set_conditioning(torch.zeros(S_txt, shape.joint_dim, device=dev), torch.zeros(1, shape.pooled_dim, device=dev), 3.5)
    sched = FlowMatchEulerScheduler()
    total = args.warmup + args.steps
    nsched = max(28, total)
    ts = sched.set_timesteps(nsched, calculate_shift(n_noise))

    def one_step(i, events=None):
        t_bf = torch.tensor(float(ts[i]), dtype=torch.float32).to(torch.bfloat16)
        t_in = float((t_bf / 1000).to(torch.float32))
        model.attn_events = events
        v = model.forward(lat, t_in)
        ops.sched_step(lat, v, sched.dsigma(i), n_noise_tokens=n_noise_sched, cond=cond)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for i in range(args.warmup):
        one_step(i)
    use_graph = args.graph and not ulysses
    if use_graph:
        model.attn_events = None
        model.capture_graph()
        one_step(0)
    # THE TIMED REGION RUNS WHAT THE PRODUCT RUNS: FluxDiT.forward's default launch path -- the C-side replay of the step's plan (utx_plan_run: one C call
    # per step, no per-kernel events), or the HIP graph with --graph; under sequence parallelism the Python launch list (its collectives are torch.distributed
    # calls).  The per-kernel durations of the roofline objects come from ONE extra step each behind the timed region, through the evented launch list.
    launch_path = "hip graph replay" if use_graph else \
        ("python launch list (ctypes call per kernel)" if next(iter(model._plans.values())).get("cplan") is None else
         "C plan ranges between the host's collectives (utx_plan_run_range; the all-to-alls are torch.distributed calls)" if ulysses else
         "C plan replay (utx_plan_run, one C call per step)")
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        one_step(i, None)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_graph:
        model.release_graph()
    events = []
    one_step(total - 1, events)      # the dominant kernel's launches of one step, HIP events on the launch stream around each
    torch.cuda.synchronize()
    model.attn_events = None
    # the same steps replayed as ONE HIP graph (what UTX_HIP_GRAPH=1 makes the denoise loop do), reported beside the default path
    graph_ms = None
    if not ulysses and not use_graph and os.environ.get("UTX_BENCH_GRAPH_FIGURE", "1") != "0":
        try:
            model.capture_graph(warm=False)
            one_step(0)
            torch.cuda.synchronize()
            n_graph = min(args.steps, 4)          # a figure beside the line, not a second timed region
            tg = time.perf_counter()
            for i in range(args.warmup, args.warmup + n_graph):
                one_step(i, None)
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - tg) / n_graph * 1e3
        except Exception as e:  # noqa: BLE001 -- a reporting extra
            graph_ms = "error: %r" % (e,)
        finally:
            model.release_graph()
    # GEMM-only roofline: ONE extra eager step behind the timed region with HIP events around every large-M GEMM of the main stream
    # (the text-side GEMMs of the double blocks run beside them on the second stream, as in the timed steps)
    gemm_ev = []
    model.gemm_events = gemm_ev
    one_step(total - 1, None)
    torch.cuda.synchronize()
    model.gemm_events = None
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(lat.float()).all(), "non-finite latents after the timed steps"

    # N > 1: everything below is reporting -- and holds collectives (the exchange timing, the view-sharded back-projection's all-gather, the group's tear-down) that no multi-GPU
    # node has run yet.  A collective that hangs would take the ONE stdout line of an otherwise measured job with it, so a watchdog on every rank ends the job cleanly after
    # UTX_BENCH_EXTRAS_TIMEOUT seconds (default 420): rank 0 prints the contract's line from the timed region alone and says so in it; cancelled where the full line is printed.
    watchdog = None
    if world > 1:
        import threading

        def _extras_timed_out():
            if rank == 0:
                S_x = S if model.text_rows is None else model.text_rows * (world if ulysses else 1) + S_img
                jobs = 1 if ulysses else world
                print(json.dumps({"metric": "denoising-steps/sec", "value": args.steps * jobs / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                  "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if ulysses else "weak", "vs_baseline": None, "dtype": "bf16",
                                  "data": "synthetic", "config": {"workload": args.workload, "parallelism": args.parallelism, "tokens_executed": S_x, "launch_path": launch_path,
                                                                  "note": "the reporting extras behind the timed region did not finish within the watchdog's limit: this line carries the "
                                                                          "timed region only (no roofline / exchange / back-projection objects)"}}))
                sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("UTX_BENCH_EXTRAS_TIMEOUT", "420")), _extras_timed_out)
        watchdog.daemon = True
        watchdog.start()

    # ---- N > 1, one job: un-overlapped cost of the two exchanges of a layer on the real buffers (outside the timed region),
    # and the view-sharded back-projection with its ONE all-gather -- every rank takes part, rank 0 reports
    exchange = None
    if ulysses:
        ex = model.ex
        tmp = torch.empty(ex.S_loc, shape.dim, dtype=torch.bfloat16, device=dev)
        reps = 8
        def heads_in():      # zero copy (UTX_SP_ZERO_COPY=1): the exchange alone -- no relayout pass behind it in the step either
            if ex.zero_copy:
                for w_ in ex.start_heads_in():
                    if w_ is not None:
                        w_.wait()
            else:
                ex.heads_in()
        for _ in range(2):
            heads_in(); ex.tokens_out(tmp)
        torch.cuda.synchronize(); barrier()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_in = t_out = 0.0
        for _ in range(reps):
            e[0].record(); heads_in(); e[1].record(); ex.tokens_out(tmp); e[2].record()
            torch.cuda.synchronize()
            t_in += e[0].elapsed_time(e[1]); t_out += e[1].elapsed_time(e[2])
        exchange = {"qkv_exchange": "zero copy: attention reads the receive buffer (utx_attn_fwd_bf16_blk)" if ex.zero_copy else "all-to-all + relayout pass",
                    "qkv_all_to_all_plus_unpack_ms": t_in / reps, "out_all_to_all_plus_unpack_ms": t_out / reps,
                    "bytes_per_rank_per_layer": ex.bytes_per_layer, "layers": N_DOUBLE + N_SINGLE, "head_groups": ex.G,
                    "note": "measured back to back without compute; in the step the Q/K/V exchange of the 38 single blocks runs beside the MLP half of the projection GEMM"}
    bp = None
    if world > 1:
        try:
            from unitex_amd.texturetools.benchmarks import time_backprojection
            bp = time_backprojection(50000, 1024, 2048, iters=2, warmup=1, device=dev, view_shard=(rank, world))
        except Exception as ex_:  # noqa: BLE001 -- the DiT number must still be reported
            bp = {"error": repr(ex_)}
        barrier()

    if rank == 0:
        from unitex_amd import _lib
        # text-token dedup (flux/transformer.py): the 512 identical text tokens of the reference are carried as 64 rows per rank
        # whose keys count 8-fold -- FLOPs below are the EXECUTED ones (S_exec tokens), never the nominal 50 688-token figure
        S_exec = S if model.text_rows is None else model.text_rows * (world if ulysses else 1) + S_img
        fl, fl_attn = step_flops(S_exec)
        fl_nominal, _ = step_flops(S)
        attn_ms = [a.elapsed_time(b) for a, b in events]
        attn_avg_ms = sum(attn_ms) / max(len(attn_ms), 1)
        # sequence parallel with the ranks' identical text rows kept ONCE among the keys (FluxDiT.sp_kv_dedup, round 6): S_exec queries over ex.S_k keys
        S_keys = model.ex.S_k if ulysses else S_exec
        if S_keys != S_exec:
            d_attn = 4.0 * S_exec * (S_exec - S_keys) * D * (N_DOUBLE + N_SINGLE)
            fl -= d_attn; fl_attn -= d_attn
        attn_launch_flops = 4.0 * S_exec * S_keys * 128 * (model.ex.Hg if ulysses else HEADS)     # sequence parallel: one launch per head group of Hg heads
        n_attn = 57 * (model.ex.G if ulysses else 1)
        if prune:
            # last-block pruning (FluxDiT.set_output_rows): the last of the 57 attention calls has n_noise queries instead of S_exec, and the
            # last block's q / MLP / output projections run on n_noise rows -- the FLOP figures are the EXECUTED ones
            dead = S_exec - n_noise
            attn_last = 4.0 * n_noise * S_keys * 128 * HEADS
            fl -= (attn_launch_flops - attn_last) + 2.0 * dead * 3072 * (3072 + 4 * 3072) + 2.0 * dead * (5 * 3072) * 3072
            fl_attn -= (attn_launch_flops - attn_last)
            attn_launch_flops = (attn_launch_flops * (n_attn - 1) + attn_last) / n_attn      # mean over the step's calls, as attn_avg_ms is
        achieved = attn_launch_flops / (attn_avg_ms * 1e-3) / 1e12
        value = (1 if ulysses else world) * args.steps / dt
        par = "DIAGNOSTIC --sp-self-test: the sequence-parallel plan on ONE rank, every all-to-all issued to itself through RCCL (not a speed)" if args.sp_self_test else \
            ("ulysses sp%d: ONE job, 2 all-to-alls / layer (RCCL) + view-sharded back-projection with one all-gather" % world) if ulysses \
            else ("single GPU" if world == 1 else "replicas x%d (independent jobs, no data-path collective)" % world)
        out = {
            "metric": "denoising-steps/sec", "value": value, "unit": "steps/s", "n_gpus": (group_info["ranks_counted_by_all_reduce"] if group_info else 1),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (world > 1 and not ulysses) else "strong", "vs_baseline": None,
            "dtype": ("mx-fp8 (e4m3 x E8M0/32) big linears + mx-fp8 attention (opt-in)" if args.fp8_attn else "mx-fp8 (e4m3 x E8M0/32) big linears + bf16 attention") if args.fp8
            else ("bf16 linears + mx-fp8 attention (opt-in)" if args.fp8_attn else "bf16"), "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "tokens": S, "text_tokens": S_txt,
                       "noise_tokens": n_noise, "control_tokens": n_ctrl, "dual_tokens": n_dual,
                       "lora_rank": args.lora_rank, "guidance": 3.5, "launch": launch_path, "ms_per_step_hip_graph_replay": graph_ms, "parallelism": par,
                       "text_half_of_double_blocks": "second HIP stream beside the image half (opt-in: UTX_TXT_STREAM=1)" if model.overlap_text else
                       "on the caller's stream (default since round 5: the two-stream form's rare corruption has no established mechanism, DESIGN 9 b; costs 0.5 % here)",
                       "tokens_computed": S_exec, "attention_keys": S_keys,
                       "sp_key_dedup": ("the ranks' identical text rows once among the keys (utx_sp_unpack_qkv_dedup): %d queries over %d keys per launch, key multiplicity on tile 0 only"
                                        % (S_exec, S_keys)) if (ulysses and model.sp_kv_dedup) else None,
                       "text_dedup": None if model.text_rows is None else "512 identical text tokens carried as %d rows per rank, key weight 2^%.2f (SURVEY 7 last bullet; UTX_TEXT_DEDUP=0 disables)" % (model.text_rows, model.key_bias_log2),
                       "last_block_pruning": ("last block: queries / MLP / out-projection for the %d noise tokens only (the prediction of the condition tail is never read: "
                                              "flux_piplines/texturing/pipeline.py:645,660,684; UTX_PRUNE_LAST=0 disables)" % n_noise) if prune else None,
                       "tflop_per_step": fl / 1e12, "tflop_per_step_reference_semantics": fl_nominal / 1e12,
                       "achieved_tflops_per_gpu": fl / (dt / args.steps) / 1e12 / (world if ulysses else 1),
                       "sec_per_mesh_texture_dit_only": 56.0 * dt / args.steps,
                       # every switch that could change what was measured: the library's launch options (all result-preserving; the
                       # wrong-result ablations do not exist in this library) and every UTX_* variable of the environment
                       "launch_options": _lib.get_options(), "gemm_launches_per_step": _gemm_census(model), "ablation_build": bool(_lib.load_library().utx_is_ablation_build()),
                       "env_UTX": {k: v for k, v in sorted(os.environ.items()) if k.startswith("UTX_")}},
            "roofline": {"bound": "mfma", "kernel": "attn_fwd_fp8_kernel (+ three MX quantiser passes, not in the launch time)" if args.fp8_attn else ("attn_fwd_q64_kernel (4 x 64, generated stream; + flag memset + repair-pass launch of attn_fwd_glds_kernel)" if _lib.get_options().get("UTX_ATTN_Q64") == 1 and (not ulysses or model.sp_kv_dedup) else "attn_fwd_glds_kernel"), "launches_per_call": "full rounds + key-split tail round (same kernel) + attn_merge_kernel; a 'launch' below is one utx_attn_fwd_bf16 call", "achieved": achieved, "peak": 5000.0 if args.fp8_attn else PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / (5000.0 if args.fp8_attn else PEAK_BF16_TFLOPS), "traffic": None,
                         "launches_timed": len(attn_ms), "avg_launch_ms": attn_avg_ms,
                         "timed_in": "one extra step behind the timed region, HIP events on the launch stream around every attention call (the timed region itself carries no events: it is the product's launch path)",
                         "flops_per_launch": attn_launch_flops,
                         # context, not the judged fraction: an MFMA-only stream of the same shape sustains 1.73 PF on random operands
                         # on this board (power-limited clock, profiles/r01_perf_attn_q64.log)
                         "sustained_mfma_only_tflops": 1730.0, "frac_of_sustained": achieved / 1730.0,
                         "attention_share_of_step_time": (attn_avg_ms * n_attn) / (dt / args.steps * 1e3)},
        }
        if gemm_ev:
            g_ms = sum(a.elapsed_time(b) for a, b, _ in gemm_ev)
            g_fl = sum(f for _, _, f in gemm_ev)
            peak = 5000.0 if args.fp8 else PEAK_BF16_TFLOPS
            lin_nominal = 24.0 * D * D * S_exec * (N_DOUBLE + N_SINGLE) / (world if ulysses else 1)
            out["roofline_gemm"] = {"bound": "mfma", "kernel": "gemm256_w4_kernel (+ gemm_w4_fixup_kernel; MX form with --fp8)", "achieved": g_fl / (g_ms * 1e-3) / 1e12,
                                    "peak": peak, "unit": "TFLOP/s", "frac": g_fl / (g_ms * 1e-3) / 1e12 / peak,
                                    "launches_timed": len(gemm_ev), "sum_ms_per_step": g_ms, "flops_timed": g_fl,
                                    "flops_linears_nominal_24D2S57": lin_nominal, "share_of_linear_flops_timed": g_fl / lin_nominal,
                                    "timed_in": "one eager step after the timed region: HIP events on the launch stream around every GEMM with M >= 4096 "
                                                "(2 M N (K + K2) FLOP each, the LoRA K-segment and the LoRA-down products x A^T included; the text-side GEMMs on the second stream are not timed)",
                                    "share_of_step_time": g_ms / (dt / args.steps * 1e3)}
        if exchange:
            out["config"]["exchange"] = exchange
        if group_info:
            out["config"]["process_group"] = group_info
        # HBM traffic of the dominant kernel: PMC passes cannot run inside the timed run (they serialise kernels), so
        # the per-launch figure is the one measured by tools/pmc_kernel.sh on THIS command and committed under profiles/
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
            if tr and world == 1 and not args.fp8_attn:
                out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, %s)" % tr["source"]
                out["roofline"]["algorithmic_hbm_bytes_per_launch"] = tr["algorithmic_bytes_per_launch"]
                # frac = matrix-pipe busy x effective clock / 2.4 GHz: both from the counter passes of tools/attn_pmc_arms.sh on this kernel (committed; a PMC pass serialises kernels and
                # cannot run inside the timed region)
                if "attn_clock_ghz" in tr:
                    out["config"]["attn_clock_ghz"] = tr["attn_clock_ghz"]
                    out["config"]["attn_mfma_busy"] = tr["attn_mfma_busy"]
                    out["config"]["attn_busy_x_clock_over_2p4"] = tr["attn_mfma_busy"] * tr["attn_clock_ghz"] / 2.4
                if "gemm" in tr and not args.fp8:      # the same two counters for the largest bf16 linear of the step (tools/gemm_one.py under tools/pmc_kernel.sh)
                    out["config"]["gemm_clock_ghz"] = tr["gemm"]["gemm_clock_ghz"]
                    out["config"]["gemm_mfma_busy"] = tr["gemm"]["gemm_mfma_busy"]
                    out["config"]["gemm_busy_x_clock_over_2p4"] = tr["gemm"]["gemm_mfma_busy"] * tr["gemm"]["gemm_clock_ghz"] / 2.4
                    if "roofline_gemm" in out:
                        out["roofline_gemm"]["traffic"] = tr["gemm"]["traffic_bytes"]
                        out["roofline_gemm"]["traffic_note"] = "bytes of ONE launch of the largest linear (M 50688, N 21504, K 3072; FETCH_SIZE x2 + WRITE_SIZE) against %d algorithmic: the re-reads are the 8 XCDs' separate L2s each streaming the operand panels of their 32 tiles (miss rate = the tile order's own at 1 ... 65 rounds, profiles/r06_gemm_l2_hit_vs_rounds.log)" % tr["gemm"]["algorithmic_bytes"]
        except OSError:
            pass
        if world == 1:
            # metric (ii) of SURVEY 8d, geometry part: render -> UV back-projection of 6 x 1024^2 views into a 2048^2
            # atlas of a 50k-face mesh (outside the timed region; a few hundred ms)
            try:
                from unitex_amd.texturetools.benchmarks import time_backprojection
                bp = time_backprojection(50000, 1024, 2048, iters=2, warmup=1, device=dev)
            except Exception as e:  # noqa: BLE001
                bp = {"error": repr(e)}
        if bp is not None and "error" not in bp:
            out["config"]["backprojection"] = {"total_ms": bp["total_ms"], "faces": bp["faces"], "atlas_px": 2048, "view_px": 1024,
                                               "stages_ms": bp["stages_ms"], "view_shard": "views split over %d rank(s)" % world}
            out["config"]["sec_per_mesh_texture"] = 56.0 * dt / args.steps + bp["total_ms"] * 1e-3
            # fused-floor traffic of the back-projection = atlas raster read (16 B/texel) + the six view images read once
            # (6 x 1024^2 x 16 B) + the uint8 atlas written = 0.15 GB, against the measured wall time of the whole stage chain
            fused_floor = 2048 * 2048 * 16.0 + 6 * 1024 * 1024 * 16.0 + 2048 * 2048 * 3.0
            rays = 6.0 * bp["texels"] * bp["covered_frac"]
            out["roofline_backprojection"] = {
                "bound": "hbm", "achieved": fused_floor / (bp["total_ms"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": fused_floor / (bp["total_ms"] * 1e-3) / 8e12, "algorithmic_bytes": fused_floor,
                "ms": bp["total_ms"], "dominant_stage": max(bp["stages_ms"], key=bp["stages_ms"].get),
                "rays": rays, "rays_per_s": rays / (bp["stages_ms"].get("backproject", bp["total_ms"]) * 1e-3),
                "nodes_visited_per_ray": bp.get("nodes_per_ray"),
                "note": "latency-bound LBVH walk, 0.02 % of a mesh's wall time; stage GB/s in stages_gbps",
                "stages_gbps": bp.get("stages_gbps")}
            if world > 1 and "all_gather" in bp["stages_ms"]:
                ag_bytes = 13.0 * 2048 * 2048 * ((6 + world - 1) // world) * (world - 1)
                out["config"]["backprojection"]["all_gather_ms"] = bp["stages_ms"]["all_gather"]
                out["config"]["backprojection"]["all_gather_bytes_received_per_rank"] = ag_bytes
        elif bp is not None:
            out["config"]["backprojection"] = bp
        if world == 1:
            # VAE part of a job (HIP AutoencoderKL): pass 1 encodes the control strip + the 512^2 reference image and
            # decodes the strip, pass 2 (delight) encodes the control strip and decodes once more
            try:
                from unitex_amd.flux.vae_hip import AutoencoderKL
                vae = AutoencoderKL.synthetic(seed=0, device=dev)
                img = (torch.rand(1, 3, h_px, w_px, device=dev) * 2 - 1).to(torch.bfloat16)
                ref_img = (torch.rand(1, 3, dual_px, dual_px, device=dev) * 2 - 1).to(torch.bfloat16)
                zz = torch.randn(1, 16, h_px // 8, w_px // 8, device=dev).to(torch.bfloat16)
                vae.encode(ref_img); vae.encode(img); vae.decode(zz)
                torch.cuda.synchronize()
                tv = time.perf_counter()
                vae.encode(img); vae.encode(ref_img); vae.decode(zz); vae.encode(img); vae.decode(zz)
                torch.cuda.synchronize()
                vae_s = time.perf_counter() - tv
                out["config"]["vae_sec_per_mesh"] = vae_s
                if "sec_per_mesh_texture" in out["config"]:
                    out["config"]["sec_per_mesh_texture"] += vae_s
                del vae, img, zz
            except Exception as e:  # noqa: BLE001
                out["config"]["vae_sec_per_mesh"] = "error: %r" % (e,)
        if world == 1 and not args.no_cpu_baseline:
            ncpu, phys = host_cores()
            try:
                # one thread per physical core, capped at 64: on the 128-core hosts of the GPU boxes torch's fp32 GEMMs measured 0.14
                # TFLOP/s with 128 threads against 0.36 with 64 (profiles/r02_bench_strip1024x6_v0.json.log vs BENCH_r01) -- the baseline
                # is the faster setting; the host's physical core count is reported next to it
                threads = max(1, min(ncpu, phys or ncpu, 64))
                out["cpu_baseline"] = cpu_baseline(S, threads, full_depth=args.cpu_full_step)
                out["cpu_baseline"]["host"] = {"usable_threads": ncpu, "physical_cores_lscpu": phys}
                try:
                    out["cpu_baseline"]["backprojection"] = cpu_baseline_backprojection()
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"]["backprojection"] = {"error": repr(e)}
                if args.cpu_config1:
                    out["cpu_baseline"]["config1_complete"] = cpu_config1(threads)
            except Exception as e:  # noqa: BLE001 -- the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": ncpu, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        # The extras below run in-process behind every measurement of the line.  They are product kernels on product shapes, but a device fault in one of them would take the ONE
        # stdout line with it: the line as it stands goes to STDERR first -- a safety copy in the logs, stdout keeps its single JSON line.
        sys.stderr.write("bench.py core line (safety copy before the reporting extras): " + json.dumps(out) + "\n")
        sys.stderr.flush()
        if world == 1 and not args.sp_self_test and not args.fp8_attn and os.environ.get("UTX_BENCH_EXPERIMENTS", "1") != "0":
            try:
                ab = attention_loop_ab(dev, S_exec, HEADS)
                out["config"]["experiments"] = {"attention_q64_vs_fast8x32": ab}
                out["config"]["experiments_summary"] = "attn 4x64 generated stream (default) vs 8x32 fast loop @%d tok: bit_id=%d %.3f vs %.3f ms = %.3fx (%.0f vs %.0f TF/s)" % (
                    ab["tokens"], int(ab["bit_identical"]), ab["q64_ms"], ab["fast8x32_ms"], ab["fast8x32_ms"] / ab["q64_ms"], ab["q64_tflops"], ab["fast8x32_tflops"])
            except Exception as e:  # noqa: BLE001 -- a reporting extra
                out["config"]["experiments_summary"] = "failed: %r" % (e,)
        if world == 1 and not args.sp_self_test and args.workload != "ref512x6" and os.environ.get("UTX_BENCH_REF_POINT", "1") != "0":
            # the reference's own operating point (512^2 x 6 views, S = 13 824) on the same model, a few steps behind everything else: a flat scalar the driver's record keeps
            try:
                out["config"]["ref512x6_ms_per_step"] = ref_point_ms(model, ops, sched, calculate_shift, shape, dev, prune)
            except Exception as e:  # noqa: BLE001 -- a reporting extra
                out["config"]["ref512x6_ms_per_step"] = "error: %r" % (e,)
        # flat scalars (the driver's record drops nested objects of `config`)
        out["config"]["attn_tflops"] = out["roofline"]["achieved"]
        out["config"]["attn_frac"] = out["roofline"]["frac"]
        if "roofline_gemm" in out:
            out["config"]["gemm_frac"] = out["roofline_gemm"]["frac"]
            out["config"]["gemm_tflops"] = out["roofline_gemm"]["achieved"]
        if bp is not None and "error" not in bp:
            out["config"]["backprojection_total_ms"] = bp["total_ms"]
            out["config"]["backprojection_kernel_sum_ms"] = bp.get("kernel_sum_ms")
        if watchdog is not None:
            watchdog.cancel()      # from here on the full line is the one that is printed
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1 or args.sp_self_test:
        import torch.distributed as dist
        if watchdog is not None:
            watchdog.cancel()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
