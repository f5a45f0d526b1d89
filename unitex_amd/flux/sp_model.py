"""Arithmetic model of ONE denoise step under head-parallel sequence parallelism (flux/ulysses.py) on an xGMI node -- what the first
multi-GPU SCALE run is to be compared with.  Pure host arithmetic, no device.

Inputs that are MEASURED on one MI355X (bench.py; round 3: profiles/r03_bench_strip1024x6_v1.json.log, S = 50 688 tokens, 57 layers; the DEFAULTS are round 6's, MEASURED_1GPU below:
step 1777.0 ms = 57 x 21.16 + 524.7 + the rest; `python -m unitex_amd.flux.sp_model` prints the predicted table):
    step 1998.9 ms = attention 57 x 25.47 ms (utx_attn_fwd_bf16) + large-M GEMMs 515.4 ms (roofline_gemm.sum_ms_per_step)
                     + 31.7 ms of everything else (LayerNorm-modulation, q/k post-processing, text-side GEMMs, GEMVs, launch gaps)
Inputs that are ASSUMED (the fabric has never been measured by this repo -- no multi-GPU box in reach):
    xGMI: 7 links per GPU, one per peer (fully connected 8-GPU node), 153 GB/s per link BIDIRECTIONAL = 76.5 GB/s per direction
    (MI355X_MICROARCH.md); an all-to-all sustains ~65 % of that per link and direction -> 50 GB/s.  A rank talks to P - 1 peers over P - 1 links:
    the per-direction aggregate is (P - 1) x 50 GB/s -- 50 at P = 2, 150 at P = 4, 350 at P = 8 (NOT the "1 TB/s per GPU" of DESIGN's round-2 text,
    which was the bidirectional nameplate of all seven links).

Per layer and rank (S_loc = S / P local tokens, D = 3072, bf16):
    exchange 1 (Q, K, V -> heads):  3 S_loc (D / P) 2 B to EVERY peer, each over its own link      t_in  = 3 S_loc D 2 / (P link)
    exchange 2 (O -> tokens)     :    S_loc (D / P) 2 B to every peer                              t_out = t_in / 3
    compute = (attention + GEMMs + elementwise) / P, times the round-quantisation of the smaller launches (utx_attn_plan, 256 x 256 tiles)
    exposed communication with G pipelined head groups (ulysses.py):
        first-in / last-out:   (t_in + t_out) / G      (single blocks: the MLP half of the fused projection runs beside exchange 1 as well)
        + what the fabric cannot finish behind the other G - 1 groups' attention:  max(0, (t_in + t_out) (G - 1) / G - t_attn (G - 1) / G)
    not sharded: the AdaLN-modulation GEMV (6.5 GB of weights streamed per step) + embedders, ~1.5 ms.
    only under sequence parallelism (measured on ONE GPU with `bench.py --sp-self-test`: the plan on a 1-rank NCCL group with every all-to-all issued to itself,
    profiles/r03_bench_sp_self_test.json.log + r03_rocprofv3_kernel_stats_sp_self_test.csv): the relayout kernels behind the exchanges -- sp_unpack_qkv 4 x 89 us
    + sp_unpack_o 4 x 32 us per layer at P = 1 = 27.6 ms per step, 1 / P of it per rank, on the compute stream -- and the attention launches running 1.3 % slower
    while RCCL's kernels hold CUs beside them (6.589 x 4 vs 26.01 ms per layer).  The whole machinery at P = 1, fabric excluded: 2106 vs 2035 ms per step (+3.5 %).
"""
import math

D, HEADS, LAYERS, N_DOUBLE, N_SINGLE = 3072, 24, 57, 19, 38

MEASURED_1GPU_R03 = dict(step_ms=1998.9, attn_ms_per_layer=25.47, gemm_ms_per_step=515.4, replicated_ms=1.5, sp_unpack_ms_per_step=27.6, attn_corun_factor=1.013)      # the round-3 inputs the docstring quotes
# ROUND 5 (profiles/r05_bench_strip1024x6_v2.json.log, r05_attn_kbp_ab_v2.log): step 1909.0 ms = attention 57 x 23.58 ms (the fast loop, two tiles per trip) + GEMMs 497 ms + the rest.
# Under sequence parallelism the launches carry periodic key multiplicity and run the fast loop's KBP instance (the same two-tile loop over the runs of ordinary tiles, key-multiplicity
# tiles through a copy of their own): 1335 / 1337 TF/s at the 4- / 8-rank per-rank shapes against 1330 for the plain instance on the same box -> no extra factor.
MEASURED_1GPU_R05 = dict(step_ms=1909.0, attn_ms_per_layer=23.58, gemm_ms_per_step=497.0, replicated_ms=1.5, sp_unpack_ms_per_step=27.6, attn_corun_factor=1.013)
# ROUND 6 (profiles/r06_bench_strip1024x6_v1.json.log): step 1777.0 ms = attention 57 x 21.16 ms (the 4 x 64 generated stream) + GEMMs 524.7 ms + the rest.  The sequence-parallel
# launches run the SAME kernel since the receive-side unpack keeps the ranks' identical text rows once among the keys (ulysses.py kv_text_rows: S_q = P S_loc queries over the
# single-GPU key sequence, key multiplicity on tile 0 only): 1406 / 1377 / 1413 TF/s for the head-group launches of 2 / 4 / 8 ranks against 1278 / 1253 / 1275 for the
# periodic-multiplicity form on the 8 x 32 KBP loop, same process (profiles/r06_attn_sp_dedup_ab.log) -- queries grow with P, keys do not (kv_dedup below).
MEASURED_1GPU = MEASURED_1GPU_R06 = dict(step_ms=1777.0, attn_ms_per_layer=21.16, gemm_ms_per_step=524.7, replicated_ms=1.5, sp_unpack_ms_per_step=27.6, attn_corun_factor=1.013)
# the large-M GEMMs lose efficiency as M = S / P shrinks (fewer rounds of 256 x 256 tiles per launch, a larger share of fill / epilogue): bf16 TF/s at
# M = 13 824 vs 50 688 on the FLUX shapes, same process (profiles/r03_perf_fp8_v0.log, bf16 column): 1224 / 1360, 1326 / 1292, 1241 / 1337 -> ~0.93 at a
# quarter of the rows; 0.97 at half and 0.85 at an eighth are interpolated / extrapolated, not measured
GEMM_EFFICIENCY = {1: 1.0, 2: 0.97, 4: 0.93, 8: 0.85}
XGMI = dict(link_gbps_per_direction=76.5, all_to_all_efficiency=0.65)


def exchange_bytes_per_peer(S_loc, P):
    """bytes one rank sends to ONE peer in exchange 1 of a layer (exchange 2 is a third of it)."""
    return 3 * S_loc * (D // P) * 2


def attention_round_factor(P, S, n_cus=256, groups=1, plan=None):
    """time of a rank's attention launches relative to the ideal 1 / P share of the single-GPU launch: rounds of 256-query workgroups over the
    CUs, the partly filled last round cut along the keys (utx_attn_plan: the library's own arithmetic)."""
    def rounds(H):
        if plan is not None:
            nwg, nfull, ns, _ = plan(H, S, S, n_cus)
        else:
            nwg = H * ((S + 255) // 256); nfull = (nwg // n_cus) * n_cus; ns = 1
        r = nwg - nfull
        tail = 0.0 if r == 0 else (math.ceil(r * ns / n_cus) * (1.0 / ns + 0.02) if ns > 1 else 1.0)
        return nfull / n_cus + tail
    Hg = HEADS // P // groups
    return groups * rounds(Hg) / (rounds(HEADS) / P)


def predict(P, S=50240, groups=None, measured=None, xgmi=None, plan=None, n_cus=256, kv_dedup=True):
    """predicted step time and speed-up on P ranks.  S = executed tokens of the single-GPU run (text de-duplication: 64 + 50 176); under P ranks every
    rank carries its own 64 text rows.  groups: head groups per rank (None: ulysses.pick_head_groups).  kv_dedup (the default since round 6): the ranks' text rows are kept once
    among the keys -- the query count grows with P, the key count does not (False: round 2's periodic form, both grow).  Returns a dict with the breakdown."""
    from .ulysses import pick_head_groups
    m = dict(MEASURED_1GPU, **(measured or {}))
    x = dict(XGMI, **(xgmi or {}))
    if P == 1:
        return {"P": 1, "step_ms": m["step_ms"], "speedup": 1.0, "exposed_comm_ms": 0.0, "groups": 1}
    S_img = S - 64
    S_loc = (64 * P + S_img) // P
    S_all = S_loc * P
    G = groups or pick_head_groups(HEADS // P, S_all, n_cus)
    link = x["link_gbps_per_direction"] * x["all_to_all_efficiency"] * 1e9
    t_in = exchange_bytes_per_peer(S_loc, P) / link * 1e3            # ms; every peer pair on its own link, all at once
    t_out = t_in / 3.0
    grow = S_all / float(S)                                          # the extra text rows
    t_attn = m["attn_ms_per_layer"] / P * grow * (1.0 if kv_dedup else grow) * attention_round_factor(P, S_all, n_cus, G, plan) * m["attn_corun_factor"]
    other_ms = m["step_ms"] - LAYERS * m["attn_ms_per_layer"] - m["gemm_ms_per_step"] - m["replicated_ms"]
    t_gemm_layer = m["gemm_ms_per_step"] / LAYERS / P * grow / GEMM_EFFICIENCY.get(P, 0.85)
    t_other_layer = (other_ms + m["sp_unpack_ms_per_step"]) / LAYERS / P * grow      # + the relayout kernels of the two exchanges
    t_mlp_half = t_gemm_layer * (12288.0 / (9216 + 12288 + 15360))  # the MLP half of a single block's fused projection, by FLOPs
    fabric = t_in + t_out
    behind = max(0.0, fabric * (G - 1) / G - t_attn * (G - 1) / G)
    exp_double = fabric / G + behind
    exp_single = max(0.0, t_in / G - t_mlp_half) + t_out / G + behind
    exposed = N_DOUBLE * exp_double + N_SINGLE * exp_single
    step = LAYERS * (t_attn + t_gemm_layer + t_other_layer) + exposed + m["replicated_ms"]
    return {"P": P, "groups": G, "S_loc": S_loc, "step_ms": step, "speedup": m["step_ms"] / step,
            "attention_ms_per_layer": t_attn, "gemm_ms_per_layer": t_gemm_layer, "other_ms_per_layer": t_other_layer,
            "exchange_in_ms": t_in, "exchange_out_ms": t_out, "bytes_per_peer_in": exchange_bytes_per_peer(S_loc, P),
            "bytes_per_rank_per_layer_on_fabric": (4.0 / 3.0) * exchange_bytes_per_peer(S_loc, P) * (P - 1),
            "exposed_comm_ms": exposed, "exposed_comm_frac": exposed / step,
            "exposed_comm_unpipelined_ms": N_DOUBLE * fabric + N_SINGLE * (max(0.0, t_in - t_mlp_half) + t_out)}


def table(Ps=(1, 2, 4, 8), **kw):
    return [predict(P, **kw) for P in Ps]


if __name__ == "__main__":
    for r in table():
        print(r)
