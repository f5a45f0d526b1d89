#!/usr/bin/env python
"""Regenerate the attention entry of profiles/pmc_traffic.json from the counter passes of tools/pmc_kernel.sh on tools/attn_one.py (run on the GPU box):
    PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/pmc_attn attn_fwd_q64 python $PWD/tools/attn_one.py 50240 > gpurun_out/pmc_attn.log
    python tools/pmc_traffic_update.py gpurun_out/pmc_attn gpurun_out/pmc_attn.log          (before the pass directories are removed)
A call = the full-rounds dispatch + the key-split tail dispatch of attn_fwd_q64_kernel: the per-dispatch averages pmc_kernel.sh prints are doubled; kernel time = the sum of
the two dispatches' durations under the tcc1 pass (GRBM_GUI_ACTIVE is collected there).  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, gfx950)."""
import csv, glob, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, log = sys.argv[1], sys.argv[2]
vals = {}
for l in open(log):
    m = re.match(r"(\S+)\s+per-dispatch avg ([0-9.e+]+)\s+\(n=(\d+)\)", l)
    if m:
        vals[m.group(1)] = (float(m.group(2)), int(m.group(3)))
durs = []
for f in glob.glob(os.path.join(out_dir, "tcc1", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "attn_fwd_q64" in r["Kernel_Name"]:
            durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
durs = [d for _, d in sorted(durs)]
n_disp = vals["FETCH_SIZE"][1]
calls = n_disp // 2
assert len(durs) == n_disp and calls >= 1, (len(durs), n_disp)
ms_call = sum(durs) / calls / 1e6
per_call = {k: 2.0 * v for k, (v, _) in vals.items()}
S, H = 50240, 24
flops = 4.0 * S * S * 128 * H
cyc = per_call["GRBM_GUI_ACTIVE"] / 8.0
clk = cyc / (ms_call * 1e6)
busy = per_call["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
p = os.path.join(root, "profiles", "pmc_traffic.json")
d = json.load(open(p))
e = d["strip1024x6"]
e.update({
    "kernel": "attn_fwd_q64_kernel<0,3,0> (round 6's default: 4 x 64 queries per workgroup, one wave per SIMD, generated stream, three tiles per loop trip; main + tail-split dispatch of one attention call; "
              "the PMC averages are per dispatch, n=%d = %d calls x 2 dispatches, doubled here)" % (n_disp, calls),
    "fetch_size_kb_per_launch": per_call["FETCH_SIZE"], "fetch_correction": 2.0, "write_size_kb_per_launch": per_call["WRITE_SIZE"],
    "traffic_bytes_per_launch": (per_call["FETCH_SIZE"] * 2.0 + per_call["WRITE_SIZE"]) * 1024.0,
    "tcc_hit_frac": vals["TCC_HIT_sum"][0] / (vals["TCC_HIT_sum"][0] + vals["TCC_MISS_sum"][0]),
    "kernel_ms_per_launch_under_pmc": ms_call, "attn_clock_ghz": clk, "attn_mfma_busy": busy, "attn_busy_x_clock_over_2p4": busy * clk / 2.4,
    "tflops_under_pmc": flops / ms_call / 1e9,
    "sq": {k: per_call[k] for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE") if k in per_call},
    "source": "profiles/%s (round 6, tools/pmc_kernel.sh on tools/attn_one.py: the bench's attention launch, text dedup 64+50176 tokens, key weight 2^3; tools/pmc_traffic_update.py); "
              "clock = GRBM_GUI_ACTIVE / 8 / kernel time, busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024)" % os.path.basename(log).replace("pmc_attn.log", "r06_pmc_attn_strip1024x6_v1.log"),
})
json.dump(d, open(p, "w"), indent=1)
print("per call: %.3f ms under the counters, clock %.3f GHz, matrix pipe %.1f %% busy -> busy x clock / 2.4 = %.4f (%.0f TF/s under the counters); traffic %.3f GB against %.3f algorithmic" %
      (ms_call, clk, 100 * busy, busy * clk / 2.4, flops / ms_call / 1e9, e["traffic_bytes_per_launch"] / 1e9, e["algorithmic_bytes_per_launch"] / 1e9))
