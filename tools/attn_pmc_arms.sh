#!/bin/bash
# SQ cycle counters + effective clock of the attention kernel's variants / ablation arms (one rocprofv3 counter pass per arm; --kernel-trace only):
# wall time alone cannot separate "fewer cycles" from "a higher clock" on this power-limited chip (an arm whose MFMA operands stop changing clocks higher).
#   bash tools/attn_pmc_arms.sh <outdir> "name:ENV=V,ENV2=V ..." ...
OUT=$1; shift
R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for arm in "$@"; do
  name=${arm%%:*}; envs=${arm#*:}; envs=${envs//,/ }
  env $envs timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$name -o run -- python $R/tools/attn_one.py ${UTX_ONE_S:-50240} > $R/$OUT/$name.log 2>&1
done
cd $R
python - "$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("%-22s %8s %7s %10s %10s %10s %10s %10s %10s %10s %10s" % ("arm", "ms", "GHz", "cyc/CU", "wave_cyc", "wait_any", "wait_inst", "active", "mfma_busy%", "valu_act", "lds_idx"))
for arm in sys.argv[2:]:
    name = arm.split(":")[0]
    dur = {}
    for f in glob.glob("%s/%s/*kernel_trace.csv" % (out, name)):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # the big dispatch of each call (full rounds): the longest attn_fwd dispatches
    big = sorted((d for d in dur.items() if "attn_fwd" in d[1][0]), key=lambda x: -x[1][1])[:3]
    ids = {d[0] for d in big}
    agg = collections.defaultdict(float)
    for f in glob.glob("%s/%s/*counter_collection.csv" % (out, name)):
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] in ids:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids)
    if not ids or not agg:
        print("%-22s no data" % name); continue
    ms = sum(d[1][1] for d in big) / len(big) / 1e6
    ghz = agg["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e6)
    cyc = ms * 1e6 * ghz
    print("%-22s %8.3f %7.3f %10.4g %10.4g %10.4g %10.4g %10.4g %9.1f%% %10.4g %10.4g" % (name, ms, ghz, cyc, agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_ANY"], agg["SQ_WAIT_INST_ANY"], agg["SQ_ACTIVE_INST_ANY"],
          100.0 * agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), agg["SQ_ACTIVE_INST_VALU"], agg["SQ_LDS_IDX_ACTIVE"]))
PY
