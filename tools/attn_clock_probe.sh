#!/bin/bash
# Effective shader clock of the attention kernels under different instruction mixes / operand data:
#   GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, per dispatch, from rocprofv3 (counter pass + kernel trace).
#   bash tools/attn_clock_probe.sh <outdir>
OUT=$1; R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$name -o run -- python $R/tools/perf_ops.py --attn-only > $R/$OUT/$name.log 2>&1
}
run default_random UTX_ATTN_Q64=0
run default_zero UTX_ATTN_Q64=0 UTX_PERF_ZERO=1
run q64_random UTX_ATTN_Q64=1
run q64_zero UTX_ATTN_Q64=1 UTX_PERF_ZERO=1
run q64_mfma_only UTX_ATTN_Q64=1 UTX_ATTN_VAR=30
run q64_no_softmax UTX_ATTN_Q64=1 UTX_ATTN_VAR=16
cd $R
python - <<PY
import csv, glob, collections
for name in ("default_random","default_zero","q64_random","q64_zero","q64_mfma_only","q64_no_softmax"):
    dur = {}
    for f in glob.glob("$OUT/%s/*kernel_trace.csv" % name):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % name):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "attn_fwd" in r["Kernel_Name"] and r["Dispatch_Id"] in dur:
                ns = dur[r["Dispatch_Id"]][1]
                if ns > 15e6:      # the S = 50 688 launches
                    rows.append((float(r["Counter_Value"]) / 8.0 / ns, ns / 1e6))
    if rows:
        clk = sum(x for x, _ in rows) / len(rows); ms = sum(y for _, y in rows) / len(rows)
        print("%-16s S=50688: %6.2f ms/launch (profiled)  effective clock %.2f GHz  (%d launches)" % (name, ms, clk, len(rows)))
PY
