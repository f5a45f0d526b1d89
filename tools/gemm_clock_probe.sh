#!/bin/bash
# bash tools/gemm_clock_probe.sh <outdir> [M N K]
OUT=$1; shift; R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PASSLIST=${PASSLIST:-"a b c"}
for pass in "a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "c SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "d FETCH_SIZE GRBM_GUI_ACTIVE" "e WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  case " $PASSLIST " in *" ${pass%% *} "*) ;; *) continue;; esac
  set -- $pass; name=$1; shift
  timeout ${PASS_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc $* --output-format csv -d $R/$OUT/$name -o run -- python $R/tools/gemm_clock_probe.py $PROBE_SHAPE > $R/$OUT/$name.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for d in "abcde":
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*kernel_trace.csv" % d):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:72]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if not ("gemm" in k.lower() or "Cijk" in k): continue
        us = sorted(dur[k])[len(dur[k]) // 2] if dur.get(k) else float("nan")
        print("%-72s n=%d  median %.1f us" % (k, len(dur.get(k, [])), us))
        for cn, v in sorted(c.items()):
            m = sorted(v)[len(v) // 2]
            extra = "  -> %.3f GHz" % (m / us * 1e-3) if cn == "GRBM_GUI_ACTIVE" else ""
            print("    %-32s %.6g%s" % (cn, m, extra))
PY
