#!/bin/bash
# PMC profile of one kernel (separate passes; --kernel-trace only, as gpurun requires).
#   bash tools/pmc_kernel.sh <outdir> <kernel-name-substring> <command ...>
OUT=$1; KN=$2; shift 2
R=$PWD; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PASSES=${PASSES:-"sq1 sq2 tcc1 tcc2"}
run() { case " $PASSES " in *" $1 "*) ;; *) return;; esac; timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/$OUT/$1 -o run -- "${@:3}" > $R/$OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES" "$@"
run tcc1 "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
run tcc2 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$@"
cd $R
python - <<PY
import csv, glob, collections
for d in ("sq1","sq2","tcc1","tcc2"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if "$KN" in r.get("Kernel_Name",""):
                k = r["Counter_Name"]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        for k,(s,n) in sorted(agg.items()):
            print("%-28s per-dispatch avg %.6g  (n=%d)" % (k, s/max(n,1), n))
PY
