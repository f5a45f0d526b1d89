# round 6, call 7: PMC passes of the attention call with the 4 x 64 kernel (traffic, clock, busy)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/pmc_q64 attn_fwd_q64 python $R/tools/attn_one.py 50240 > gpurun_out/r06_pmc_attn_strip1024x6.log 2>&1
cat gpurun_out/r06_pmc_attn_strip1024x6.log
head -3 gpurun_out/pmc_q64/sq1.log
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_q64/tcc1/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] or "merge" in r["Kernel_Name"]]
    for r in rows[-8:]:
        print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
rm -rf gpurun_out/pmc_q64/*/*.db
