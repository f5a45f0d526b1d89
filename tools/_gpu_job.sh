# round 6, call 26: L2 hit rate of the bf16 GEMM against the number of tile rounds (is the 70 % of the full-size launch a lockstep drift of the persistent workgroups?)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
for shape in "2048 8192 3072" "4096 8192 3072" "8192 8192 3072" "16384 8192 3072" "50688 8192 3072" "50688 21504 3072" "50688 3072 15360"; do
  set -- $shape
  echo "== M=$1 N=$2 K=$3 (tiles = $(( ($1/256) * ($2/256) )), rounds = $(python -c "print(round(($1/256)*($2/256)/256.0,2))"))"
  PASSES="tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/pmc_gemm_rounds gemm256_w4 python $GRAFT_REPO_ROOT/tools/gemm_one.py bf16 $1 $2 $3 2>&1 | grep -E "FETCH|TCC|WRITE|GRBM"
  python - <<'PY'
import csv, glob
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob("gpurun_out/pmc_gemm_rounds/tcc1/*kernel_trace.csv") for r in csv.DictReader(open(f)) if "gemm256_w4" in r["Kernel_Name"]]
print("kernel duration: n=%d avg %.4f ms" % (len(d), sum(d) / max(len(d), 1) / 1e6))
PY
  rm -rf gpurun_out/pmc_gemm_rounds
done 2>&1 | tee gpurun_out/r06_gemm_l2_hit_vs_rounds.log
