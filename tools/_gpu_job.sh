# round 6, call 1: per-arm PMC table (clock, matrix-pipe busy, LDS cycles) of the fast 8 x 32 loop, the general loop and the 4 x 64 kernel, both operating points
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
for S in 50240 13376; do
  UTX_ONE_S=$S bash tools/attn_pmc_arms.sh gpurun_out/pmc_arms_$S "fast:UTX_ATTN_PEEL=1,UTX_ONE_KB=0" "general:UTX_ATTN_PEEL=0,UTX_ONE_KB=0" "q64:UTX_ATTN_Q64=1,UTX_ONE_KB=0" > gpurun_out/r06_attn_pmc_arms_q64_$S.log 2>&1
  cat gpurun_out/r06_attn_pmc_arms_q64_$S.log
done
UTX_AB_ARMS=default,peel1,q64 python tools/attn_q64_ab.py 2>&1 | tee gpurun_out/r06_attn_q64_ab_v0.log
rm -rf gpurun_out/pmc_arms_*/*/*/*.db 2>/dev/null
du -sh gpurun_out
