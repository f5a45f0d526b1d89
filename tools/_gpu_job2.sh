cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/pmc_attn_r05 attn_fwd_glds python $R/tools/attn_one.py > gpurun_out/r05_pmc_attn_strip1024x6.log 2>&1
cat gpurun_out/r05_pmc_attn_strip1024x6.log | grep -v "^$" | tail -20
rm -rf gpurun_out/pmc_attn_r05/*/
