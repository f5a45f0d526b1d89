cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out
PASSES="tcc1 tcc2 sq1" bash tools/pmc_kernel.sh gpurun_out/pmc_r02 attn_fwd_glds python $R/tools/attn_one.py > gpurun_out/r02_pmc_attn_strip1024x6.log 2>&1
cat gpurun_out/r02_pmc_attn_strip1024x6.log
tail -3 gpurun_out/pmc_r02/tcc1.log
rm -rf gpurun_out/pmc_r02
