cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/r03_pmc_attn attn_fwd_glds python $GRAFT_REPO_ROOT/tools/attn_one.py > gpurun_out/r03_pmc_attn_strip1024x6.log 2>&1; echo "pmc attn rc=$?"
cat gpurun_out/r03_pmc_attn_strip1024x6.log
rm -rf gpurun_out/r03_pmc_attn
