# round 6, call 31: counter passes of the attention launch on the final stream (three tiles per trip) -> profiles/pmc_traffic.json
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
PASSES="sq1 tcc1 tcc2" bash tools/pmc_kernel.sh gpurun_out/pmc_attn attn_fwd_q64 python $GRAFT_REPO_ROOT/tools/attn_one.py 50240 > gpurun_out/pmc_attn.log 2>&1
cat gpurun_out/pmc_attn.log
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.before.json
python tools/pmc_traffic_update.py gpurun_out/pmc_attn gpurun_out/pmc_attn.log
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_attn/*/
