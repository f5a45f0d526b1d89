cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sp -o sp -- python $R/bench.py --sp-self-test --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03_rocprof_bench_sp_self_test.log 2>&1; echo "rc=$?"
cd $R
f=$(find gpurun_out/prof_sp -name "*kernel_stats.csv" | head -1)
grep -i "unpack\|nccl\|rccl\|AllToAll\|SendRecv\|attn_fwd\|qkv_post\|memcpy\|copy" "$f" | cut -c1-260
cp "$f" gpurun_out/r03_rocprofv3_kernel_stats_sp_self_test.csv
rm -rf gpurun_out/prof_sp
