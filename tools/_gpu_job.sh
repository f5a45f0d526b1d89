cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_strip1024x6_v2.json.log 2>&1
python bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_ref512x6_v1.json.log 2>&1
UTX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02_bench_strip1024x6_2ranks_1gpu.json.log 2>&1
python bench.py --fp8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_strip1024x6_fp8.json.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_rocprofv3_bench_strip1024x6_v2.log 2>&1)
cp gpurun_out/prof_r02/*kernel_stats.csv gpurun_out/r02_rocprofv3_kernel_stats_strip1024x6_v2.csv 2>/dev/null
rm -rf gpurun_out/prof_r02
for f in gpurun_out/r02_bench_strip1024x6_v2.json.log gpurun_out/r02_bench_ref512x6_v1.json.log gpurun_out/r02_bench_strip1024x6_2ranks_1gpu.json.log gpurun_out/r02_bench_strip1024x6_fp8.json.log; do echo "== $f"; grep '^{' $f | cut -c1-600; done
head -8 gpurun_out/r02_rocprofv3_kernel_stats_strip1024x6_v2.csv | cut -c1-200
