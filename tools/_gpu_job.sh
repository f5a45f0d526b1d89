cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export UTX_DIST_BACKEND=gloo
timeout 560 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 8 --steps 1 --warmup 1 > gpurun_out/r03_bench_strip1024x6_8ranks_1gpu.json.log 2>&1; echo "rc=$?"
grep '^{' gpurun_out/r03_bench_strip1024x6_8ranks_1gpu.json.log | cut -c1-1500
tail -5 gpurun_out/r03_bench_strip1024x6_8ranks_1gpu.json.log | cut -c1-300
rocm-smi --showmemuse 2>/dev/null | tail -5
