cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# round 3, job 3: pipelined sequence-parallel exchanges (2 processes on one GPU), 26-direction chart unwrap, mesh preparation, 2-rank bench control flow
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -x -q -m gpu -k "sequence or head_group or relayout" > gpurun_out/r03_sp_tests_c.log 2>&1; echo "sp tests rc=$?"
tail -n 8 gpurun_out/r03_sp_tests_c.log
timeout 900 python -m pytest tests/test_geometry_gpu.py tests/test_pipeline_gpu.py tests/test_multigpu_gpu.py -x -q -m gpu > gpurun_out/r03_geom_pipe_tests_c.log 2>&1; echo "geom/pipe/multigpu rc=$?"
tail -n 12 gpurun_out/r03_geom_pipe_tests_c.log
UTX_DIST_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_strip1024x6_2ranks_1gpu.json.log 2>&1; echo "bench 2 ranks rc=$?"
tail -n 2 gpurun_out/r03_bench_strip1024x6_2ranks_1gpu.json.log | cut -c 1-1500
