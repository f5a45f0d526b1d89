cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_fullsize_gpu.py -x -q -k "split_tail or full_width" > gpurun_out/r02_streamk_tests.log 2>&1
tail -5 gpurun_out/r02_streamk_tests.log
for w in strip1024x6 ref512x6; do for sk in 0 1 0 1; do
  UTX_GEMM_STREAMK=$sk timeout 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w streamk=$sk', d['ms_per_step'], d['value'])"
done; done > gpurun_out/r02_bench_streamk_ab.log 2>&1
cat gpurun_out/r02_bench_streamk_ab.log
