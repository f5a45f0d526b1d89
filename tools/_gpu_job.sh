cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_h.log 2>&1
tail -3 gpurun_out/r02_gpu_tests_h.log
for w in strip1024x6 ref512x6; do for gm in 4 0 4 0; do
  UTX_GEMM_GROUP_M=$gm timeout 150 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w group_m=$gm', d['ms_per_step'], d['value'])"
done; done > gpurun_out/r02_bench_group_m_ab.log 2>&1
cat gpurun_out/r02_bench_group_m_ab.log
