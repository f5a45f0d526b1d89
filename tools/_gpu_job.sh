cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in view1024 view2048; do
  python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'], c['tokens'], c.get('tokens_computed'), d['ms_per_step'], d['value'], d['roofline']['achieved'], c.get('last_block_pruning') is not None)"
done > gpurun_out/r02_bench_other_workloads.log 2>&1
cat gpurun_out/r02_bench_other_workloads.log
