cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r02_bench_fuse_qk_ab.log
for rep in 1 2; do
  for t in 0 1; do
    echo "== UTX_FUSE_QK=$t strip1024x6" >> gpurun_out/r02_bench_fuse_qk_ab.log
    UTX_FUSE_QK=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/r02_bench_fuse_qk_ab.log
  done
done
for t in 0 1 0 1; do
  echo "== UTX_FUSE_QK=$t ref512x6" >> gpurun_out/r02_bench_fuse_qk_ab.log
  UTX_FUSE_QK=$t python bench.py --workload ref512x6 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/r02_bench_fuse_qk_ab.log
done
cat gpurun_out/r02_bench_fuse_qk_ab.log
