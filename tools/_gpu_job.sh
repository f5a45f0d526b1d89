cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 python bench.py --workload ref512x6 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_ref512x6_v6.json 2> gpurun_out/r02_bench_ref512x6_v6.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_ref512x6_v6.json')); print(d['ms_per_step'], d['config']['gemm_launches_per_step'])"
