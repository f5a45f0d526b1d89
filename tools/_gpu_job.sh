cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 python bench.py --workload ref512x6 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['gemm_launches_per_step'])"
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
