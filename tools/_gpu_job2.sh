cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 200 python tools/bench_backproject.py --faces 50000 > gpurun_out/r05_bp_stages_v0.log 2>&1; grep -v "^{" gpurun_out/r05_bp_stages_v0.log; python - <<'PY'
import json
for l in open('gpurun_out/r05_bp_stages_v0.log'):
    if l.startswith('{'):
        r=json.loads(l); print({k:r[k] for k in ('total_ms','kernel_sum_ms','host_enqueue_ms')})
PY
( timeout -s KILL 400 python -m pytest tests/test_geometry_gpu.py tests/test_c_host_gpu.py -m gpu -q -x 2>&1 | tail -4 ) 
