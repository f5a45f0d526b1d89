cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_geometry_gpu.py -x -q -k "chart_unwrap" 2>&1 | tail -15
