cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -x -q -k "multiplicity or dedup or sequence_parallel" 2>&1 | tail -8
