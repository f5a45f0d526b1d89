# round 6, call 38: smoke + a fast cross-section of the GPU suite on the final tree (after the streaming-GEMM experiment left the product)
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_dit_ops_gpu.py tests/test_attention_q64_gpu.py tests/test_fixtures_direct_gpu.py -q -m gpu 2>&1 | tail -3
