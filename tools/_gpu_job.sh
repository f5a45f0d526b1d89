# round 6, call 21: sequence-parallel key de-duplication (S_q > S_kv launches through the 4 x 64 stream): new tests, the SP tests, A/B at the per-rank shapes
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dit_ops_gpu.py -q -m gpu -k "more_queries or relayout or sequence_parallel or head_group or block_strided" > gpurun_out/r06_sp_dedup_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06_sp_dedup_tests.log
timeout 900 python tools/attn_sp_dedup_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_attn_sp_dedup_ab.log
