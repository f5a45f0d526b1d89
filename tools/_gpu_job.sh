# round 6, call 2: the generated 4 x 64 stream -- bit identity sweep against the 8 x 32 fast loop + timing; then the new direct fixture tests and the repaired strict test
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/attn_q64_check.py > gpurun_out/r06_attn_q64_check_v0.log 2>&1; echo "check rc=$?"
tail -40 gpurun_out/r06_attn_q64_check_v0.log
timeout 900 python -m pytest tests/test_fixtures_direct_gpu.py "tests/test_dit_ops_gpu.py::test_attention_q64_kernel_and_repair_pass" tests/test_fullsize_gpu.py::test_full_width_dit_blocks_at_config1_shape_match_oracle -x -q -s -m gpu > gpurun_out/r06_fixtures_direct.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r06_fixtures_direct.log
