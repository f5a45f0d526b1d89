cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout -s KILL 300 python -m pytest tests/test_attention_peel_gpu.py "tests/test_determinism_stress_gpu.py::test_fast_attention_loop_320_launches_against_the_general_loops_bits" tests/test_variants_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r05_tests_c.log 2>&1
tail -3 gpurun_out/r05_tests_c.log
timeout -s KILL 120 python tools/attn_peel_ab.py > gpurun_out/r05_attn_fast_loop_ab_v2.log 2>&1; grep bf16 gpurun_out/r05_attn_fast_loop_ab_v2.log
