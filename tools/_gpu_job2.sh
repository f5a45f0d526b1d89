cd $GRAFT_REPO_ROOT; exec < /dev/null; mkdir -p gpurun_out; export TMPDIR=/tmp
( echo "== product library (no packed fp32 in the elementwise kernels), two streams on"; UTX_TXT_STREAM=1 timeout -s KILL 400 python tools/two_stream_probe.py 10000 base 2>&1 | grep -v amdgpu ) > gpurun_out/r05_two_stream_soak_product.log 2>&1
cat gpurun_out/r05_two_stream_soak_product.log
( timeout -s KILL 600 python -m pytest tests/test_geometry_gpu.py tests/test_vae_gpu.py tests/test_variants_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r05_tests_d.log 2>&1; tail -3 gpurun_out/r05_tests_d.log
