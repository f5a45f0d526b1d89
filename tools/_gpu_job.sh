cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. hardware ordering rule behind the counted waits
hipcc --offload-arch=gfx950 -O2 tools/vmcnt_order_probe.hip -o /tmp/vmcnt_probe > gpurun_out/r04_vmcnt_probe_build.log 2>&1
timeout 240 /tmp/vmcnt_probe 4000 2048 > gpurun_out/r04_vmcnt_order_probe.log 2>&1; echo "probe rc=$?"
cat gpurun_out/r04_vmcnt_order_probe.log
# 2. reproducer: round-3 library vs this build
timeout 300 python tools/qkf_ragged_repro.py 40 tools/bin/libunitex_hip_r03.so > gpurun_out/r04_qkf_repro_r03lib.log 2>&1; echo "repro old rc=$?"; tail -4 gpurun_out/r04_qkf_repro_r03lib.log | cut -c1-300
timeout 300 python tools/qkf_ragged_repro.py 40 > gpurun_out/r04_qkf_repro_newlib.log 2>&1; echo "repro new rc=$?"; tail -2 gpurun_out/r04_qkf_repro_newlib.log | cut -c1-300
# 3. the strict test + the stress
timeout 600 python -m pytest tests/test_determinism_stress_gpu.py tests/test_fullsize_gpu.py tests/test_dit_ops_gpu.py -m gpu -x -q -k "stress or bit_identical or cold or gated or full_width_dit_blocks or fused" --durations=8 > gpurun_out/r04_stress_a.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r04_stress_a.log | cut -c1-300
# 4. step time after the wait-macro change
timeout 400 python bench.py --steps 6 --warmup 2 > gpurun_out/r04_bench_a.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r04_bench_a.log | cut -c1-600
