cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_geometry_gpu.py tests/test_variants_gpu.py tests/test_multigpu_gpu.py -x -q -m gpu > gpurun_out/r02_gpu_tests_geom_packed.log 2>&1
tail -5 gpurun_out/r02_gpu_tests_geom_packed.log
timeout 300 python tools/bp_ab.py > gpurun_out/r02_backproject_packed_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_backproject_packed_ab.log
