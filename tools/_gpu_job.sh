cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/run_full_pipeline.py --view 1024 --reps 1 > gpurun_out/r02_full_pipeline_e2e_1024.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_full_pipeline_e2e_1024.log | tail -12
