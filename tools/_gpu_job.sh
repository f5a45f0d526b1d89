cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.log 2>&1; tail -2 gpurun_out/r02_smoke.log
timeout 600 python tools/run_full_pipeline.py --reps 2 > gpurun_out/r02_full_pipeline_e2e.log 2>&1
timeout 600 python tools/run_full_pipeline.py --reps 1 --no-uv >> gpurun_out/r02_full_pipeline_e2e.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_full_pipeline_e2e.log | tail -14
