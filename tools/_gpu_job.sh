cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python tools/run_full_pipeline.py > gpurun_out/r03_full_pipeline_e2e_final.log 2>&1; echo "rc=$?"
grep -n "sec_per_mesh\|>>> infer_mv\|peak mem" gpurun_out/r03_full_pipeline_e2e_final.log | tail -8
