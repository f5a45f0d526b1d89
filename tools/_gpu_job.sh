cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/pk_fp32_mfma_probe.hip -o /tmp/pk_probe > gpurun_out/r04_pk_probe_build.log 2>&1
timeout -s KILL 60 /tmp/pk_probe 100000 > gpurun_out/r04_pk_fp32_mfma_probe2.log 2>&1; echo "rc=$?"; cat gpurun_out/r04_pk_fp32_mfma_probe2.log
