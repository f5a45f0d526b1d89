cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 110 tools/bin/qkv_pk_probe 5000 50176 > gpurun_out/r04_qkv_post_pk_probe.log 2>&1; echo "rc=$?"; cat gpurun_out/r04_qkv_post_pk_probe.log
