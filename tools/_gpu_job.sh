cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# round 3, job 12: build() + smoke() in ONE process (library loaded before / after torch), and smoke() alone as the driver calls it
timeout 300 python __graft_entry__.py smoke > gpurun_out/r03_smoke.log 2>&1; echo "build+smoke rc=$?"
tail -n 2 gpurun_out/r03_smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_alone.log 2>&1; echo "smoke alone rc=$?"
tail -n 1 gpurun_out/r03_smoke_alone.log
