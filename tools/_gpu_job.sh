cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
timeout -s KILL 100 python bench.py --fp8 --fp8-attn --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_strip1024x6_fp8_attn_v2.json.log 2> gpurun_out/r04_bench_fp8_attn_v2.err; echo "rc=$?"; tail -c 1500 gpurun_out/r04_bench_strip1024x6_fp8_attn_v2.json.log | cut -c1-1500; tail -3 gpurun_out/r04_bench_fp8_attn_v2.err
