# round 6, call 36: the one-pass streaming kernel for the LoRA-down products (gemm_skinny.hip): bit identity with the tile kernel, A/B
cd $GRAFT_REPO_ROOT
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_ops_gpu.py -q -m gpu -k "skinny" -x 2>&1 | tail -15 | tee gpurun_out/r06_gemm_skinny_tests.log
timeout 900 python tools/gemm_skinny_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_gemm_skinny_ab.log
