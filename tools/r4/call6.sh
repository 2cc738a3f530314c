#!/bin/bash
# round 4, GPU call 6: cross-item prefetch -- GEMM suite, A/B against COGV_GEMM_XP=0 in alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -x -q -k "gemm or cross_item" > gpurun_out/r4/c6_tests.log 2>&1
tail -3 gpurun_out/r4/c6_tests.log
rm -f gpurun_out/r4/c6_gemm_xp_ab.log
for rep in 1 2; do
  for xp in 1 0; do
    COGV_GEMM_XP=$xp timeout 600 python tools/r4/mb_gemm_ab.py xp$xp >> gpurun_out/r4/c6_gemm_xp_ab.log 2>&1
  done
done
grep -v check gpurun_out/r4/c6_gemm_xp_ab.log | grep -v amdgpu | sort -k3,4 -k1,1
