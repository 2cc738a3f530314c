#!/bin/bash
# round 4, GPU call 2: attention keep-bit tests + A/B, GEMM suite + A/B of the pipelined epilogue
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -x -q -k "gemm or attention" > gpurun_out/r4/c2_tests.log 2>&1
tail -3 gpurun_out/r4/c2_tests.log
timeout 600 python tools/r4/mb_attn_keepbits.py > gpurun_out/r4/c2_attn_ab.log 2>&1
for rep in 1 2; do
  for lib in new oldepi; do
    if [ $lib = new ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
    timeout 600 python tools/r4/mb_gemm_ab.py $lib >> gpurun_out/r4/c2_gemm_ab.log 2>&1
  done
done
unset COGVIEW_HIP_LIB
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q > gpurun_out/r4/c2_model_tests.log 2>&1
tail -3 gpurun_out/r4/c2_model_tests.log
cat gpurun_out/r4/c2_attn_ab.log | tail -8
