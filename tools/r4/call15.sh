#!/bin/bash
# round 4, second session: second-generation skinny-M kernels (gemv.hip): decode tests, then captured decode latency new vs COGV_GEMV2=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_stream_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "skinny or gemv or decode or decoding or decoder or output_projection or kv_cache" ) > gpurun_out/r4/c15_tests.log 2>&1
tail -15 gpurun_out/r4/c15_tests.log
for rep in 1 2; do
  for v in 1 0; do
    echo "== COGV_GEMV2=$v (rep $rep)"
    COGV_GEMV2=$v MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
  done
done | tee gpurun_out/r4/c15_decode_ab.log
