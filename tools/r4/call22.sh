#!/bin/bash
# round 4: wave counts per row count + one-column K = 10240 class: tests, captured decode at batch 1 / 2 / 4 / 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_stream_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "skinny or gemv or decode or decoding or decoder or output_projection or kv_cache or gelu_epilogue_forms" ) > gpurun_out/r4/c22_tests.log 2>&1
tail -5 gpurun_out/r4/c22_tests.log | cut -c1-200
for b in 2 4 8; do
  echo "== batch $b"
  MB_DECODE_BATCH=$b MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
done | tee gpurun_out/r4/c22_decode_batch.log
