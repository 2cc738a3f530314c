#!/bin/bash
# round 4: captured decode step at batch 2 / 4 / 8, second-generation skinny-M kernels vs COGV_GEMV2=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
for b in 2 4 8; do
  for v in 1 0; do
    echo "== batch $b COGV_GEMV2=$v"
    MB_DECODE_BATCH=$b COGV_GEMV2=$v MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
  done
done | tee gpurun_out/r4/c19_decode_batch_ab.log
