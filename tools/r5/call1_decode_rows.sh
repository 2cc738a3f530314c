#!/bin/bash
# next round, first decode call: fused chain vs layer-by-layer step at 5 .. 8 rows.
# COGV_DECODE_CHAIN_MAX_ROWS=4 sends batches above four rows through the Sandwich-LN launches + plain matrix-core products.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
for b in ${MB_ROWS_BATCHES:-6 8}; do
  for rows in 8 4; do
    echo "== batch $b COGV_DECODE_CHAIN_MAX_ROWS=$rows"
    COGV_DECODE_CHAIN_MAX_ROWS=$rows MB_DECODE_BATCH=$b MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
  done
done | tee gpurun_out/r5/c1_decode_chain_rows_ab.log
