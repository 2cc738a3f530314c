#!/bin/bash
# round 4: kernel statistics of the captured decode step with the second-generation skinny-M kernels; fused-combine form under capture
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
(cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/r4/prof; MB_DECODE_GRAPH_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof -- python $R/tools/mb_decode.py > $R/gpurun_out/r4/c16_prof.log 2>&1)
cp $(ls $R/gpurun_out/r4/prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/c16_decode_kernel_stats.csv
rm -rf $R/gpurun_out/r4/prof
head -12 gpurun_out/r4/c16_decode_kernel_stats.csv | cut -c1-150
for v in 1 0; do
  echo "== COGV_DECODE_FUSE_COMBINE=$v"
  COGV_DECODE_FUSE_COMBINE=$v MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
done | tee gpurun_out/r4/c16_fuse_ab.log
