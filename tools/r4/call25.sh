#!/bin/bash
# round 4: kernel statistics of the captured decode step at batch 8 (FormM)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
(cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/r4/prof; MB_DECODE_BATCH=8 MB_DECODE_GRAPH_ONLY=1 timeout 50 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof -- python $R/tools/mb_decode.py > $R/gpurun_out/r4/c25_prof.log 2>&1)
cp $(ls $R/gpurun_out/r4/prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/c25_decode_b8_kernel_stats.csv
rm -rf $R/gpurun_out/r4/prof
head -12 gpurun_out/r4/c25_decode_b8_kernel_stats.csv | cut -c1-170
