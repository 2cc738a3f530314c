#!/bin/bash
# round 4: decode step after the tail trims (bias / abs-max prefetch, direct atomic, unconditional cache reads in the decode attention,
# combine loads hoisted): tests, then captured decode latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_stream_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "skinny or gemv or decode or decoding or decoder or output_projection or kv_cache" ) > gpurun_out/r4/c17_tests.log 2>&1
tail -6 gpurun_out/r4/c17_tests.log
for rep in 1 2; do
  MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder
done | tee gpurun_out/r4/c17_decode.log
(cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/r4/prof; MB_DECODE_GRAPH_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof -- python $R/tools/mb_decode.py > $R/gpurun_out/r4/c17_prof.log 2>&1)
cp $(ls $R/gpurun_out/r4/prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/c17_decode_kernel_stats.csv
rm -rf $R/gpurun_out/r4/prof
head -9 gpurun_out/r4/c17_decode_kernel_stats.csv | cut -c1-150
