#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_model_gpu.py tests/test_bench_launch.py -m gpu -q -x -k "rccl or data_parallel_wrapper or bench_gpus" > gpurun_out/r3_e1_tests.log 2>&1; tail -3 gpurun_out/r3_e1_tests.log | cut -c1-300
for blk in 0 256 512 768 1024; do
  echo "COGV_LN_BWD_BLOCKS=$blk"; COGV_LN_BWD_BLOCKS=$blk python tools/r3/mb_ln_stream.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['h'], {k: v for k, v in d.items() if k.startswith('bwd') and k.endswith('_us')})"
done
for blk in 512 768 1024 1536; do
  echo "COGV_LN_FWD_BLOCKS=$blk"; COGV_LN_FWD_BLOCKS=$blk python tools/r3/mb_ln_stream.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['h'], {k: v for k, v in d.items() if k.startswith('fwd') and k.endswith('_us')})"
done
