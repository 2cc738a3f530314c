#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_stream_kernels_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "sandwich or ln or layernorm or gemv" > gpurun_out/r3_e2_tests.log 2>&1; tail -2 gpurun_out/r3_e2_tests.log | cut -c1-200
show() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); print(d['h'], {k[:-3]: v for k, v in d.items() if k.endswith('_us') and 'all16' not in k})"; }
echo "quad map (default)"; python tools/r3/mb_ln_stream.py 2>/dev/null | show
echo "quad map, residual loaded late"; COGV_LN_EARLY_RES=0 python tools/r3/mb_ln_stream.py 2>/dev/null | show
echo "old map"; COGV_LN_QUAD_MAP=0 python tools/r3/mb_ln_stream.py 2>/dev/null | show
echo "quad map, bwd 4 rows"; COGV_LN_BWD_ROWS=4 python tools/r3/mb_ln_stream.py 2>/dev/null | show
