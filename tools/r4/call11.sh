#!/bin/bash
# round 4, GPU call 11: two cheap step-level knobs (weight-gradient grouping 2 layers; raster group height 8), alternating runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/c11_knobs.log
run() { env "$@" timeout 600 python bench.py --dtype fp16 --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],2))" >> gpurun_out/r4/c11_knobs.log; }
run X=base
run COGV_WGRAD_GROUP_LAYERS=2
run COGV_GEMM_GROUP_M=8
run X=base
run COGV_WGRAD_GROUP_LAYERS=2
run COGV_GEMM_GROUP_M=2
cat gpurun_out/r4/c11_knobs.log
