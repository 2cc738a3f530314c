#!/bin/bash
# round 4, GPU call 8: non-temporal C stores A/B (microbench + a short in-step check)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/c8_nt_ab.log
for rep in 1 2; do
  for lib in prod ntstore; do
    if [ $lib = prod ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
    timeout 600 python tools/r4/mb_gemm_ab.py $lib >> gpurun_out/r4/c8_nt_ab.log 2>&1
  done
done
grep -v "check\|amdgpu" gpurun_out/r4/c8_nt_ab.log | sort -k3,4 -k1,1
for lib in prod ntstore prod ntstore; do
  if [ $lib = prod ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
  timeout 600 python bench.py --dtype fp16 --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), round(d['ms_per_step'],2))" >> gpurun_out/r4/c8_nt_step.log
done
cat gpurun_out/r4/c8_nt_step.log
