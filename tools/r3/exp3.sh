#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -q -x -k "gelu or gemv or scale" > gpurun_out/r3_e3_tests.log 2>&1; tail -2 gpurun_out/r3_e3_tests.log | cut -c1-200
for lib in base new base new; do
  if [ $lib = base ]; then export COGVIEW_HIP_LIB=$R/build/ab/libcogview_r3base.so; else unset COGVIEW_HIP_LIB; fi
  echo "lib=$lib"; python tools/mb_epi.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['K'] in (2560, 1024) and d['N'] in (10240, 4096): print(d['M'], d['N'], d['K'], {k: v for k, v in d.items() if 'gelu' in k or k.startswith('bias_')})"
done
