#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -q -x -k "gemm" > gpurun_out/r3_e5_tests.log 2>&1; tail -2 gpurun_out/r3_e5_tests.log | cut -c1-300
for lib in r3base new r3base new; do
  if [ $lib = new ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
  echo "lib=$lib"; python tools/mb_epi.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['N'], d['K'], {k[:-3]: v for k, v in d.items() if k.endswith('_TF')})"
done
