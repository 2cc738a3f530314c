#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or sparse" 2>&1 | tail -2
for lib in nopipe new nopipe new; do
  if [ $lib = new ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
  echo "lib=$lib"; python tools/mb_attn.py 2>/dev/null | grep '"H": 40'
done
