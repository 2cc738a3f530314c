#!/bin/bash
# round 5, fifth GPU call: lean LayerNorm backward mismatch statistics; the whole GPU suite at HEAD with durations; smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
timeout 300 python tools/r5/dbg_ln_lean.py 2>&1 | tee $OUT/c5_dbg_ln_lean.log
( time timeout 1500 python -m pytest tests -m gpu -q --durations=30 ) > $OUT/c5_gpu_tests.log 2>&1; tail -45 $OUT/c5_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/c5_smoke.log
