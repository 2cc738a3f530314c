#!/bin/bash
# round 4: GEMM generations 2 / 3 / 4 on the 336M shapes (K = 1024 / 4096), bf16, b = 30
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
MB_BATCH=30 MB_SMALL=1 MB_VARIANTS=0,3,9 timeout 300 python tools/mb_gemm.py 2>&1 | tee gpurun_out/r4/c18_gemm_variants_336M.log
