#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 600 python tools/r4/find_small_ops.py > gpurun_out/r4/c13_small_ops.log 2>&1
tail -45 gpurun_out/r4/c13_small_ops.log
