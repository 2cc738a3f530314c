#!/bin/bash
# round 4, GPU call 4: counters of the dominant kernel, CU-contention experiment, re-run of the fixed test
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_stream_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 bash tools/r4/pmc_gemm.sh > /dev/null 2>&1
cat gpurun_out/r4/pmc_gemm_report.txt | cut -c1-330
timeout 600 python tools/r4/contention.py --steps 4 > gpurun_out/r4/contention.log 2> gpurun_out/r4/contention.err
cat gpurun_out/r4/contention.log; tail -3 gpurun_out/r4/contention.err
