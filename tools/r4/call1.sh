#!/bin/bash
# round 4, GPU call 1: host facts, the new parity tests (cfg 1, gradients at 24 / 48 layers, MP = 2 gradient shards), GEMM item-phase probe,
# GEMM suite on the PEEL-default library
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
{ nproc; grep -E "MemTotal|MemAvailable" /proc/meminfo; rocm-smi --showmeminfo vram 2>/dev/null | head -5; } > gpurun_out/r4/host.txt 2>&1
timeout 900 python tools/probes/w4_ts.py > gpurun_out/r4/w4_ts.log 2>&1
timeout 900 python -m pytest tests/test_cfg1_gpu.py -m gpu -x -q -s > gpurun_out/r4/cfg1.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -x -q -k "gemm" > gpurun_out/r4/gemm_tests.log 2>&1
timeout 1500 python -m pytest tests/test_depth_parity_gpu.py -m gpu -q -s -k "gradients" > gpurun_out/r4/depth_grads.log 2>&1
tail -3 gpurun_out/r4/cfg1.log gpurun_out/r4/gemm_tests.log gpurun_out/r4/depth_grads.log; cat gpurun_out/r4/host.txt
