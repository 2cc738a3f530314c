#!/bin/bash
# round 4, GPU call 10: arbitrary-mask attention, VQ-VAE topologies, smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vqvae_gpu.py -m gpu -x -q -k "attention or vqvae or topolog or mask or img2code or code2img or conv" > gpurun_out/r4/c10_tests.log 2>&1
tail -25 gpurun_out/r4/c10_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
