#!/bin/bash
python -m pytest tests/test_depth_parity_gpu.py tests/test_stream_kernels_gpu.py tests/test_vqvae_gpu.py -m gpu -q -x -k "not 4B" > gpurun_out/r3_bis_a.log 2>&1
echo "a (depth small) rc=$? $(tail -1 gpurun_out/r3_bis_a.log | cut -c1-100)"
python -m pytest tests/test_depth_parity_gpu.py tests/test_stream_kernels_gpu.py tests/test_vqvae_gpu.py -m gpu -q -x > gpurun_out/r3_bis_b.log 2>&1
echo "b (depth all) rc=$? $(tail -1 gpurun_out/r3_bis_b.log | cut -c1-100)"
